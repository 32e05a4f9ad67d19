"""In-tree build of the C-ABI library ``lib/libpyg_amd.so`` (hipcc, gfx950 only).

The library is plain HIP + rocPRIM headers; it does not link against torch.  Objects go to
``build/`` (git-ignored), the shared object stays in-tree next to the package so that it travels
with a repo snapshot to a GPU box (``*.so`` is git-ignored but not gpurun-ignored).
"""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC_DIR = os.path.join(PKG_DIR, 'csrc')
LIB_DIR = os.path.join(PKG_DIR, 'lib')
LIB_PATH = os.path.join(LIB_DIR, 'libpyg_amd.so')
BUILD_DIR = os.path.join(os.path.dirname(PKG_DIR), 'build', 'pyg_amd')
SOURCES = ['capi.hip', 'graph.hip', 'spmm.hip', 'scatter.hip', 'softmax.hip', 'segmm.hip',
           'sample.hip', 'gemm.hip', 'sage_fused.hip']
ARCH = 'gfx950'
FLAGS = [f'--offload-arch={ARCH}', '-O3', '-std=c++17', '-fPIC', '-fvisibility=hidden',
         '-Wall', '-Wno-unused-function']


def find_hipcc():
    cand = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    return cand if os.path.exists(cand) else None


STAMP_PATH = LIB_PATH + '.stamp'


def source_hash():
    """sha1 over every source the library is built from (mtimes do not survive a snapshot)."""
    import hashlib
    h = hashlib.sha1()
    paths = sorted(os.path.join(CSRC_DIR, f) for f in os.listdir(CSRC_DIR))
    paths.append(os.path.join(os.path.dirname(PKG_DIR), 'include', 'pyg_amd.h'))
    for p in paths:
        if os.path.isfile(p):
            h.update(os.path.basename(p).encode())
            with open(p, 'rb') as f:
                h.update(f.read())
    h.update(' '.join(FLAGS).encode())
    return h.hexdigest()


def is_stale():
    if not (os.path.exists(LIB_PATH) and os.path.exists(STAMP_PATH)):
        return True
    with open(STAMP_PATH) as f:
        return f.read().strip() != source_hash()


def build_library(force=False, verbose=True):
    """Compile every HIP source for gfx950 and link ``libpyg_amd.so``. Returns its path."""
    if not force and not is_stale():
        return LIB_PATH
    hipcc = find_hipcc()
    if hipcc is None:
        raise RuntimeError('hipcc not found: cannot build libpyg_amd.so')
    os.makedirs(BUILD_DIR, exist_ok=True)
    os.makedirs(LIB_DIR, exist_ok=True)
    headers = [os.path.join(CSRC_DIR, f) for f in os.listdir(CSRC_DIR) if f.endswith('.h')]
    headers.append(os.path.join(os.path.dirname(PKG_DIR), 'include', 'pyg_amd.h'))
    headers_mtime = max(os.path.getmtime(h) for h in headers)

    def compile_one(src):
        src_path = os.path.join(CSRC_DIR, src)
        obj = os.path.join(BUILD_DIR, src.replace('.hip', '.o'))
        if (not force and os.path.exists(obj)
                and os.path.getmtime(obj) >= max(os.path.getmtime(src_path), headers_mtime)):
            return obj
        cmd = [hipcc] + FLAGS + ['-c', src_path, '-o', obj]
        if verbose:
            print('[pyg_amd build]', ' '.join(cmd), file=sys.stderr, flush=True)
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError(f'hipcc failed for {src}:\n{res.stdout}\n{res.stderr}')
        return obj

    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    tmp = LIB_PATH + '.tmp'
    cmd = [hipcc, f'--offload-arch={ARCH}', '-shared', '-fPIC', '-o', tmp] + objs
    if verbose:
        print('[pyg_amd build]', ' '.join(cmd), file=sys.stderr, flush=True)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f'link failed:\n{res.stdout}\n{res.stderr}')
    os.replace(tmp, LIB_PATH)
    with open(STAMP_PATH, 'w') as f:
        f.write(source_hash())
    return LIB_PATH


if __name__ == '__main__':
    print(build_library(force='--force' in sys.argv))
