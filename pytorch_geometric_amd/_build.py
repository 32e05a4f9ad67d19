"""In-tree build of the C-ABI library ``lib/libpyg_amd.so`` (hipcc, gfx950 only), of its laboratory
twin ``lib/libpyg_amd_lab.so`` and of the compiled PyTorch binding ``lib/libpyg_amd_torch.so``
(csrc/torch_binding.cpp: host code only, links the C-ABI library and libtorch).

The C-ABI library is hand-written HIP and nothing else: no rocPRIM / hipCUB / rocBLAS / hipBLASLt
and no torch in it (the radix sort, the scans and the GEMMs are its own kernels).  The laboratory
library is the same objects plus ``csrc/sage_fused_lab.hip`` and ``gemm.hip`` compiled with
``-DPYGAMD_LAB=1`` (schedules measured and not adopted, timing probes: include/pyg_amd_lab.h); it is
loaded by scripts/, by the tests that pin those schedules to the production results and by the
copy-rate side figure of bench.py — never by the product path.  Objects go to ``build/``
(git-ignored), the shared objects stay in-tree next to the package so that they travel with a repo
snapshot to a GPU box (``*.so`` is git-ignored but not gpurun-ignored).
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
LAB_LIB_PATH = os.path.join(LIB_DIR, 'libpyg_amd_lab.so')
BUILD_DIR = os.path.join(os.path.dirname(PKG_DIR), 'build', 'pyg_amd')
SOURCES = ['capi.hip', 'graph.hip', 'spmm.hip', 'scatter.hip', 'softmax.hip', 'segmm.hip',
           'sample.hip', 'minibatch.hip', 'train.hip', 'gemm.hip', 'sage_fused.hip']
LAB_SOURCES = SOURCES + ['sage_fused_lab.hip']
LAB_FLAGS = {'gemm.hip': ['-DPYGAMD_LAB=1']}   # the weight-gradient variants and probes
ARCH = 'gfx950'
FLAGS = [f'--offload-arch={ARCH}', '-O3', '-std=c++17', '-fPIC', '-fvisibility=hidden',
         '-Wall', '-Wno-unused-function']
# lab builds (e.g. PYGAMD_EXTRA_HIPCC_FLAGS=-DPYGAMD_GATHER_NT=1): part of the source hash, so the
# library is rebuilt when the variable changes — and rebuilt back when it is unset again
FLAGS += os.environ.get('PYGAMD_EXTRA_HIPCC_FLAGS', '').split()
# Per-file additions (none at present; -fno-slp-vectorize on sage_fused.hip — no v_pk_add_f32 in
# the gather loop — was measured neutral to slightly negative, profiles/r04_fused_noslp_probe.txt).
EXTRA_FLAGS = {}


def find_hipcc():
    cand = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    return cand if os.path.exists(cand) else None


STAMP_PATH = LIB_PATH + '.stamp'
LAB_STAMP_PATH = LAB_LIB_PATH + '.stamp'


BINDING_SRC = os.path.join(CSRC_DIR, 'torch_binding.cpp')
BINDING_PATH = os.path.join(LIB_DIR, 'libpyg_amd_torch.so')
BINDING_STAMP = BINDING_PATH + '.stamp'


def source_hash():
    """sha1 over every source the library is built from (mtimes do not survive a snapshot)."""
    import hashlib
    h = hashlib.sha1()
    paths = sorted(os.path.join(CSRC_DIR, f) for f in os.listdir(CSRC_DIR)
                   if f != os.path.basename(BINDING_SRC))
    paths.append(os.path.join(os.path.dirname(PKG_DIR), 'include', 'pyg_amd.h'))
    paths.append(os.path.join(os.path.dirname(PKG_DIR), 'include', 'pyg_amd_lab.h'))
    for p in paths:
        if os.path.isfile(p):
            h.update(os.path.basename(p).encode())
            with open(p, 'rb') as f:
                h.update(f.read())
    h.update(' '.join(FLAGS).encode())
    h.update(repr(sorted(EXTRA_FLAGS.items())).encode())
    return h.hexdigest()


def is_stale():
    if not (os.path.exists(LIB_PATH) and os.path.exists(STAMP_PATH)):
        return True
    with open(STAMP_PATH) as f:
        return f.read().strip() != source_hash()


def lab_is_stale():
    if not (os.path.exists(LAB_LIB_PATH) and os.path.exists(LAB_STAMP_PATH)):
        return True
    with open(LAB_STAMP_PATH) as f:
        return f.read().strip() != source_hash()


def build_library(force=False, verbose=True):
    """Compile every product HIP source for gfx950 and link ``libpyg_amd.so``. Returns its path."""
    if not force and not is_stale():
        return LIB_PATH
    return _build(SOURCES, {}, LIB_PATH, STAMP_PATH, force, verbose)


def build_lab_library(force=False, verbose=True):
    """``libpyg_amd_lab.so``: the product objects (re-used) + the laboratory sources."""
    if not force and not lab_is_stale():
        return LAB_LIB_PATH
    return _build(LAB_SOURCES, LAB_FLAGS, LAB_LIB_PATH, LAB_STAMP_PATH, force, verbose)


def _build(sources, more_flags, lib_path, stamp_path, force, verbose):
    hipcc = find_hipcc()
    if hipcc is None:
        raise RuntimeError('hipcc not found: cannot build libpyg_amd.so')
    os.makedirs(BUILD_DIR, exist_ok=True)
    os.makedirs(LIB_DIR, exist_ok=True)
    headers = [os.path.join(CSRC_DIR, f) for f in os.listdir(CSRC_DIR) if f.endswith('.h')]
    headers.append(os.path.join(os.path.dirname(PKG_DIR), 'include', 'pyg_amd.h'))
    headers.append(os.path.join(os.path.dirname(PKG_DIR), 'include', 'pyg_amd_lab.h'))
    headers_mtime = max(os.path.getmtime(h) for h in headers)

    def compile_one(src):
        src_path = os.path.join(CSRC_DIR, src)
        # (objects are kept per flag set: a lab build must not pick up the default build's)
        import hashlib
        flags = FLAGS + EXTRA_FLAGS.get(src, []) + more_flags.get(src, [])
        tag = hashlib.sha1(' '.join(flags).encode()).hexdigest()[:8]
        obj = os.path.join(BUILD_DIR, src.replace('.hip', f'.{tag}.o'))
        if (not force and os.path.exists(obj)
                and os.path.getmtime(obj) >= max(os.path.getmtime(src_path), headers_mtime)):
            return obj
        cmd = [hipcc] + flags + ['-c', src_path, '-o', obj]
        if verbose:
            print('[pyg_amd build]', ' '.join(cmd), file=sys.stderr, flush=True)
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError(f'hipcc failed for {src}:\n{res.stdout}\n{res.stderr}')
        return obj

    with ThreadPoolExecutor(max_workers=min(len(sources), os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, sources))
    tmp = lib_path + '.tmp'
    cmd = [hipcc, f'--offload-arch={ARCH}', '-shared', '-fPIC', '-o', tmp] + objs
    if verbose:
        print('[pyg_amd build]', ' '.join(cmd), file=sys.stderr, flush=True)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f'link failed:\n{res.stdout}\n{res.stderr}')
    os.replace(tmp, lib_path)
    with open(stamp_path, 'w') as f:
        f.write(source_hash())
    return lib_path


def binding_hash():
    import hashlib
    import torch
    h = hashlib.sha1()
    for p in (BINDING_SRC, os.path.join(os.path.dirname(PKG_DIR), 'include', 'pyg_amd.h')):
        with open(p, 'rb') as f:
            h.update(f.read())
    h.update(torch.__version__.encode())
    return h.hexdigest()


def binding_is_stale():
    if not (os.path.exists(BINDING_PATH) and os.path.exists(BINDING_STAMP)):
        return True
    with open(BINDING_STAMP) as f:
        return f.read().strip() != binding_hash()


def build_torch_binding(force=False, verbose=True):
    """hipcc (host C++ only) -> ``lib/libpyg_amd_torch.so``: TORCH_LIBRARY operators over the C ABI,
    linked against libpyg_amd.so (found through ``$ORIGIN``) and libtorch.  Returns its path."""
    if not force and not binding_is_stale():
        return BINDING_PATH
    import torch
    from torch.utils import cpp_extension
    hipcc = find_hipcc()
    if hipcc is None:
        raise RuntimeError('hipcc not found: cannot build libpyg_amd_torch.so')
    build_library(verbose=verbose)  # the binding links the C-ABI library
    tlib = os.path.join(os.path.dirname(torch.__file__), 'lib')
    inc = cpp_extension.include_paths(device_type='cuda')
    tmp = BINDING_PATH + '.tmp'
    cmd = [hipcc, '-x', 'c++', '-O2', '-std=c++17', '-fPIC', '-shared', '-fvisibility=hidden',
           '-D__HIP_PLATFORM_AMD__=1', '-DUSE_ROCM=1',
           f'-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}']
    cmd += [f'-I{i}' for i in inc]
    cmd += [BINDING_SRC, '-o', tmp, f'-L{tlib}', '-ltorch', '-ltorch_cpu', '-lc10', '-lc10_hip',
            '-ltorch_hip', f'-L{LIB_DIR}', '-lpyg_amd', '-Wl,-rpath,$ORIGIN',
            f'-Wl,-rpath,{tlib}']
    if verbose:
        print('[pyg_amd build]', ' '.join(cmd), file=sys.stderr, flush=True)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f'building the torch binding failed:\n{res.stdout}\n{res.stderr}')
    os.replace(tmp, BINDING_PATH)
    with open(BINDING_STAMP, 'w') as f:
        f.write(binding_hash())
    return BINDING_PATH


if __name__ == '__main__':
    print(build_library(force='--force' in sys.argv))
    print(build_lab_library(force='--force' in sys.argv))
    print(build_torch_binding(force='--force' in sys.argv))
