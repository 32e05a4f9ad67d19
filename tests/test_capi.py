"""The C-ABI library builds for gfx950, loads without a GPU, and exports exactly the symbols
include/pyg_amd.h declares (no compute calls here)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(header='pyg_amd.h'):
    text = open(os.path.join(ROOT, 'include', header)).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'PYGAMD_API\s+[\w\s\*]+?\b(pygamd_\w+)\s*\(', text)))


def test_header_declares_entry_points():
    syms = declared_symbols()
    assert len(syms) >= 25
    for must in ['pygamd_spmm_csr', 'pygamd_index_sort', 'pygamd_scatter_rows',
                 'pygamd_segment_softmax_forward', 'pygamd_gat_edge_softmax_forward']:
        assert must in syms


def test_library_loads_and_exports_every_declared_symbol():
    from pytorch_geometric_amd import _build, _lib
    if _build.is_stale() and _build.find_hipcc() is None:
        pytest.skip('library not built and no hipcc here')
    lib = _lib.load()
    assert lib.pygamd_abi_version() == _lib.ABI_VERSION == 10
    assert lib.pygamd_build_arch() == b'gfx950'
    assert lib.pygamd_status_string(0) == b'ok'
    assert lib.pygamd_status_string(3) == b'workspace too small'
    declared = declared_symbols()
    for name in declared:
        assert hasattr(lib, name), f'{name} declared in pyg_amd.h but not exported'
    assert sorted(_lib.SIGNATURES) == declared, 'ctypes table and header disagree'
    out = subprocess.run(['nm', '-D', '--defined-only', _lib.lib_path()], capture_output=True,
                         text=True).stdout
    exported = sorted(set(re.findall(r' T (pygamd_\w+)', out)))
    # the laboratory entry points (schedules not adopted, timing probes) live in their own header
    # and carry their own prefix: nothing of them may leak into the boundary, and vice versa
    lab = declared_symbols('pyg_amd_lab.h')
    assert lab and all(n.startswith('pygamd_lab_') for n in lab)
    assert not any(n.startswith('pygamd_lab_') for n in declared)
    assert sorted(_lib.LAB_SIGNATURES) == lab
    # the PRODUCT library exports the boundary and nothing else; the laboratory build
    # (libpyg_amd_lab.so: scripts, schedule-pinning tests, bench side figure) adds its own
    assert exported == declared, 'libpyg_amd.so: exported symbols differ from pyg_amd.h'
    if _build.lab_is_stale() and _build.find_hipcc() is None:
        return
    lab_lib = _lib.load_lab()
    for name in declared + lab:
        assert hasattr(lab_lib, name), f'{name} not exported by the laboratory build'
    out = subprocess.run(['nm', '-D', '--defined-only', _build.LAB_LIB_PATH],
                         capture_output=True, text=True).stdout
    assert sorted(set(re.findall(r' T (pygamd_\w+)', out))) == sorted(declared + lab)
    # and the compiled binding resolves against the product library only
    if os.path.exists(_build.BINDING_PATH):
        needed = subprocess.run(['readelf', '-d', _build.BINDING_PATH], capture_output=True,
                                text=True).stdout
        assert 'libpyg_amd.so' in needed and 'libpyg_amd_lab' not in needed


def test_code_object_is_gfx950_only():
    from pytorch_geometric_amd import _build, _lib
    if not os.path.exists(_lib.lib_path()):
        pytest.skip('library not built')
    blob = open(_lib.lib_path(), 'rb').read()
    # offload-bundle entry ids name the device targets the fat binary carries
    targets = set(re.findall(rb'hipv4-amdgcn-amd-amdhsa--(gfx[0-9a-z]+)', blob))
    assert targets == {b'gfx950'}, targets
    assert '--offload-arch=gfx950' in _build.FLAGS


def test_argument_validation_without_gpu():
    """Entry points reject bad arguments before touching the device."""
    import ctypes
    from pytorch_geometric_amd import _lib
    lib = _lib.load()
    assert lib.pygamd_index_sort(None, 7, 4, -1, None, None, None, 0, None) != 0
    assert lib.pygamd_index2ptr(None, 1, -1, 4, None, None) == 1
    a = _lib.SpmmArgs()
    a.n_rows, a.F, a.ldx, a.ldo, a.idx_dtype, a.reduce = 4, 8, 4, 8, 1, 0  # ldx < F
    assert lib.pygamd_spmm_csr(ctypes.byref(a), None, 0, None) == 1
    a.ldx, a.reduce = 8, 4  # MUL is not an SpMM reduce
    assert lib.pygamd_spmm_csr(ctypes.byref(a), None, 0, None) == 2
    n = ctypes.c_size_t(123)
    a.reduce = 0
    assert lib.pygamd_spmm_csr_workspace_bytes(ctypes.byref(a), ctypes.byref(n)) == 0
    assert n.value == 0
    a.n_hub, a.n_chunks = 2, 5
    assert lib.pygamd_spmm_csr_workspace_bytes(ctypes.byref(a), ctypes.byref(n)) == 0
    assert n.value == 5 * 8 * 4


def test_pack_relu_bits_layout():
    """Host helper for the one-bit-per-element ReLU mask of include/pyg_amd.h
    (pygamd_spmm_args.relu_bits): bit (c & 31) of word [r >> 5, c >> 5, r & 31] <=> act[r, c] > 0
    (tiles of 32 x 32), unused bits / rows zero, int32 two's complement storage."""
    import torch
    from pytorch_geometric_amd._native import pack_relu_bits
    g = torch.Generator().manual_seed(5)
    for f in (1, 31, 32, 33, 100, 256):
        n = 37
        act = torch.randn(n, f, generator=g)
        act[0] = 1.0    # all bits set: exercises the sign bit of the int32 words
        act[1] = -0.0
        words = pack_relu_bits(act)
        assert words.dtype == torch.int32 and tuple(words.shape) == (2, (f + 31) // 32, 32)
        for r in range(64):
            for w in range(words.size(1)):
                want = 0
                for b in range(32):
                    c = 32 * w + b
                    if r < n and c < f and float(act[r, c]) > 0:
                        want |= 1 << b
                assert (int(words[r >> 5, w, r & 31]) & 0xffffffff) == want


def test_round2_entry_points_validate_without_gpu():
    """The entry points added in round 2 reject inconsistent arguments on the host (no device
    work is reached): mask arguments of the SpMM / dgrad epilogues, the two-operand weight
    gradient, the arithmetic mode switch."""
    import ctypes
    from pytorch_geometric_amd import _lib
    lib = _lib.load()
    # arithmetic mode: process-wide switch with a checked range
    assert lib.pygamd_get_gemm_mode() in (0, 1)
    prev = lib.pygamd_get_gemm_mode()
    assert lib.pygamd_set_gemm_mode(7) == 1 and lib.pygamd_get_gemm_mode() == prev
    assert lib.pygamd_set_gemm_mode(1) == 0 and lib.pygamd_get_gemm_mode() == 1
    assert lib.pygamd_set_gemm_mode(prev) == 0
    # SpMM: a float mask narrower than F, both mask forms at once, a mask on an extremum
    a = _lib.SpmmArgs()
    a.n_rows, a.F, a.ldx, a.ldo, a.idx_dtype, a.reduce = 4, 64, 64, 64, 1, 0
    a.rowptr = a.col = a.x = a.out = 16  # (never dereferenced: every call below is rejected)
    a.relu_mask, a.ld_mask = 16, 32
    assert lib.pygamd_spmm_csr(ctypes.byref(a), None, 0, None) == 1
    a.ld_mask, a.relu_bits, a.ld_bits = 64, 16, 2
    assert lib.pygamd_spmm_csr(ctypes.byref(a), None, 0, None) == 1
    a.relu_mask, a.ld_bits = None, 1  # 64 columns need two words per row tile
    assert lib.pygamd_spmm_csr(ctypes.byref(a), None, 0, None) == 1
    a.ld_bits, a.reduce = 2, 3
    assert lib.pygamd_spmm_csr(ctypes.byref(a), None, 0, None) == 2
    # dgrad: mask leading dimensions
    P = ctypes.c_void_p
    assert lib.pygamd_linear_dgrad(P(16), 8, P(16), 8, None, 0, 4, 8, 64, 0, P(16), 32, None, 0,
                                   P(16), 64, None) == 1
    assert lib.pygamd_linear_dgrad(P(16), 8, P(16), 8, None, 0, 4, 8, 64, 0, None, 0, P(16), 1,
                                   P(16), 64, None) == 1
    # two-operand weight gradient: a second width without a second operand (M > 0), an empty
    # first operand, an output narrower than K1 + K2
    ws = ctypes.c_size_t(0)
    assert lib.pygamd_linear_wgrad_workspace_bytes(1000, 8, 24, ctypes.byref(ws)) == 0
    assert ws.value >= 8 * 25 * 4
    assert lib.pygamd_linear_wgrad2(P(16), 8, P(16), 16, 16, None, 8, 8, 1000, 8, 0, 0, P(16), 24,
                                    None, P(16), ws.value, None) == 1
    assert lib.pygamd_linear_wgrad2(P(16), 8, P(16), 16, 0, P(16), 8, 8, 1000, 8, 0, 0, P(16), 24,
                                    None, P(16), ws.value, None) == 1
    assert lib.pygamd_linear_wgrad2(P(16), 8, P(16), 16, 16, P(16), 8, 8, 1000, 8, 0, 0, P(16), 16,
                                    None, P(16), ws.value, None) == 1
    assert lib.pygamd_linear_wgrad2(P(16), 8, P(16), 16, 16, P(16), 8, 8, 1000, 8, 0, 0, P(16), 24,
                                    None, P(16), 4, None) == 3


def test_gemm_mode_python_api_and_relu_bits_checks():
    import torch
    import pytorch_geometric_amd as pga
    from pytorch_geometric_amd import _native
    prev = pga.get_gemm_mode()
    assert prev in ('fp32', 'split')
    with pytest.raises(ValueError):
        pga.set_gemm_mode('tf32')
    assert pga.set_gemm_mode('split') == prev and pga.get_gemm_mode() == 'split'
    pga.set_gemm_mode(prev)
    bits = _native.relu_bits_like(70, 100, 'cpu')
    assert tuple(bits.shape) == (3, 4, 32) and bits.dtype == torch.int32
    _native._check_bits(bits, 70, 100)
    for bad in (bits[:, :3], bits.to(torch.int64), bits[:2], bits.transpose(0, 1)):
        with pytest.raises(ValueError):
            _native._check_bits(bad, 70, 100)


def test_torch_binding_builds_loads_and_declares_its_operators():
    """csrc/torch_binding.cpp: the TORCH_LIBRARY binding over the C ABI compiles with hipcc on a
    CPU-only box, loads, reports the library's ABI version and registers every operator with the
    `out=`-style schema (mutable arguments, no alias-or-not results) plus a fake kernel."""
    import torch
    from pytorch_geometric_amd import _build, _compiled, _lib
    if _build.binding_is_stale() and _build.find_hipcc() is None:
        pytest.skip('binding not built and no hipcc here')
    ns = _compiled.ops()
    assert ns is not None, _compiled.status()
    assert int(ns.abi_version()) == _lib.ABI_VERSION
    for name in _compiled.OPS:
        schema = str(getattr(ns, name).default._schema)
        assert '-> ()' in schema and '(a!)' in schema, schema
    for name in _compiled.FUNCTIONAL_OPS:
        schema = str(getattr(ns, name).default._schema)
        assert '-> Tensor' in schema and '!' not in schema, schema
    # CPU tensors: no kernel is registered for them (there is no CPU fallback)
    with pytest.raises((NotImplementedError, RuntimeError)):
        ns.index2ptr(torch.tensor([0, 1, 1]), 3)
    # fake tensors propagate shapes without running anything
    from torch._subclasses.fake_tensor import FakeTensorMode
    with FakeTensorMode():
        x = torch.empty(10, 8, device='cuda')
        idx = torch.empty(5, dtype=torch.int64, device='cuda')
        assert ns.gather_rows(x, idx).shape == (5, 8)
        assert ns.index2ptr(idx, 7).shape == (8, )
        out = torch.empty(10, 4, device='cuda')
        assert ns.linear_forward(x, torch.empty(4, 8, device='cuda'), None, False, out,
                                 False) is None


def test_struct_mirrors_match_the_header_field_for_field(tmp_path):
    """`_lib.SpmmArgs` / `_lib.SageFusedArgs` are hand-written mirrors of the two argument structs of
    include/pyg_amd.h.  Compile the header with gcc and compare size and every field offset: a field
    added on one side only would silently shift everything behind it."""
    import ctypes
    import re
    import subprocess
    from pytorch_geometric_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, 'include', 'pyg_amd.h')).read()

    def c_fields(struct_body):
        body = re.sub(r'/\*.*?\*/', '', struct_body, flags=re.S)
        names = []
        for decl in body.split(';'):
            decl = decl.strip()
            if decl:
                names.append(re.split(r'[\s\*]+', decl)[-1])
        return names

    m1 = re.search(r'typedef struct \{(.*?)\} pygamd_spmm_args;', header, flags=re.S)
    m2 = re.search(r'typedef struct pygamd_sage_fused_args \{(.*?)\} pygamd_sage_fused_args;',
                   header, flags=re.S)
    assert m1 and m2
    structs = {'pygamd_spmm_args': (c_fields(m1.group(1)), _lib.SpmmArgs),
               'pygamd_sage_fused_args': (c_fields(m2.group(1)), _lib.SageFusedArgs)}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "pyg_amd.h"', 'int main(void) {']
    for name, (fields, _) in structs.items():
        lines.append(f'  printf("{name} %zu\\n", sizeof({name}));')
        for f in fields:
            lines.append(f'  printf("{name}.{f} %zu\\n", offsetof({name}, {f}));')
    lines += ['  return 0;', '}']
    src = tmp_path / 'layout.c'
    src.write_text('\n'.join(lines))
    exe = tmp_path / 'layout'
    subprocess.run(['gcc', '-I', os.path.join(root, 'include'), str(src), '-o', str(exe)],
                   check=True)
    got = dict(line.rsplit(' ', 1) for line in
               subprocess.run([str(exe)], check=True, capture_output=True,
                              text=True).stdout.strip().splitlines())
    for name, (fields, mirror) in structs.items():
        assert [f for f, _ in mirror._fields_] == fields, f'{name}: field names / order differ'
        assert int(got[name]) == ctypes.sizeof(mirror), f'{name}: size'
        for f in fields:
            assert int(got[f'{name}.{f}']) == getattr(mirror, f).offset, f'{name}.{f}: offset'
