"""The C-ABI library builds for gfx950, loads without a GPU, and exports exactly the symbols
include/pyg_amd.h declares (no compute calls here)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'pyg_amd.h')).read()
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
    assert lib.pygamd_abi_version() == _lib.ABI_VERSION == 2
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
    assert exported == declared, 'exported symbols differ from the header'


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
