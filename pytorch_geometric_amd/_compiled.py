"""Loader of the compiled PyTorch binding (``lib/libpyg_amd_torch.so``, csrc/torch_binding.cpp):
``torch.ops.pyg_amd_c.*`` = the hot entry points of the C ABI as dispatcher operators with a HIP
implementation in C++ (TORCH_LIBRARY / TORCH_LIBRARY_IMPL).  ``ops()`` returns the operator
namespace, or ``None`` when the binding is switched off (``PYGAMD_BINDING=ctypes``) or its shared
object is absent and cannot be built — ``_native`` then takes its ctypes path, which calls the very
same C entry points.  Every operator follows the ``out=`` convention (mutable arguments, nothing
returned), so its shape function for FakeTensor / torch.compile is trivial; they are registered
here."""
import os

import torch

from . import _build, _lib

_state = {'tried': False, 'ns': None, 'error': None}

OPS = ('spmm_csr', 'linear_forward', 'linear_dgrad', 'linear_wgrad', 'sage_layer_fused',
       'gather_scatter_add')                                   # write into caller-allocated tensors
FUNCTIONAL_OPS = ('index2ptr', 'ptr2index', 'gather_rows', 'sddmm_csr', 'segment_softmax_forward',
                  'segment_softmax_backward')                  # return a fresh tensor


def _register_fakes():
    from torch.library import register_fake

    for name in OPS:
        register_fake(f'pyg_amd_c::{name}')(lambda *args, **kwargs: None)
    register_fake('pyg_amd_c::index2ptr')(lambda index, size: index.new_empty(size + 1))
    register_fake('pyg_amd_c::ptr2index')(lambda ptr, n: ptr.new_empty(n))
    register_fake('pyg_amd_c::gather_rows')(
        lambda x, index: x.new_empty(index.numel(), x.size(1)))
    register_fake('pyg_amd_c::sddmm_csr')(
        lambda rowptr, col, eid, grad_out, x, n_edges, w_heads: x.new_empty(n_edges, w_heads))
    register_fake('pyg_amd_c::segment_softmax_forward')(lambda src, ptr: torch.empty_like(src))
    register_fake('pyg_amd_c::segment_softmax_backward')(
        lambda out, grad_out, ptr: torch.empty_like(out))
    # the C++ autograd nodes (shape functions only: under torch.compile the Python route is taken)
    register_fake('pyg_amd_c::linear_ag')(
        lambda x, weight, bias: x.new_empty(*x.shape[:-1], weight.size(0)))
    register_fake('pyg_amd_c::bias_act_ag')(lambda x, bias, relu: torch.empty_like(x))
    register_fake('pyg_amd_c::spmm_ag')(
        lambda x, w, rowptr, *rest: x.new_empty(rowptr.numel() - 1, *x.shape[1:]))


def ops():
    if _lib.lab_active():  # a laboratory schedule is selected: every call through ctypes into
        return None        # libpyg_amd_lab.so (the binding is linked against the product library)
    if _state['tried']:
        return _state['ns']
    _state['tried'] = True
    if os.environ.get('PYGAMD_BINDING', 'compiled') == 'ctypes':
        return None
    try:
        _lib.load()  # libpyg_amd.so first: the binding resolves its symbols against it
        path = _build.BINDING_PATH
        if _build.binding_is_stale() and _build.find_hipcc() is not None:
            path = _build.build_torch_binding(verbose=False)
        torch.ops.load_library(path)
        ns = torch.ops.pyg_amd_c
        if int(ns.abi_version()) != _lib.ABI_VERSION:
            raise RuntimeError(f'binding built against ABI {int(ns.abi_version())}, '
                               f'library is {_lib.ABI_VERSION}')
        _register_fakes()
        _state['ns'] = ns
    except Exception as exc:  # noqa: BLE001 (absent / unbuildable binding: ctypes path)
        _state['error'] = exc
    return _state['ns']


def status() -> str:
    ops()
    if _state['ns'] is not None:
        return f'compiled ({_build.BINDING_PATH})'
    return f'ctypes ({_state["error"]})' if _state['error'] else 'ctypes (PYGAMD_BINDING=ctypes)'
