"""``install()`` — plug the MI355X kernels into an importable ``torch_geometric`` so that the
reference's own ``nn.conv.*`` / ``nn.aggr.*`` modules pick them up unchanged.

The reference has no plugin registry: ``torch_geometric.backend`` is a module of flags
(torch_geometric/backend.py:5-11) and its dispatchers are plain functions imported by name in ~60
modules.  ``install()`` therefore

1. publishes itself on ``torch_geometric.backend`` (``backend.mi355x`` = this module,
   ``backend.use_mi355x`` flag, ``None`` = auto like ``use_segment_matmul``) — seam S5;
2. rebinds the dispatcher functions ``scatter``, ``segment``, ``segment_logsumexp``, ``softmax``, ``spmm``,
   ``index_sort``, ``scatter_argmax``, ``sort_edge_index``, ``coalesce`` in EVERY loaded ``torch_geometric*`` module whose attribute ``is`` the
   original function — seam S3 (utils/__init__.py:5-10,36);
3. replaces ``torch_geometric.edge_index._spmm`` (what ``EdgeIndex.matmul`` and therefore the
   reference's ``EdgeIndex`` fused route end in, edge_index.py:1925-1986);
4. wraps ``propagate`` of the hot conv classes (SAGEConv, GCNConv, GraphConv, GATConv) so that a
   plain ``edge_index`` tensor is sorted once (cached handle) and gather -> message -> aggregate
   runs as ONE CSR SpMM — the fused route the reference only takes for sparse ``adj_t`` inputs
   (nn/conv/message_passing.py:469-479);
5. memoises the per-forward graph rewrites of the reference's layers by INPUT IDENTITY —
   ``gcn_norm`` inside ``GCNConv(cached=False)`` (nn/conv/gcn_conv.py:241-258) and the
   ``remove_self_loops`` + ``add_self_loops`` pair of ``GATConv`` (nn/conv/gat_conv.py:334-347):
   both build a NEW ``edge_index`` tensor every call, which would make the handle cache of step 4
   miss and re-sort the graph per layer per step; the memo returns the SAME output tensors for
   the same input tensors (no values are compared, no host sync);
6. routes ``torch_geometric.nn.GraphSAGE.forward`` (models/basic_gnn.py:175-270) to the fused
   whole-stack schedule (``nn/models/_fused_sage.py``) when that computes exactly the same thing
   — plain mean/sum ``SAGEConv`` layers, ReLU, Identity norms, no dropout in effect — and a
   single ``SAGEConv.forward`` (nn/conv/sage_conv.py:118-139) of ANY model to the one-kernel layer
   (aggregation + ``lin_l`` + ``lin_r`` + bias as one kernel and one autograd node) for a square
   graph given as a plain ``edge_index`` tensor; an unweighted ``GraphConv.forward``
   (nn/conv/graph_conv.py:77-92) is the same layer and takes the same route;
   ``GCNConv.forward`` (nn/conv/gcn_conv.py:226-266) aggregates BEFORE it transforms when its input
   is the narrower side (``A (X W) = (A X) W``).

Every wrapper STEPS ASIDE to the original reference function for anything that is not a float32
HIP tensor, under ``torch.compile`` / TorchScript, or when ``backend.use_mi355x`` is False —
exactly where the reference steps aside for its own extensions (utils/_scatter.py:85,
edge_index.py:1946).  ``uninstall()`` restores every binding.
"""
import sys
from typing import Any, Callable, Dict, List, Tuple

import torch
from torch import Tensor

_state: Dict[str, Any] = {'installed': False, 'rebinds': [], 'classes': [], 'forwards': []}


def _flag_on() -> bool:
    import torch_geometric
    flag = getattr(torch_geometric.backend, 'use_mi355x', None)
    return flag is not False and not torch.jit.is_scripting()


def _compiling() -> bool:
    import torch_geometric
    try:
        return bool(torch_geometric.is_compiling())
    except Exception:  # pragma: no cover
        return False


def _enabled() -> bool:
    """The eager route (autograd Functions over ctypes) — not traceable, so it is only taken
    outside ``torch.compile``; while compiling, the dispatchers below hand the same kernels over
    as registered operators (``torch.ops.pyg_amd.*``, :mod:`.ops`) where the call maps onto one,
    and step aside to the reference otherwise."""
    return _flag_on() and not _compiling()


def _traced() -> bool:
    return _flag_on() and _compiling()


def _ours(t: Any) -> bool:
    return isinstance(t, Tensor) and t.is_cuda and t.dtype == torch.float32


def _ours_index(t: Any) -> bool:
    """A plain (not sparse, not subclassed) ``[2, E]`` int32/int64 edge list on the device."""
    return (isinstance(t, Tensor) and type(t) is Tensor and t.is_cuda and not t.is_sparse
            and t.dim() == 2 and t.size(0) == 2 and t.dtype in (torch.int32, torch.int64))


def _is_ref_edge_index(t: Any) -> bool:
    """The reference's own ``EdgeIndex`` (edge_index.py:173: a wrapper subclass over ``_data``)."""
    mod = sys.modules.get('torch_geometric.edge_index')
    return mod is not None and isinstance(t, mod.EdgeIndex)


def _plain_index(t: Any) -> Any:
    """The plain ``[2, E]`` tensor under any of the three graph arguments a layer may be handed: a
    tensor, this package's handle, or the reference's ``EdgeIndex``."""
    from .edge_index import EdgeIndex as Handle
    if isinstance(t, Handle):
        return t.edge_index
    if _is_ref_edge_index(t):
        return t._data
    return t


def _adopt(edge_index, n_src, n_dst, flip: bool = False):
    """This package's handle for a reference ``EdgeIndex`` living on the device (None when it
    cannot stand in: sizes that disagree, a CPU tensor).  The reference object already KNOWS its
    order and may hold the sorted forms — ``_indptr`` and the transposed ``_T_perm / _T_index /
    _T_indptr`` (edge_index.py:589-663) — so the handle is built with that ``sort_order`` (no
    sortedness re-check, no sort for the forward) and, for the non-flipped case, seeded with the
    cached transposed form; it is cached by the identity of ``_data`` like any plain tensor's
    handle, so the by-source sort of the backward happens once per graph object."""
    from .edge_index import CSR, adopt_sorted
    data = edge_index._data
    if not (data.is_cuda and data.dim() == 2 and data.dtype in (torch.int32, torch.int64)):
        return None
    rows, cols = edge_index._sparse_size   # sizes of data[0] / data[1]; entries may be None
    want = (n_dst, n_src) if flip else (n_src, n_dst)
    if (rows is not None and rows != want[0]) or (cols is not None and cols != want[1]):
        return None
    order = edge_index.sort_order
    if flip:
        order = {'row': 'col', 'col': 'row'}.get(order)
    handle, fresh = adopt_sorted(data, n_src, n_dst, flip, order)
    if fresh and not flip and order is not None:
        E = data.size(1)
        ptr = edge_index._indptr
        t_perm, t_idx, t_ptr = edge_index._T_perm, edge_index._T_index, edge_index._T_indptr
        same = lambda t: t is not None and t.dtype == data.dtype and t.is_contiguous()
        if order == 'col':   # sorted by destination: `_indptr` over data[1], transposed = by source
            if same(ptr) and ptr.numel() == n_dst + 1 and handle._csr is None:
                handle._csr = CSR(ptr, data[0].contiguous(),
                                  torch.arange(E, dtype=data.dtype, device=data.device),
                                  n_dst, n_src)
            if (same(t_perm) and same(t_ptr) and same(t_idx[1]) and t_ptr.numel() == n_src + 1):
                handle._csc = CSR(t_ptr, t_idx[1], t_perm, n_src, n_dst)
        else:                # sorted by source: `_indptr` over data[0], transposed = by destination
            if same(ptr) and ptr.numel() == n_src + 1 and handle._csc is None:
                handle._csc = CSR(ptr, data[1].contiguous(),
                                  torch.arange(E, dtype=data.dtype, device=data.device),
                                  n_src, n_dst)
            if (same(t_perm) and same(t_ptr) and same(t_idx[0]) and t_ptr.numel() == n_dst + 1):
                handle._csr = CSR(t_ptr, t_idx[0], t_perm, n_dst, n_src)
    return handle


def _device_graph(t: Any) -> bool:
    """A graph argument the layer routes take: a plain device ``[2, E]`` index tensor, this
    package's handle, or the reference's ``EdgeIndex`` over a device tensor."""
    return _ours_index(_plain_index(t))


def _square_graph(edge_index, n: int, s2t: bool = True):
    """``edge_index`` as the layer / stack routes want it for a square graph over ``n`` nodes: a
    reference ``EdgeIndex`` becomes this package's handle (:func:`_adopt`); anything else — and
    anything handed to a ``target_to_source`` layer, which those routes decline — is returned
    unchanged."""
    if s2t and _is_ref_edge_index(edge_index):
        got = _adopt(edge_index, n, n)
        if got is not None:
            return got
    return edge_index


def _stand_in(impl: Callable, orig: Callable) -> Callable:
    """The function object that is bound in ``orig``'s place: calling it runs ``impl`` (this
    backend's route, which falls back to ``orig`` itself), while everything that INSPECTS it finds
    ``orig`` — its source (``inspect`` follows ``__wrapped__``), its signature, its name and module,
    its attributes (TorchScript modifiers) and, because the stand-in's code runs in ``orig``'s
    module globals, the names its source refers to.  That is what TorchScript needs to "step
    aside": ``torch.jit.script(conv)`` / ``torch.jit.script(softmax)`` compile the ORIGINAL body
    (reference tests: test_gcn_conv.py:75, test_softmax.py:24, test_linear.py:135-180) exactly as
    without install(), instead of failing on a Python closure it cannot resolve.  The trampoline
    itself touches no global name (only its closure cell)."""
    import functools
    import types

    def call(*args, **kwargs):
        return impl(*args, **kwargs)

    fn = types.FunctionType(call.__code__, getattr(orig, '__globals__', call.__globals__),
                            getattr(orig, '__name__', 'call'), None, call.__closure__)
    try:
        functools.update_wrapper(fn, orig)      # __module__, __name__, __qualname__, __doc__, __dict__
    except (AttributeError, TypeError):         # pragma: no cover  (exotic callables)
        pass
    fn.__wrapped__ = orig
    fn._pygamd_impl = impl
    _share_script_overloads(fn, orig)
    return fn


_overload_names: List[str] = []


def _share_script_overloads(fn: Callable, orig: Callable) -> None:
    """``add_self_loops``, ``remove_self_loops``, ``coalesce`` and ``sort_edge_index`` are
    ``@torch.jit._overload``-ed in the reference (utils/loop.py:203-373, _coalesce.py:23-59, …).
    TorchScript keeps the uncompiled overload declarations per QUALIFIED NAME, compiles them for
    the first function object of that name it meets and then drops them (torch/jit/_script.py
    ``_get_overloads``): a stand-in and its original sharing one name would leave whichever comes
    second without overloads (seen as "Arguments for call are not valid" in a later
    ``torch.jit.script(GCNConv(...))``).  The stand-in therefore carries its own qualified name and
    its own copy of the declarations (or of the compiled set, if the original was scripted before
    install()).  Private torch registries: guarded, a torch that lacks them is left alone."""
    try:
        import torch._jit_internal as ji
        from torch.jit import _state as jit_state
        qual = ji._qualified_name(orig)
        decls = ji._get_fn_overloads(qual)
        done = jit_state._jit_function_overload_caching.get(orig, None)
        if not decls and not done:
            return
        mine = qual + '__pygamd'
        fn._jit_override_qualname = mine
        if decls:
            ji._overloaded_fns[mine] = list(decls)
            _overload_names.append(mine)
        if done:
            jit_state._jit_function_overload_caching[fn] = list(done)
    except Exception:  # pragma: no cover
        pass


def _make_dispatchers(orig: Dict[str, Callable]) -> Dict[str, Callable]:
    from . import utils as U

    norm = {'add': 'sum', 'amin': 'min', 'amax': 'max'}

    def scatter(src, index, dim=0, dim_size=None, reduce='sum'):
        if _ours(src) and _enabled():
            return U.scatter(src, index, dim, dim_size, reduce)
        r = norm.get(reduce, reduce)
        if (_ours(src) and _traced() and dim in (0, -src.dim()) and isinstance(dim_size, int)
                and index.dim() == 1 and r in ('sum', 'mean', 'min', 'max', 'mul')):
            from . import ops  # noqa: F401  (registers torch.ops.pyg_amd.*)
            return torch.ops.pyg_amd.scatter(src, index, dim_size, r)
        return orig['scatter'](src, index, dim, dim_size, reduce)

    def segment(src, ptr, reduce='sum'):
        if _ours(src) and ptr.dim() == 1 and _enabled():
            return U.segment(src, ptr, reduce)
        if (_ours(src) and ptr.dim() == 1 and _traced()
                and norm.get(reduce, reduce) in ('sum', 'mean', 'min', 'max')):
            from . import ops  # noqa: F401
            return torch.ops.pyg_amd.segment_csr(src, ptr, norm.get(reduce, reduce))
        return orig['segment'](src, ptr, reduce)

    def segment_logsumexp(src, ptr, dim):
        if _ours(src) and ptr.dim() == 1 and _enabled():
            return U.segment_logsumexp(src, ptr, dim)
        return orig['segment_logsumexp'](src, ptr, dim)

    def softmax(src, index=None, ptr=None, num_nodes=None, dim=0):
        if _ours(src) and _enabled():
            return U.softmax(src, index, ptr, num_nodes, dim)
        if (_ours(src) and _traced() and ptr is not None and ptr.dim() == 1
                and dim in (0, -src.dim())):
            from . import ops  # noqa: F401
            return torch.ops.pyg_amd.softmax_csr(src, ptr)
        return orig['softmax'](src, index, ptr, num_nodes, dim)

    def index_sort(inputs, max_value=None, stable=False):
        if (isinstance(inputs, Tensor) and inputs.is_cuda and inputs.dim() == 1
                and inputs.dtype in (torch.int32, torch.int64) and _enabled()):
            # the caller's `max_value` is a hint the reference's GPU path (torch.sort) never
            # trusts; a too-small one would drop radix passes and mis-sort — so it is ignored
            # here (all key bits are sorted), like torch.sort
            return U.index_sort(inputs, None, stable)
        if (isinstance(inputs, Tensor) and inputs.is_cuda and inputs.dim() == 1
                and inputs.dtype in (torch.int32, torch.int64) and _traced()):
            from . import ops  # noqa: F401
            return torch.ops.pyg_amd.index_sort(inputs, None)
        return orig['index_sort'](inputs, max_value, stable)

    def scatter_argmax(src, index, dim=0, dim_size=None):
        if _ours(src) and src.dim() == 1 and _enabled():
            return U.scatter_argmax(src, index, dim, dim_size)
        return orig['scatter_argmax'](src, index, dim, dim_size)

    def spmm(src, other, reduce='sum'):
        # sum / mean only: for min / max over a `torch.sparse` matrix the reference's own device
        # path raises NotImplementedError("... not yet supported ...") (utils/_spmm.py:92-99, pinned
        # by its test_spmm.py:50-82) — the call is left to it.  (This package's own
        # `utils.spmm` does take unit-valued min / max.)
        if (isinstance(src, Tensor) and type(src) is Tensor and src.is_cuda and _ours(other)
                and src.layout in (torch.sparse_csr, torch.sparse_coo, torch.sparse_csc)
                and reduce in ('sum', 'add', 'mean')
                and src.dim() == 2 and src.values().dim() == 1 and _enabled()):
            # the conversion the reference warns about happens here too (the handle is built from
            # a CSR view of the matrix): same warning, same conditions (utils/_spmm.py:101-118)
            if src.layout == torch.sparse_coo or (src.layout == torch.sparse_csc
                                                  and not other.requires_grad):
                import warnings
                warnings.warn(f"Converting sparse tensor to CSR format for more "
                              f"efficient processing. Consider converting your "
                              f"sparse tensor to CSR format beforehand to avoid "
                              f"repeated conversion (got '{src.layout}')", stacklevel=3)
            return U.spmm(src, other, reduce)
        return orig['spmm'](src, other, reduce)

    def _plain_edges(edge_index) -> bool:
        return _ours_index(edge_index) and _enabled()

    def sort_edge_index(edge_index, edge_attr=U._sort_edge_index.MISSING, num_nodes=None,
                        sort_by_row=True):
        if _plain_edges(edge_index):
            return U.sort_edge_index(edge_index, edge_attr, num_nodes, sort_by_row)
        return orig['sort_edge_index'](edge_index, edge_attr, num_nodes, sort_by_row)

    def coalesce(edge_index, edge_attr=U._sort_edge_index.MISSING, num_nodes=None, reduce='sum',
                 is_sorted=False, sort_by_row=True):
        attrs = edge_attr if isinstance(edge_attr, (list, tuple)) else [edge_attr]
        if _plain_edges(edge_index) and all(not isinstance(a, Tensor) or _ours(a)
                                            for a in attrs):
            return U.coalesce(edge_index, edge_attr, num_nodes, reduce, is_sorted, sort_by_row)
        return orig['coalesce'](edge_index, edge_attr, num_nodes, reduce, is_sorted, sort_by_row)

    new = dict(scatter=scatter, segment=segment, segment_logsumexp=segment_logsumexp,
               softmax=softmax, index_sort=index_sort,
               scatter_argmax=scatter_argmax, spmm=spmm, sort_edge_index=sort_edge_index,
               coalesce=coalesce)
    return {name: _stand_in(fn, orig[name]) for name, fn in new.items()}


def _sweep(old: Callable, new: Callable) -> List[Tuple[Any, str, Callable]]:
    """Rebind every attribute of every loaded torch_geometric module that IS ``old``."""
    done = []
    for modname, mod in list(sys.modules.items()):
        if mod is None or not (modname == 'torch_geometric'
                               or modname.startswith('torch_geometric.')):
            continue
        for attr, val in list(vars(mod).items()):
            if val is old:
                setattr(mod, attr, new)
                done.append((mod, attr, old))
    return done


# ---- fused propagate for the reference's own conv classes ------------------------------------
def _fused_propagate(conv, edge_index, size, kwargs):
    """Returns the aggregated tensor, or NotImplemented when this call must take the reference
    path."""
    from ._functions import spmm_node
    from .edge_index import EdgeIndex as Handle, as_edge_index
    # `fuse` is False for layers without `message_and_aggregate` (GATConv,
    # message_passing.py:154); for the others it stays the user's off switch
    fuse = True if type(conv).__name__ == 'GATConv' else getattr(conv, 'fuse', True)
    if not (_enabled() and fuse) or getattr(conv, 'explain', False):
        return NotImplemented
    # this package's handle (a Tensor subclass) keeps its sorted forms; the reference's own
    # EdgeIndex is adopted below (once the sizes are known) with the order and caches it carries
    handle = edge_index if isinstance(edge_index, Handle) else None
    if handle is not None and not (handle.is_cuda and conv.flow == 'source_to_target'):
        return NotImplemented
    ref_ei = handle is None and _is_ref_edge_index(edge_index)
    if handle is None and not ref_ei and not _ours_index(edge_index):
        return NotImplemented
    if conv._propagate_forward_pre_hooks or conv._propagate_forward_hooks:
        return NotImplemented
    if getattr(conv, 'decomposed_layers', 1) != 1:
        return NotImplemented
    aggr = conv.aggr if isinstance(conv.aggr, str) else None
    if aggr not in ('sum', 'add', 'mean', 'max', 'min'):
        return NotImplemented
    x = kwargs.get('x')
    s2t = conv.flow == 'source_to_target'
    x_src, x_dst = (x if isinstance(x, (tuple, list)) else (x, x))
    if not s2t:  # x = (x_0, x_1): messages come from x_1 and land on the x_0 side
        x_src, x_dst = x_dst, x_src
    if not _ours(x_src):
        return NotImplemented
    name = type(conv).__name__
    weight, order = None, 'coo'
    if name in ('SAGEConv', ):
        extra = set(kwargs) - {'x'}
    elif name in ('GCNConv', 'GraphConv'):
        weight = kwargs.get('edge_weight')
        extra = set(kwargs) - {'x', 'edge_weight'}
    elif name == 'GATConv':
        weight = kwargs.get('alpha')
        extra = set(kwargs) - {'x', 'alpha'}
        if weight is None:
            return NotImplemented
    else:
        return NotImplemented
    if extra or (weight is not None and not _ours(weight)):
        return NotImplemented
    if weight is not None and aggr in ('max', 'min'):
        return NotImplemented
    node_dim = conv.node_dim + x_src.dim() if conv.node_dim < 0 else conv.node_dim
    if node_dim != 0:
        return NotImplemented
    n_src = x_src.size(0)
    n_dst = x_dst.size(0) if isinstance(x_dst, Tensor) else None
    if size is not None:
        s_src, s_dst = (size[0], size[1]) if s2t else (size[1], size[0])
        n_src = s_src if s_src is not None else n_src
        n_dst = s_dst if s_dst is not None else n_dst
    if n_dst is None:
        n_dst = n_src
    if handle is not None:
        if handle.sparse_size != (n_src, n_dst):
            return NotImplemented
        graph = handle
    elif ref_ei:
        graph = _adopt(edge_index, n_src, n_dst, flip=not s2t)
        if graph is None:
            return NotImplemented
    else:
        graph = as_edge_index(edge_index, n_src, n_dst, flip=not s2t)
    reduce = 'sum' if aggr == 'add' else aggr
    if weight is not None and weight.dim() == 1 and x_src.dim() > 2:
        return NotImplemented
    # (sum / mean with plain operands: the C++ autograd node — an eager, launch-bound step such as
    # GCN on the Cora shape spends most of its host time in the Python Functions)
    out = spmm_node(x_src, weight, graph, reduce, order)
    return conv.update(out)


def _make_edge_index_spmm(orig: Callable) -> Callable:
    """Replacement for ``torch_geometric.edge_index._spmm`` (edge_index.py:1925-1970), which
    ``EdgeIndex.matmul`` resolves from the module globals at call time: a SORTED reference
    ``EdgeIndex`` times a dense float32 HIP matrix runs as our CSR SpMM (all four reductions,
    differentiable in ``other`` AND ``value``).  Unsorted inputs, bad ``reduce`` strings, CPU
    tensors etc. go to the original, which raises / computes exactly as before."""
    def _spmm(input, other, value=None, reduce='sum', transpose=False):
        from ._functions import SpmmFunction
        from .edge_index import as_edge_index
        ok = (_ours(other) and other.dim() == 2 and _enabled()
              and reduce in ('sum', 'add', 'mean', 'min', 'max')
              and (value is None or (_ours(value) and value.dim() == 1
                                     and reduce not in ('min', 'max')))  # (weighted
              # extrema: the reference's own scatter route, edge_index.py:1903-1922)
              and (input.is_sorted_by_col if transpose else input.is_sorted_by_row))
        if not ok:
            return orig(input, other, value, reduce, transpose)
        n_row, n_col = input.get_sparse_size(0), input.get_sparse_size(1)
        data = input._data if hasattr(input, '_data') else input.as_tensor()
        if transpose:   # out[col] = reduce_e value_e * other[row_e]: rows are the sources
            graph = as_edge_index(data, n_row, n_col)
        else:           # out[row] = reduce_e value_e * other[col_e]: columns are the sources
            graph = as_edge_index(data, n_col, n_row, flip=True)
        return SpmmFunction.apply(other, value, graph, 'sum' if reduce == 'add' else reduce,
                                  'coo')

    return _stand_in(_spmm, orig)


class _IdentityMemo:
    """``fn(*tensors, *rest)`` memoised on the IDENTITY (id + in-place version) of its tensor
    arguments; entries die with their inputs (weakref finalisers).  Holds at most 8 results."""

    def __init__(self, fn: Callable):
        self.fn, self.store = fn, {}

    def __call__(self, tensors: Tuple[Any, ...], rest: Tuple[Any, ...]):
        import weakref
        key = tuple((id(t), t._version) if isinstance(t, Tensor) else None for t in tensors) + rest
        hit = self.store.get(key)
        if hit is not None and all(r() is t for r, t in zip(hit[0], tensors)
                                   if isinstance(t, Tensor)):
            return hit[1]
        out = self.fn(*tensors, *rest)
        if len(self.store) >= 8:
            self.store.pop(next(iter(self.store)))

        def _drop(_, key=key, store=self.store):
            store.pop(key, None)

        refs = tuple(weakref.ref(t, _drop) if isinstance(t, Tensor) else None for t in tensors)
        self.store[key] = (refs, out)
        return out


def _memoisable(edge_index, edge_attr) -> bool:
    """(callers pass the PLAIN tensor under a handle / reference ``EdgeIndex``: the rewrites
    append self-loops, so the result is a new unsorted edge list whatever the input's class)"""
    return (_ours_index(edge_index) and _enabled()
            and (edge_attr is None or (isinstance(edge_attr, Tensor) and edge_attr.is_cuda
                                       and not edge_attr.requires_grad)))


def _make_graph_rewrite_memos(gcn_mod, gat_mod) -> List[Tuple[Any, str, Callable]]:
    """Step 5 of the module docstring.  Returns the (module, attribute, original) rebinds."""
    from .nn.conv.gcn_conv import gcn_norm as our_gcn_norm
    from .utils import loop as our_loop
    orig_norm = gcn_mod.gcn_norm
    orig_rm, orig_add = gat_mod.remove_self_loops, gat_mod.add_self_loops
    memo_norm = _IdentityMemo(our_gcn_norm)
    memo_rm = _IdentityMemo(lambda ei: our_loop.remove_self_loops(ei, None))
    memo_add = _IdentityMemo(lambda ei, n: our_loop.add_self_loops(ei, None, None, n))

    def gcn_norm(edge_index, edge_weight=None, num_nodes=None, improved=False,
                 add_self_loops=True, flow='source_to_target', dtype=None):
        plain = _plain_index(edge_index)
        if (_memoisable(plain, edge_weight) and dtype in (None, torch.float32)
                and (edge_weight is None or (edge_weight.dtype == torch.float32
                                             and edge_weight.dim() == 1))):
            return memo_norm((plain, edge_weight),
                             (num_nodes, bool(improved), bool(add_self_loops), flow, dtype))
        return orig_norm(edge_index, edge_weight, num_nodes, improved, add_self_loops, flow, dtype)

    def remove_self_loops(edge_index, edge_attr=None):
        plain = _plain_index(edge_index)
        if edge_attr is None and _memoisable(plain, None):
            return memo_rm((plain, ), ())
        return orig_rm(edge_index, edge_attr)

    def add_self_loops(edge_index, edge_attr=None, fill_value=None, num_nodes=None):
        plain = _plain_index(edge_index)
        if edge_attr is None and _memoisable(plain, None) and isinstance(num_nodes, int):
            return memo_add((plain, ), (num_nodes, ))
        return orig_add(edge_index, edge_attr, fill_value, num_nodes)

    gcn_mod.gcn_norm = _stand_in(gcn_norm, orig_norm)
    gat_mod.remove_self_loops = _stand_in(remove_self_loops, orig_rm)
    gat_mod.add_self_loops = _stand_in(add_self_loops, orig_add)
    return [(gcn_mod, 'gcn_norm', orig_norm), (gat_mod, 'remove_self_loops', orig_rm),
            (gat_mod, 'add_self_loops', orig_add)]


def _wrap_graphsage_forward(cls) -> Callable:
    """Step 6: the reference's ``GraphSAGE`` (a ``BasicGNN`` whose layers are the reference's
    ``SAGEConv``) through the fused whole-stack schedule when it is eligible."""
    orig = cls.forward

    def forward(self, x, edge_index, edge_weight=None, edge_attr=None, batch=None,
                batch_size=None, num_sampled_nodes_per_hop=None,
                num_sampled_edges_per_hop=None):
        from .nn.models import _fused_sage
        if (_enabled() and edge_weight is None and edge_attr is None
                and isinstance(x, Tensor) and num_sampled_nodes_per_hop is None):
            graph = _square_graph(edge_index, x.size(0))
            if _fused_sage.eligible(self, x, graph, False):
                return _fused_sage.run(self, x, graph)
        return orig(self, x, edge_index, edge_weight, edge_attr, batch, batch_size,
                    num_sampled_nodes_per_hop, num_sampled_edges_per_hop)

    return _stand_in(forward, orig)


def _wrap_sageconv_forward(cls) -> Callable:
    """The reference's ``SAGEConv.forward`` (nn/conv/sage_conv.py:118-139: ``propagate`` ->
    ``lin_l`` -> ``+ lin_r(x_r)`` -> optional L2 normalisation) as the one-kernel layer when the
    call is what that kernel computes (square graph, float32 device features, mean / sum, plain
    tensor ``edge_index``: ``_fused_sage.layer_eligible``); every other call runs the original,
    whose ``propagate`` is wrapped above."""
    orig = cls.forward

    def forward(self, x, edge_index, size=None):
        from .nn.models import _fused_sage
        if _enabled() and isinstance(x, torch.Tensor):
            graph = _square_graph(edge_index, x.size(0), self.flow == 'source_to_target')
            if _fused_sage.layer_eligible(self, x, graph, size):
                h = _fused_sage.run_layer(self, x, graph)
                return torch.nn.functional.normalize(h, p=2.0, dim=-1) if self.normalize else h
        return orig(self, x, edge_index, size)

    return _stand_in(forward, orig)


def _wrap_graphconv_forward(cls) -> Callable:
    """The reference's ``GraphConv.forward`` (nn/conv/graph_conv.py:77-92) without edge weights: the
    same layer as SAGEConv under the names ``lin_rel`` / ``lin_root``."""
    orig = cls.forward

    def forward(self, x, edge_index, edge_weight=None, size=None):
        from .nn.models import _fused_sage
        if _enabled() and edge_weight is None and isinstance(x, torch.Tensor):
            graph = _square_graph(edge_index, x.size(0), self.flow == 'source_to_target')
            if _fused_sage.layer_eligible(self, x, graph, size):
                return _fused_sage.run_layer(self, x, graph)
        return orig(self, x, edge_index, edge_weight, size)

    return _stand_in(forward, orig)


def _wrap_gcnconv_forward(cls) -> Callable:
    """The reference's ``GCNConv.forward`` (nn/conv/gcn_conv.py:226-266) transforms first and
    aggregates second whatever the widths.  ``A (X W) = (A X) W``: when the input is the NARROWER
    side (100 -> 256, 128 -> 256) the same layer as ``lin(propagate(x))`` gathers that much less
    and, for a first layer whose ``x`` takes no gradient, has no aggregation in its backward.
    Taken for float32 device features when the aggregation is linear (``add`` / ``sum`` / ``mean``
    with the stock ``message``: ``GCNConv(16, 64, aggr='max')`` and subclasses overriding the
    message keep the reference's order) and nobody observes the message flow (hooks, explain);
    the normalisation and its cache are the layer's own
    (``gcn_norm`` looked up in the reference's module: the identity memo of step 5 applies)."""
    orig = cls.forward

    def forward(self, x, edge_index, edge_weight=None):
        from .nn.conv.gcn_conv import linear_message_flow
        if not (_enabled() and _ours(x) and x.dim() == 2 and _device_graph(edge_index)
                and x.size(-1) < self.out_channels and getattr(self, 'aggregate_first', True)
                and getattr(self, 'fuse', True) and linear_message_flow(self, cls)):
            return orig(self, x, edge_index, edge_weight)
        import torch_geometric.nn.conv.gcn_conv as gcn_mod
        if self.normalize:
            cache = self._cached_edge_index
            if cache is None:
                edge_index, edge_weight = gcn_mod.gcn_norm(
                    edge_index, edge_weight, x.size(self.node_dim), self.improved,
                    self.add_self_loops, self.flow, x.dtype)
                if self.cached:
                    self._cached_edge_index = (edge_index, edge_weight)
            else:
                edge_index, edge_weight = cache[0], cache[1]
        out = self.lin(self.propagate(edge_index, x=x, edge_weight=edge_weight))
        return out if self.bias is None else out + self.bias

    return _stand_in(forward, orig)


_FLOW_HOOKS = ('_propagate_forward_pre_hooks', '_propagate_forward_hooks',
               '_message_forward_pre_hooks', '_message_forward_hooks',
               '_aggregate_forward_pre_hooks', '_aggregate_forward_hooks',
               '_message_and_aggregate_forward_pre_hooks',
               '_message_and_aggregate_forward_hooks', '_edge_update_forward_pre_hooks',
               '_edge_update_forward_hooks')


def _wrap_propagate(cls) -> Callable:
    orig = cls.propagate

    def propagate(self, edge_index, size=None, **kwargs):
        res = _fused_propagate(self, edge_index, size, kwargs)
        if res is NotImplemented:
            # (`size` by keyword: a layer object built BEFORE install() has left the generated
            # `propagate(self, edge_index, x, ..., size=None)` on its class, propagate.jinja:19)
            return orig(self, edge_index, size=size, **kwargs)
        return res

    fn = _stand_in(propagate, orig)   # (__module__ = orig's: `_set_jittable_templates` looks at it)
    fn._pygamd_propagate = True
    return fn


def _capturing(t: Tensor) -> bool:
    """The routes below read sizes back to the host (segment pointers, hub plans, index checks):
    not inside a hipGraph capture."""
    return t.is_cuda and torch.cuda.is_current_stream_capturing()


def _quiet(conv) -> bool:
    """Nobody observes this layer's message flow: no hooks of any kind on ``propagate`` /
    ``message`` / ``aggregate`` / ``edge_update``, no explain mode, no decomposed layers."""
    return (not getattr(conv, 'explain', False) and getattr(conv, 'decomposed_layers', 1) == 1
            and not any(getattr(conv, name, None) for name in _FLOW_HOOKS))


def _stock(conv, base, names) -> bool:
    """``conv``'s class takes these methods from ``base`` unchanged (a subclass that overrides
    ``message`` inherits the wrapped ``forward`` too and must keep the reference's route)."""
    kind = type(conv)
    return all(getattr(kind, n, None) is getattr(base, n, None) for n in names)


def _make_index_select(orig: Callable) -> Callable:
    """Replacement for ``MessagePassing._index_select`` (nn/conv/message_passing.py:263-290) — the
    gather behind every ``x_j`` / ``x_i`` / ``alpha_j`` of the general (un-fused) route, in the
    Python ``_collect`` and in the generated ``collect`` alike (collect.jinja:124-137): float32
    device rows go through ``pygamd_gather_rows``, whose backward is the sorted scatter (one cached
    radix sort of the index instead of one atomic per element — profiles/r05_unfused_propagate.md).
    Out-of-range / negative indices raise the reference's ``IndexError`` texts; WHEN follows
    ``PYGAMD_CHECK_INDEX`` (``sync``: at the call, as the reference's CPU path does; ``async``,
    the default: at the next checked launch, as the reference's GPU path — a device assert —
    does).  Everything else runs the original."""
    def _index_select(self, src, index):
        if (_ours(src) and _enabled() and isinstance(index, Tensor) and index.dim() == 1
                and index.dtype in (torch.int32, torch.int64) and src.dim() >= 1):
            from ._functions import GatherFunction
            if hasattr(index, '_data'):   # the reference's `Index` (a row of its EdgeIndex)
                index = index._data
            if type(index) is Tensor and index.is_cuda:
                d = self.node_dim + src.dim() if self.node_dim < 0 else self.node_dim
                s0 = src if d == 0 else src.movedim(d, 0).contiguous()
                out = GatherFunction.apply(s0, index, 'edge_index')
                return out if d == 0 else out.movedim(0, d)
        return orig(self, src, index)

    return _stand_in(_index_select, orig)


def _wrap_gatconv_forward(cls) -> Callable:
    """The reference's ``GATConv.forward`` (nn/conv/gat_conv.py:254-385) as projection + ONE
    autograd node for node terms, edge softmax and aggregation (``GatAttendFunction``: three
    forward kernels, the two gradients of the projected features meeting inside one backward
    kernel) when nothing in between is observable: one shared projection (``lin``), no edge
    features, no dropout in effect, nobody asking for the coefficients, no hooks.  Every other
    call runs the original, whose ``propagate`` / ``softmax`` / ``_index_select`` are served by
    this backend too."""
    orig = cls.forward

    def forward(self, x, edge_index, edge_attr=None, size=None, return_attention_weights=None):
        if not (_enabled() and _ours(x) and x.dim() == 2 and x.size(0) > 0
                and edge_attr is None and size is None
                and return_attention_weights is None and self.lin is not None
                and self.edge_dim is None and self.flow == 'source_to_target'
                and getattr(self, 'fuse_attention', True)
                and isinstance(self.aggr, str) and self.aggr in ('add', 'sum')
                and not (self.training and self.dropout > 0) and _device_graph(edge_index)
                and _quiet(self)
                and _stock(self, cls, ('message', 'edge_update', 'aggregate', 'update'))
                and _ours(self.att_src) and not torch.is_autocast_enabled()
                and not _capturing(x)):
            return orig(self, x, edge_index, edge_attr, size, return_attention_weights)
        import torch_geometric.nn.conv.gat_conv as gat_mod
        from ._functions import GatAttendFunction, bias_act
        from .edge_index import as_edge_index
        H, C, n = self.heads, self.out_channels, x.size(0)
        res = self.res(x) if getattr(self, 'res', None) is not None else None
        x_src = self.lin(x).view(-1, H, C)
        if self.add_self_loops:
            # (the identity memos of step 5: the same tensors for the same input, so the handle
            # below is found again on the next forward)
            ei, _ = gat_mod.remove_self_loops(edge_index, None)
            ei, _ = gat_mod.add_self_loops(ei, None, fill_value=self.fill_value, num_nodes=n)
            graph = as_edge_index(ei, n, n)
        else:
            graph = _square_graph(edge_index, n)
            if _is_ref_edge_index(graph):   # (sizes that disagree with `x`)
                return orig(self, x, edge_index, edge_attr, size, return_attention_weights)
            graph = as_edge_index(graph, n, n)
        out = GatAttendFunction.apply(x_src, self.att_src, self.att_dst, graph,
                                      self.negative_slope, n)
        out = out.reshape(-1, H * C) if self.concat else out.mean(dim=1)
        if res is not None:
            out = out + res
        return bias_act(out, self.bias, False) if self.bias is not None else out

    return _stand_in(forward, orig)


def _rgcn_args_ok(conv, x, edge_index, edge_type) -> bool:
    x_l = x[0] if isinstance(x, tuple) else x
    x_r = x[1] if isinstance(x, tuple) else x_l
    w = conv.weight
    if not (_ours(w) and isinstance(edge_type, Tensor) and edge_type.is_cuda
            and edge_type.dim() == 1 and edge_type.dtype in (torch.int32, torch.int64)
            and _device_graph(edge_index)):
        return False
    if edge_type.numel() != _plain_index(edge_index).size(1) or edge_type.numel() == 0:
        return False
    for t in (x_l, x_r):
        if t is None:
            continue
        if not (isinstance(t, Tensor) and t.is_cuda):
            return False
        if t.is_floating_point() and not (t.dtype == torch.float32 and t.dim() == 2):
            return False
        if not t.is_floating_point() and t.dim() != 1:
            return False
    if isinstance(x, tuple) and (x_l is None or x_r is None):
        return False
    return (isinstance(conv.aggr, str) and conv.aggr in ('mean', 'add', 'sum', 'max', 'min')
            and conv.flow == 'source_to_target' and _quiet(conv)
            and not torch.is_autocast_enabled()
            and not _capturing(w))


def _note_segment_matmul_heuristic(conv) -> None:
    """The reference's forward leaves `_use_segment_matmul_heuristic_output` on the layer
    (rgcn_conv.py:246-260: a bool once a forward has run with `backend.use_segment_matmul = None`);
    TorchScript types the attribute from that value (`torch.jit.script(conv)` after a forward:
    test_rgcn_conv.py:84).  Same inputs — the relation histogram's maximum comes from the cached
    handle — same helper, once per handle."""
    import torch_geometric.backend as pyg_backend
    if pyg_backend.use_segment_matmul is not None:
        return
    hit = conv.__dict__.get('_handle_cache')
    handle = hit[-1] if hit is not None else None
    if handle is None or getattr(handle, '_heuristic_for', None) is conv:
        return
    conv._use_segment_matmul_heuristic_output = pyg_backend.use_segment_matmul_heuristic(
        num_segments=conv.num_relations, max_segment_size=handle.max_edges_per_relation,
        in_channels=conv.weight.size(1), out_channels=conv.weight.size(2))
    handle._heuristic_for = conv


def _wrap_rgcn_forward(cls, fast: bool) -> Callable:
    """The reference's ``RGCNConv.forward`` — a Python loop of masked ``propagate`` calls, 474
    iterations at the FB15k-237 shape (nn/conv/rgcn_conv.py:243-282) — and ``FastRGCNConv.forward``
    (one ``bmm`` over per-edge weight copies, rgcn_conv.py:302-374) on this package's sorted,
    segmented schedule (``nn/conv/rgcn_conv.py``: one sort per graph, one SpMM per
    (relation, destination) pair segment, one grouped fp32-MFMA GEMM = the ``segment_matmul``
    of seam S2, one SpMM back onto the nodes).  Same parameters, same result as the loop (the
    per-relation mean included); dense / ``num_bases`` / ``num_blocks`` weights, feature and
    node-index inputs.  Note the reference's OWN ``segment_matmul`` branch (rgcn_conv.py:264-271,
    only with pyg-lib) normalises ``aggr='mean'`` over all relations together (its TODO at
    :286): this route follows the loop, which is what the reference computes on the CPU."""
    orig = cls.forward
    names = ('message', 'aggregate', 'update', 'message_and_aggregate')

    def forward(self, x, edge_index, edge_type=None):
        if (_enabled() and edge_type is not None and _stock(self, cls, names)
                and _rgcn_args_ok(self, x, edge_index, edge_type)
                and getattr(self, 'fuse_relations', True)):
            from .nn.conv import rgcn_conv as own
            ei = _plain_index(edge_index)
            run = own.fast_rgcn_forward if fast else own.rgcn_forward
            out = run(self, x, ei, edge_type)
            if not fast and self.num_blocks is None:
                _note_segment_matmul_heuristic(self)
            return out
        return orig(self, x, edge_index, edge_type)

    return _stand_in(forward, orig)


def _wrap_heterolinear_forward(cls) -> Callable:
    """The reference's ``HeteroLinear.forward`` (nn/dense/linear.py:287-329: a Python loop of
    per-type ``matmul`` calls unless pyg-lib's ``segment_matmul`` is importable) as sort by type ->
    ONE grouped fp32-MFMA GEMM -> per-type bias -> original order, for float32 device rows."""
    orig = cls.forward

    def forward(self, x, type_vec):
        w = self.weight
        if (_enabled() and _ours(x) and x.dim() == 2 and x.size(0) > 0
                and not isinstance(w, torch.nn.parameter.UninitializedParameter) and _ours(w)
                and isinstance(type_vec, Tensor) and type_vec.is_cuda and type_vec.dim() == 1
                and type_vec.dtype in (torch.int32, torch.int64)
                and type_vec.numel() == x.size(0) and x.size(1) == w.size(1)
                and not torch.is_autocast_enabled()
                and not _capturing(x)):
            from .nn.dense.linear import hetero_linear_forward
            return hetero_linear_forward(self, x, type_vec)
        return orig(self, x, type_vec)

    return _stand_in(forward, orig)


class _SegmentMatmulOps:
    """``pyg_lib.ops`` as far as seam S2 goes: ``segment_matmul(inputs, ptr,
    other)`` with pyg-lib's contract (call sites nn/conv/rgcn_conv.py:288, nn/dense/linear.py:255)
    on the grouped fp32-MFMA GEMM.  Device float32 only — there is no host computation here."""

    @staticmethod
    def segment_matmul(inputs, ptr, other):
        from .utils import segment_matmul
        if not (_ours(inputs) and _ours(other)):
            raise NotImplementedError(
                "pytorch_geometric_amd serves 'segment_matmul' for float32 HIP tensors only "
                f"(got {inputs.dtype} on {inputs.device})")
        return segment_matmul(inputs, ptr, other)


class _PygLibShim:
    ops = _SegmentMatmulOps


def _bind_segment_matmul(mods) -> List[Tuple[Any, str, Any]]:
    """Binds the ``pyg_lib`` NAME the reference's two call sites resolve at call time
    (``pyg_lib.ops.segment_matmul``: rgcn_conv.py:13-19,288; linear.py:16,255) when pyg-lib itself
    is absent.  The flag that sends the reference there, ``torch_geometric.typing.WITH_SEGMM``
    (typing.py:47-63), stays as it is: it is process-global and read for CPU tensors as well,
    which this backend does not compute — the layer wrappers above take the float32 device calls
    before the flag is looked at; a user who sets it gets this kernel from the reference's own
    branches."""
    import torch_geometric.typing as pyg_typing
    if getattr(pyg_typing, 'WITH_PYG_LIB', False):
        return []
    done = []
    for mod in mods:
        if hasattr(mod, 'pyg_lib'):
            done.append((mod, 'pyg_lib', mod.pyg_lib))
            mod.pyg_lib = _PygLibShim
    return done


_sampler_cls = None


def neighbor_sampler(data, num_neighbors: List[int], seed: int = 0, replace: bool = False,
                     disjoint: bool = False, subgraph_type='directional'):
    """A ``torch_geometric.sampler.BaseSampler`` (sampler/base.py:932-998) whose
    ``sample_from_nodes(NodeSamplerInput) -> SamplerOutput`` runs on the GPU
    (:class:`pytorch_geometric_amd.sampler.NeighborSampler`), so that the reference's
    ``NodeLoader(data, node_sampler=...)`` (loader/node_loader.py:90-152) drives it unchanged and
    joins the features with its own ``filter_data``.  ``data``: a ``torch_geometric.data.Data`` on
    the device (``edge_index``, ``num_nodes``) or a ``(edge_index, num_nodes)`` pair.  ``replace``,
    ``disjoint`` and ``subgraph_type`` (``'directional'`` | ``'bidirectional'``, string or the
    reference's ``SubgraphType``) are the options ``NeighborLoader`` forwards to its sampler
    (loader/neighbor_loader.py:209-233)."""
    global _sampler_cls
    import torch_geometric.sampler as pyg_sampler
    from .sampler import NeighborSampler
    if _sampler_cls is None:

        class MI355XNeighborSampler(pyg_sampler.BaseSampler):
            def __init__(self, edge_index, num_nodes, num_neighbors, seed=0, replace=False,
                         disjoint=False, subgraph_type='directional'):
                self.impl = NeighborSampler(edge_index, num_nodes, num_neighbors, seed=seed,
                                            output_cls=pyg_sampler.SamplerOutput,
                                            replace=replace, disjoint=disjoint,
                                            subgraph_type=subgraph_type)
                self.num_neighbors = list(num_neighbors)
                self.replace, self.disjoint = self.impl.replace, self.impl.disjoint
                self.subgraph_type = self.impl.subgraph_type

            def sample_from_nodes(self, index, **kwargs):
                return self.impl.sample_from_nodes(index, **kwargs)

            def sample_from_edges(self, index, neg_sampling=None):
                raise NotImplementedError('link-level sampling is out of scope (SURVEY.md §8)')

            @property
            def edge_permutation(self):
                return None  # `edge` already indexes the caller's edge_index

        _sampler_cls = MI355XNeighborSampler
    if isinstance(data, (tuple, list)):
        edge_index, num_nodes = data
    else:
        edge_index, num_nodes = data.edge_index, data.num_nodes
    if not (isinstance(edge_index, Tensor) and edge_index.is_cuda):
        raise ValueError("the sampler needs 'edge_index' on the HIP device (there is no CPU "
                         "fallback): move the data with `.to('cuda')` first")
    return _sampler_cls(edge_index, int(num_nodes), num_neighbors, seed, replace, disjoint,
                        subgraph_type)


def _wrap_linear_forward(cls):
    orig = cls.forward

    def forward(self, x):
        from ._functions import linear, own_linear_eligible
        w = self.weight
        if (_enabled() and isinstance(x, Tensor)
                and not isinstance(w, torch.nn.parameter.UninitializedParameter)
                and w.device == x.device and own_linear_eligible(x, w)):
            return linear(x, w, self.bias)   # (the C++ node for plain operands)
        return orig(self, x)

    return _stand_in(forward, orig)


def install() -> None:
    """Idempotent.  Needs ``torch_geometric`` importable; raises ImportError otherwise."""
    if _state['installed']:
        return
    import torch_geometric
    import torch_geometric.backend as pyg_backend
    import torch_geometric.nn  # noqa: F401  (make sure the conv modules are loaded before sweeping)
    import torch_geometric.utils as pyg_utils
    from torch_geometric.utils import _scatter as pyg_scatter

    orig = {
        'scatter': pyg_utils.scatter, 'segment': pyg_utils.segment,
        'segment_logsumexp': pyg_utils.segment_logsumexp,
        'softmax': pyg_utils.softmax, 'index_sort': pyg_utils.index_sort,
        'scatter_argmax': pyg_scatter.scatter_argmax, 'spmm': pyg_utils.spmm,
        'sort_edge_index': pyg_utils.sort_edge_index, 'coalesce': pyg_utils.coalesce,
    }
    new = _make_dispatchers(orig)
    for name in orig:
        _state['rebinds'] += _sweep(orig[name], new[name])

    import torch_geometric.edge_index as pyg_edge_index
    orig_spmm = pyg_edge_index._spmm
    pyg_edge_index._spmm = _make_edge_index_spmm(orig_spmm)
    _state['rebinds'].append((pyg_edge_index, '_spmm', orig_spmm))

    from torch_geometric.nn.conv import GATConv, GCNConv, GraphConv, SAGEConv
    for cls in (SAGEConv, GCNConv, GraphConv, GATConv):
        had_own = 'propagate' in cls.__dict__
        prev = cls.__dict__.get('propagate')
        cls.propagate = _wrap_propagate(cls)
        _state['classes'].append((cls, had_own, prev))

    # A layer object renders its TorchScript-able `propagate` from a template when it is BUILT
    # (`_set_jittable_templates`, message_passing.py:926-1000) — unless its class carries a custom
    # `propagate`, which the stand-in above would look like.  So for the wrapped classes that step
    # runs against the class as the reference left it, and the stand-in goes back on top of
    # whatever it produced (the generated function: its module name and globals are the stand-in's
    # too, so later objects skip the rendering and TorchScript compiles the generated body).
    from torch_geometric.nn.conv import MessagePassing as _MP
    orig_templates = _MP.__dict__['_set_jittable_templates']

    def _set_jittable_templates(self, raise_on_error=False):
        cls = type(self)
        at = next((i for i, r in enumerate(_state['classes']) if r[0] is cls), None)
        if at is None or not getattr(cls.__dict__.get('propagate'), '_pygamd_propagate', False):
            return orig_templates(self, raise_on_error)
        _, had_own, prev = _state['classes'][at]
        if had_own:
            cls.propagate = prev
        else:
            delattr(cls, 'propagate')
        try:
            return orig_templates(self, raise_on_error)
        finally:
            had_now, now = 'propagate' in cls.__dict__, cls.__dict__.get('propagate')
            cls.propagate = _wrap_propagate(cls)
            _state['classes'][at] = (cls, had_now, now)

    _MP._set_jittable_templates = _stand_in(_set_jittable_templates, orig_templates)
    _state['rebinds'].append((_MP, '_set_jittable_templates', orig_templates))

    import torch_geometric.nn.conv.gat_conv as pyg_gat_mod
    import torch_geometric.nn.conv.gcn_conv as pyg_gcn_mod
    _state['rebinds'] += _make_graph_rewrite_memos(pyg_gcn_mod, pyg_gat_mod)

    from torch_geometric.nn.models import GraphSAGE
    had_own = 'forward' in GraphSAGE.__dict__
    prev = GraphSAGE.__dict__.get('forward')
    GraphSAGE.forward = _wrap_graphsage_forward(GraphSAGE)
    _state['forwards'].append((GraphSAGE, had_own, prev))
    # ... and a single SAGEConv layer of any model
    had_own = 'forward' in SAGEConv.__dict__
    prev = SAGEConv.__dict__.get('forward')
    SAGEConv.forward = _wrap_sageconv_forward(SAGEConv)
    _state['forwards'].append((SAGEConv, had_own, prev))
    had_own = 'forward' in GraphConv.__dict__
    prev = GraphConv.__dict__.get('forward')
    GraphConv.forward = _wrap_graphconv_forward(GraphConv)
    _state['forwards'].append((GraphConv, had_own, prev))
    # ... and GCNConv aggregates before it transforms when its input is the narrower side
    had_own = 'forward' in GCNConv.__dict__
    prev = GCNConv.__dict__.get('forward')
    GCNConv.forward = _wrap_gcnconv_forward(GCNConv)
    _state['forwards'].append((GCNConv, had_own, prev))

    # ... GATConv as projection + one attention node
    had_own = 'forward' in GATConv.__dict__
    prev = GATConv.__dict__.get('forward')
    GATConv.forward = _wrap_gatconv_forward(GATConv)
    _state['forwards'].append((GATConv, had_own, prev))
    # ... the relational layers on the sorted, segmented schedule (BASELINE config 5), HeteroLinear
    # as one grouped GEMM, and the `pyg_lib.ops.segment_matmul` name of seam S2
    from torch_geometric.nn.conv import FastRGCNConv, RGCNConv
    from torch_geometric.nn.dense.linear import HeteroLinear as PygHeteroLinear
    for kls, wrap in ((RGCNConv, _wrap_rgcn_forward(RGCNConv, False)),
                      (FastRGCNConv, _wrap_rgcn_forward(FastRGCNConv, True)),
                      (PygHeteroLinear, _wrap_heterolinear_forward(PygHeteroLinear))):
        had_own = 'forward' in kls.__dict__
        prev = kls.__dict__.get('forward')
        kls.forward = wrap
        _state['forwards'].append((kls, had_own, prev))
    import torch_geometric.nn.conv.rgcn_conv as pyg_rgcn_mod
    import torch_geometric.nn.dense.linear as pyg_linear_mod
    _state['rebinds'] += _bind_segment_matmul((pyg_rgcn_mod, pyg_linear_mod))

    # the gather of the general route: every `x_j` / `x_i` / `alpha_j` of any MessagePassing layer
    from torch_geometric.nn.conv import MessagePassing as PygMessagePassing
    orig_select = PygMessagePassing.__dict__['_index_select']
    PygMessagePassing._index_select = _make_index_select(orig_select)
    _state['rebinds'].append((PygMessagePassing, '_index_select', orig_select))

    # the reference's own dense layer (nn/dense/linear.py:121-127: F.linear) on the fp32-MFMA
    # kernels for float32 device inputs of >= OWN_GEMM_MIN_ROWS rows; everything else unchanged
    from torch_geometric.nn.dense.linear import Linear as PygLinear
    had_own = 'forward' in PygLinear.__dict__
    prev = PygLinear.__dict__.get('forward')
    PygLinear.forward = _wrap_linear_forward(PygLinear)
    _state['forwards'].append((PygLinear, had_own, prev))

    # seam S2: the operator names the reference calls when torch-sparse is importable
    # (edge_index.py:1798-1810) — defined by this backend only when that extension is absent
    from . import torch_sparse_ops
    torch_sparse_ops.register()

    pyg_backend.mi355x = sys.modules[__name__]
    if not hasattr(pyg_backend, 'use_mi355x'):
        pyg_backend.use_mi355x = None  # None = auto (on for float32 HIP tensors)
    _state['installed'] = True


def uninstall() -> None:
    if not _state['installed']:
        return
    import torch_geometric.backend as pyg_backend
    for mod, attr, old in _state['rebinds']:
        setattr(mod, attr, old)
    for cls, had_own, prev in _state['classes']:
        if had_own:
            cls.propagate = prev
        else:
            try:
                delattr(cls, 'propagate')
            except AttributeError:  # pragma: no cover
                pass
    for cls, had_own, prev in _state['forwards']:
        if had_own:
            cls.forward = prev
        else:
            try:
                delattr(cls, 'forward')
            except AttributeError:  # pragma: no cover
                pass
    for attr in ('mi355x', 'use_mi355x'):
        if hasattr(pyg_backend, attr):
            delattr(pyg_backend, attr)
    try:  # the stand-ins' copies of TorchScript overload declarations (_share_script_overloads)
        import torch._jit_internal as ji
        while _overload_names:
            ji._overloaded_fns.pop(_overload_names.pop(), None)
    except Exception:  # pragma: no cover
        pass
    _state.update(installed=False, rebinds=[], classes=[], forwards=[])


def is_installed() -> bool:
    return bool(_state['installed'])
