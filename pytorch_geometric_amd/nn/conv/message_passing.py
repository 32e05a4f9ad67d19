import inspect
from typing import Any, Dict, List, Optional, Tuple, Union

import torch
from torch import Tensor

from ..._functions import GatherFunction
from ...edge_index import EdgeIndex, as_edge_index
from ..aggr import Aggregation, aggregation_resolver

_SPECIAL = {'edge_index', 'edge_index_i', 'edge_index_j', 'size', 'size_i', 'size_j', 'ptr',
            'index', 'dim_size'}
_INT_DTYPES = (torch.uint8, torch.int8, torch.int16, torch.int32, torch.int64)


def _params(fn) -> List[str]:
    return [n for n, p in inspect.signature(fn).parameters.items()
            if p.kind in (p.POSITIONAL_OR_KEYWORD, p.KEYWORD_ONLY)]


class MessagePassing(torch.nn.Module):
    r"""Base class of the message-passing layers:
    ``x_i' = update(x_i, aggr_{j in N(i)} message(x_i, x_j, e_ji))``.

    Same contract as ``torch_geometric.nn.conv.MessagePassing``
    (torch_geometric/nn/conv/message_passing.py:421-616): ``propagate(edge_index, size, **kwargs)``
    collects the ``*_i`` / ``*_j`` arguments of ``message`` by gathering rows with
    ``edge_index[i]`` / ``edge_index[j]`` (``flow='source_to_target'``: i = 1, j = 0), calls
    ``message``, reduces with ``aggregate`` onto ``edge_index[i]`` and calls ``update``.

    When a layer defines ``message_and_aggregate(graph, ...)`` and ``self.fuse`` is true, the three
    steps run as ONE CSR SpMM kernel on a cached, destination-sorted handle instead of
    materialising ``[E, F]`` messages (the reference only fuses for sparse ``adj_t`` inputs,
    message_passing.py:469-479; here every plain ``edge_index`` tensor is sorted once and cached).
    """

    def __init__(self, aggr: Optional[Union[str, Aggregation]] = 'sum', *,
                 aggr_kwargs: Optional[Dict[str, Any]] = None, flow: str = 'source_to_target',
                 node_dim: int = -2):
        super().__init__()
        if flow not in ('source_to_target', 'target_to_source'):
            raise ValueError(f"Expected 'flow' to be either 'source_to_target' or "
                             f"'target_to_source' (got '{flow}')")
        self.aggr = aggr if isinstance(aggr, str) or aggr is None else None
        self.aggr_module = (None if aggr is None else
                            aggregation_resolver(aggr, **(aggr_kwargs or {})))
        self.flow = flow
        self.node_dim = node_dim
        self._msg_params = _params(self.message)
        self._upd_params = _params(self.update)[1:]
        self._edge_params = _params(self.edge_update)
        self._propagate_forward_pre_hooks = {}
        self._propagate_forward_hooks = {}
        self._hook_id = 0
        has_fused = type(self).message_and_aggregate is not MessagePassing.message_and_aggregate
        self._fused_params = _params(self.message_and_aggregate)[1:] if has_fused else None
        self.fuse = has_fused

    def reset_parameters(self):
        if self.aggr_module is not None:
            self.aggr_module.reset_parameters()

    # -- validation (message_passing.py:204-259) ---------------------------------------------------
    def _check_input(self, edge_index, size) -> List[Optional[int]]:
        if isinstance(edge_index, EdgeIndex):
            return [edge_index.num_src_nodes, edge_index.num_dst_nodes]
        if isinstance(edge_index, Tensor):
            if edge_index.dtype not in _INT_DTYPES:
                raise ValueError(f"Expected 'edge_index' to be of integer "
                                 f"type (got '{edge_index.dtype}')")
            if edge_index.dim() != 2:
                raise ValueError(f"Expected 'edge_index' to be two-dimensional"
                                 f" (got {edge_index.dim()} dimensions)")
            if edge_index.size(0) != 2:
                raise ValueError(f"Expected 'edge_index' to have size '2' in "
                                 f"the first dimension (got "
                                 f"'{edge_index.size(0)}')")
            return list(size) if size is not None else [None, None]
        raise ValueError('`MessagePassing.propagate` only supports integer tensors of shape '
                         '`[2, num_messages]` or `EdgeIndex` handles for argument `edge_index`.')

    def _set_size(self, size: List[Optional[int]], dim: int, src: Tensor):
        the_size = size[dim]
        if the_size is None:
            size[dim] = src.size(self.node_dim)
        elif the_size != src.size(self.node_dim):
            raise ValueError(f'Encountered tensor with size {src.size(self.node_dim)} in '
                             f'dimension {self.node_dim}, but expected size {the_size}.')

    def _lift(self, src: Tensor, index: Tensor) -> Tensor:
        """``src.index_select(node_dim, index)`` with the reference's IndexError on bad indices
        (message_passing.py:263-290)."""
        d = self.node_dim + src.dim() if self.node_dim < 0 else self.node_dim
        s0 = src if d == 0 else src.movedim(d, 0).contiguous()
        try:
            out = GatherFunction.apply(s0, index, True)
        except IndexError as e:
            from ... import _native
            lo, hi = _native.index_minmax(index)
            n = src.size(self.node_dim)
            if lo < 0:
                raise IndexError(
                    f"Found negative indices in 'edge_index' (got {lo}). Please ensure that all "
                    f"indices in 'edge_index' point to valid indices in the interval [0, {n}) in "
                    f"your node feature matrix and try again.") from e
            raise IndexError(
                f"Found indices in 'edge_index' that are larger than {n - 1} (got {hi}). Please "
                f"ensure that all indices in 'edge_index' point to valid indices in the interval "
                f"[0, {n}) in your node feature matrix and try again.") from e
        return out if d == 0 else out.movedim(0, d)

    def _ij(self) -> Tuple[int, int]:
        return (1, 0) if self.flow == 'source_to_target' else (0, 1)

    def _raw(self, edge_index) -> Tensor:
        return edge_index.edge_index if isinstance(edge_index, EdgeIndex) else edge_index

    def _collect(self, args: List[str], edge_index, size: List[Optional[int]],
                 kwargs: Dict[str, Any], lift: bool = True) -> Dict[str, Any]:
        i, j = self._ij()
        ei = self._raw(edge_index)
        out: Dict[str, Any] = {}
        for arg in args:
            if arg in _SPECIAL:
                continue
            if arg[-2:] not in ('_i', '_j'):
                out[arg] = kwargs.get(arg, inspect.Parameter.empty)
                continue
            dim = j if arg[-2:] == '_j' else i
            data = kwargs.get(arg[:-2], inspect.Parameter.empty)
            if isinstance(data, (tuple, list)):
                assert len(data) == 2
                if isinstance(data[1 - dim], Tensor):
                    self._set_size(size, 1 - dim, data[1 - dim])
                data = data[dim]
            if isinstance(data, Tensor):
                self._set_size(size, dim, data)
                if lift:
                    data = self._lift(data, ei[dim])
            out[arg] = data
        out['edge_index'] = edge_index
        out['edge_index_i'], out['edge_index_j'] = ei[i], ei[j]
        out['ptr'] = None
        out['index'] = out['edge_index_i']
        out['size'] = size
        out['size_i'] = size[i] if size[i] is not None else size[j]
        out['size_j'] = size[j] if size[j] is not None else size[i]
        out['dim_size'] = out['size_i']
        return out

    @staticmethod
    def _select(names: List[str], coll: Dict[str, Any], what: str) -> Dict[str, Any]:
        sel = {}
        for n in names:
            v = coll.get(n, inspect.Parameter.empty)
            if v is inspect.Parameter.empty:
                raise TypeError(f"Required parameter '{n}' of '{what}' is empty")
            sel[n] = v
        return sel

    # -- hooks (message_passing.py:461-464, 558-561; seam S4 of SURVEY.md §8(b)) -------------------
    def _register(self, table: dict, hook):
        self._hook_id += 1
        key = self._hook_id
        table[key] = hook

        class _Handle:
            def remove(_self):
                table.pop(key, None)

        return _Handle()

    def register_propagate_forward_pre_hook(self, hook):
        """``hook(module, (edge_index, size, kwargs))`` may return a replacement triple; runs
        before anything else in ``propagate`` — e.g. to swap a raw ``edge_index`` for a handle."""
        return self._register(self._propagate_forward_pre_hooks, hook)

    def register_propagate_forward_hook(self, hook):
        """``hook(module, (edge_index, size, kwargs), output)`` may return a replacement output."""
        return self._register(self._propagate_forward_hooks, hook)

    # -- the hot path -----------------------------------------------------------------------------
    def propagate(self, edge_index, size: Optional[Tuple[int, int]] = None, **kwargs) -> Tensor:
        r"""Gather -> message -> aggregate -> update (message_passing.py:421-563)."""
        for hook in list(self._propagate_forward_pre_hooks.values()):
            res = hook(self, (edge_index, size, kwargs))
            if res is not None:
                edge_index, size, kwargs = res
        out = self._propagate(edge_index, size, kwargs)
        for hook in list(self._propagate_forward_hooks.values()):
            res = hook(self, (edge_index, size, kwargs), out)
            if res is not None:
                out = res
        return out

    def _propagate(self, edge_index, size, kwargs) -> Tensor:
        size = self._check_input(edge_index, size)
        if self.fuse and self._fused_params is not None and self._can_fuse(kwargs):
            coll = self._collect(self._msg_params, edge_index, size, kwargs, lift=False)
            i, j = self._ij()
            n_src = size[j] if size[j] is not None else size[i]
            n_dst = size[i] if size[i] is not None else size[j]
            graph = as_edge_index(edge_index, n_src, n_dst,
                                  flip=self.flow == 'target_to_source')
            fused = {n: kwargs.get(n) for n in self._fused_params}
            out = self.message_and_aggregate(graph, **fused)
        else:
            coll = self._collect(self._msg_params, edge_index, size, kwargs)
            msg = self.message(**self._select(self._msg_params, coll, 'message'))
            out = self.aggregate(msg, index=coll['index'], ptr=coll['ptr'],
                                 dim_size=coll['dim_size'])
        for n in self._upd_params:
            coll.setdefault(n, kwargs.get(n, inspect.Parameter.empty))
        return self.update(out, **self._select(self._upd_params, coll, 'update'))

    def _can_fuse(self, kwargs: Dict[str, Any]) -> bool:
        if self.flow != 'source_to_target' and any(isinstance(v, (tuple, list))
                                                   for v in kwargs.values()):
            return False  # (src, dst) pairs swap roles; take the general path
        if self.aggr in ('min', 'max') and kwargs.get('edge_weight') is not None:
            return False  # the SpMM kernels weight sums and means only: max_j (w_e x_j) takes
        return self.aggr in ('sum', 'add', 'mean', 'min', 'max')  # the gather -> message route

    def edge_updater(self, edge_index, size: Optional[Tuple[int, int]] = None, **kwargs):
        r"""Computes per-edge features via ``edge_update`` (message_passing.py:620-665)."""
        size = self._check_input(edge_index, size)
        coll = self._collect(self._edge_params, edge_index, size, kwargs)
        return self.edge_update(**self._select(self._edge_params, coll, 'edge_update'))

    def message(self, x_j: Tensor) -> Tensor:
        return x_j

    def aggregate(self, inputs: Tensor, index: Tensor, ptr: Optional[Tensor] = None,
                  dim_size: Optional[int] = None) -> Tensor:
        return self.aggr_module(inputs, index, ptr=ptr, dim_size=dim_size, dim=self.node_dim)

    def message_and_aggregate(self, graph: EdgeIndex) -> Tensor:
        raise NotImplementedError

    def update(self, inputs: Tensor) -> Tensor:
        return inputs

    def edge_update(self) -> Tensor:
        raise NotImplementedError

    def __repr__(self) -> str:
        if hasattr(self, 'in_channels') and hasattr(self, 'out_channels'):
            return f'{self.__class__.__name__}({self.in_channels}, {self.out_channels})'
        return f'{self.__class__.__name__}()'
