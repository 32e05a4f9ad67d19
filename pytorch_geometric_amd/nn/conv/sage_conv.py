from typing import Optional, Tuple, Union

import torch.nn.functional as F
from torch import Tensor

from ..._functions import SpmmFunction, spmm_node
from ...edge_index import EdgeIndex
from ..dense.linear import Linear
from .message_passing import MessagePassing

PairOrTensor = Union[Tensor, Tuple[Tensor, Optional[Tensor]]]


class SAGEConv(MessagePassing):
    r"""GraphSAGE operator ``x_i' = W_r x_i + W_l * aggr_{j in N(i)} x_j`` with the constructor
    arguments, parameter names (``lin_l``, ``lin_r``, ``lin`` when ``project=True``) and forward
    semantics of ``torch_geometric.nn.SAGEConv`` (torch_geometric/nn/conv/sage_conv.py:68-152),
    so ``state_dict``s interchange.

    On a square graph with float32 device features the whole layer is ONE kernel and one autograd
    node (``nn/models/_fused_sage.py:layer_eligible``; ``fuse = False`` on the layer opts out).
    Otherwise (bipartite pairs, ``project=True``, hooks on ``propagate``, other dtypes) the
    neighbourhood reduction runs at the INPUT width as one CSR SpMM launch
    (``message_and_aggregate``) and the linear maps on the repo's GEMM kernels.
    """

    def __init__(self, in_channels: Union[int, Tuple[int, int]], out_channels: int,
                 aggr: str = 'mean', normalize: bool = False, root_weight: bool = True,
                 project: bool = False, bias: bool = True, **kwargs):
        super().__init__(aggr, **kwargs)
        dims = (in_channels, in_channels) if isinstance(in_channels, int) else tuple(in_channels)
        if min(dims) <= 0:
            raise ValueError(f"'{type(self).__name__}' needs explicit input sizes (lazy "
                             f"initialization is not supported)")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.normalize, self.root_weight, self.project = normalize, root_weight, project
        src_dim, dst_dim = dims
        if project:  # Eq. (3) of the paper: x_j <- relu(W x_j + b) before aggregating
            self.lin = Linear(src_dim, src_dim, bias=True)
        self.lin_l = Linear(src_dim, out_channels, bias=bias)
        if root_weight:
            self.lin_r = Linear(dst_dim, out_channels, bias=False)
        self.reset_parameters()

    def _linears(self):
        return [getattr(self, n) for n in ('lin', 'lin_l', 'lin_r') if hasattr(self, n)]

    def reset_parameters(self):
        super().reset_parameters()
        for lin in self._linears():
            lin.reset_parameters()

    def forward(self, x: PairOrTensor, edge_index,
                size: Optional[Tuple[int, int]] = None) -> Tensor:
        if isinstance(x, Tensor):
            from ..models import _fused_sage
            if _fused_sage.layer_eligible(self, x, edge_index, size):
                # aggregation, both linear maps and the bias in one kernel; one autograd node
                h = _fused_sage.run_layer(self, x, edge_index)
                return F.normalize(h, p=2.0, dim=-1) if self.normalize else h
        x_src, x_dst = (x, x) if isinstance(x, Tensor) else x
        if self.project:
            x_src = self.lin(x_src).relu()
        h = self.lin_l(self.propagate(edge_index, x=(x_src, x_dst), size=size))
        if self.root_weight and x_dst is not None:
            h = h + self.lin_r(x_dst)
        return F.normalize(h, p=2.0, dim=-1) if self.normalize else h

    def message(self, x_j: Tensor) -> Tensor:
        return x_j

    def message_and_aggregate(self, graph: EdgeIndex, x) -> Tensor:
        reduce = {'add': 'sum'}.get(self.aggr, self.aggr)
        return spmm_node(x[0], None, graph, reduce, 'coo')

    def __repr__(self) -> str:
        return (f'{type(self).__name__}({self.in_channels}, {self.out_channels}, '
                f'aggr={self.aggr})')
