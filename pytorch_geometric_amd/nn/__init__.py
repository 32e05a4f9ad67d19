from .aggr import (Aggregation, FusedAggregation, MaxAggregation, MeanAggregation,
                   MinAggregation, MulAggregation, MultiAggregation, PowerMeanAggregation,
                   SoftmaxAggregation, StdAggregation, SumAggregation, VarAggregation)
from .conv import (FastRGCNConv, GATConv, GCNConv, GraphConv, MessagePassing, RGCNConv, SAGEConv,
                   gcn_norm)
from .dense import HeteroLinear, Linear
from .models import GAT, GCN, BasicGNN, GraphSAGE
from . import functional  # noqa: F401

__all__ = [
    'Aggregation', 'SumAggregation', 'MeanAggregation', 'MaxAggregation', 'MinAggregation',
    'MulAggregation', 'VarAggregation', 'StdAggregation', 'FusedAggregation',
    'MultiAggregation', 'SoftmaxAggregation', 'PowerMeanAggregation', 'MessagePassing', 'SAGEConv', 'GCNConv', 'gcn_norm', 'GATConv', 'RGCNConv', 'FastRGCNConv', 'GraphConv', 'Linear', 'HeteroLinear',
    'BasicGNN', 'GCN', 'GraphSAGE', 'GAT',
]
