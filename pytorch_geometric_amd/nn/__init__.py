from .aggr import (Aggregation, MaxAggregation, MeanAggregation, MinAggregation, MulAggregation,
                   SumAggregation)
from .conv import GATConv, GCNConv, MessagePassing, SAGEConv, gcn_norm
from .dense import Linear
from .models import GAT, GCN, BasicGNN, GraphSAGE

__all__ = [
    'Aggregation', 'SumAggregation', 'MeanAggregation', 'MaxAggregation', 'MinAggregation',
    'MulAggregation', 'MessagePassing', 'SAGEConv', 'GCNConv', 'gcn_norm', 'GATConv', 'Linear',
    'BasicGNN', 'GCN', 'GraphSAGE', 'GAT',
]
