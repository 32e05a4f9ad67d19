from .basic_gnn import GAT, GCN, BasicGNN, GraphSAGE

__all__ = ['BasicGNN', 'GCN', 'GraphSAGE', 'GAT']
