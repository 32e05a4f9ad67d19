from .message_passing import MessagePassing
from .sage_conv import SAGEConv
from .gcn_conv import GCNConv, gcn_norm
from .gat_conv import GATConv
from .rgcn_conv import FastRGCNConv, RGCNConv
from .graph_conv import GraphConv

__all__ = ['MessagePassing', 'SAGEConv', 'GCNConv', 'gcn_norm', 'GATConv', 'RGCNConv', 'FastRGCNConv',
           'GraphConv']
