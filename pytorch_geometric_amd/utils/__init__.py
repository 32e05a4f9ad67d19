from ._scatter import scatter, scatter_argmax
from ._segment import segment, segment_logsumexp
from ._softmax import softmax
from ._spmm import spmm
from ._index_sort import index_sort
from ._degree import degree
from .num_nodes import maybe_num_nodes
from ._trim_to_layer import trim_to_layer
from ._segment_matmul import segment_matmul
from ._sort_edge_index import coalesce, is_undirected, sort_edge_index, to_undirected
from .loop import (add_remaining_self_loops, add_self_loops, contains_self_loops,
                   remove_self_loops)

__all__ = [
    'scatter', 'scatter_argmax', 'segment', 'segment_logsumexp', 'softmax', 'spmm', 'index_sort', 'degree',
    'maybe_num_nodes', 'trim_to_layer', 'segment_matmul', 'add_remaining_self_loops', 'add_self_loops', 'contains_self_loops',
    'remove_self_loops', 'sort_edge_index', 'coalesce', 'to_undirected', 'is_undirected',
]
