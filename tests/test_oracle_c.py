"""The plain C restatement (oracle/scatter_oracle.c) agrees with the torch restatement and with the
reference's golden vectors — two independent statements of the same CPU scatter path."""
import torch

from oracle import c_oracle as C
from oracle import pyg_oracle as O


def close(a, b, atol=1e-5):
    assert a.shape == b.shape
    assert torch.allclose(a, b, atol=atol, rtol=1e-5), (a - b).abs().max()


def test_integer_functions(golden):
    ix = golden['index']
    s, p = C.index_sort(ix['keys'], 37)
    assert torch.equal(s, ix['sorted']) and torch.equal(p, ix['perm'])
    assert torch.equal(C.index2ptr(s, 40), ix['ptr'])
    assert torch.equal(C.ptr2index(ix['ptr']), ix['ptr2index'])
    assert C.index2ptr(torch.tensor([0, 1, 1, 2]), 3).tolist() == [0, 1, 3, 4]


def test_scatter_segment_softmax(golden):
    sc = golden['scatter']
    for red in ['sum', 'mean', 'min', 'max', 'mul']:
        close(C.scatter(sc['src'], sc['index'], sc['dim_size'], red), sc[red]['out'])
    sg = golden['segment']
    for red in ['sum', 'mean', 'min', 'max']:
        close(C.segment(sg['src'], sg['ptr'], red), sg[red]['out'])
    sm = golden['softmax']
    assert C.softmax(sm['known']['src'].view(-1, 1), sm['known']['index'], 3).view(-1).tolist() \
        == [0.5, 0.5, 1, 1]
    close(C.softmax(sm['index']['src'], sm['index']['index'], 11), sm['index']['out'], 1e-6)
    close(C.softmax(sm['unsorted']['src'], sm['unsorted']['index'], 11), sm['unsorted']['out'],
          1e-6)


def test_propagate_and_sage_conv(golden):
    gr, L = golden['graph'], golden['layers']
    ei, x = gr['edge_index'], gr['x']
    for red in ['sum', 'mean', 'min', 'max']:
        close(C.propagate(x, ei, gr['N'], red), O.propagate(x, ei, gr['N'], red))
    close(C.propagate(x, ei, gr['N'], 'sum', gr['edge_weight']),
          O.propagate(x, ei, gr['N'], 'sum', gr['edge_weight']))
    st = L['sage_mean']['state']
    close(C.sage_conv(x, ei, st['lin_l.weight'], st['lin_l.bias'], st['lin_r.weight'], 'mean'),
          L['sage_mean']['out'], 2e-5)
    st = L['sage_max']['state']
    close(C.sage_conv(x, ei, st['lin_l.weight'], st['lin_l.bias'], st['lin_r.weight'], 'max'),
          L['sage_max']['out'], 2e-5)
    st = L['sage_sum_noroot']['state']
    close(C.sage_conv(x, ei, st['lin_l.weight'], None, None, 'sum'),
          L['sage_sum_noroot']['out'], 2e-5)


def test_sort_edges_and_coalesce(golden_preproc):
    """C restatement of the integer part of sort_edge_index / coalesce vs the real reference."""
    S, Cg = golden_preproc['sort_edge_index'], golden_preproc['coalesce']
    n = S['num_nodes']
    for by_row in (True, False):
        want = S[f'simple_by_row={by_row}']
        ei, perm, _ = C.sort_edges(S['simple'], n, by_row)
        assert torch.equal(ei, want['edge_index'])
        assert torch.equal(S['attr_i'][perm], want['attr_i'])       # attributes follow the edges
        ei, _, _ = C.sort_edges(S['dup'], n, by_row)
        assert torch.equal(ei, S[f'dup_by_row={by_row}'])
    ei, _, group = C.sort_edges(Cg['dup'], n, True, dedup=True)
    assert torch.equal(ei, Cg['sum']['edge_index'])
    merged = torch.zeros(ei.size(1), Cg['attr'].size(1)).index_add_(0, group, Cg['attr'])
    assert torch.allclose(merged, Cg['sum']['attr'], atol=1e-5)
    ei, _, _ = C.sort_edges(Cg['dup'], n, False, dedup=True)
    assert torch.equal(ei, Cg['list_by_col']['edge_index'])
