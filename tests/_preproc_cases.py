"""Shared checks of sort_edge_index / coalesce / to_undirected / is_undirected against the
goldens of the real reference (tests/golden/make_golden_preproc.py).  `ns` is the namespace under
test (the oracle on CPU, the HIP-backed utils on the GPU); `to` moves a tensor to its device."""
import torch

from tests._util import assert_close


def _eq(got, want, what):
    assert got.dtype == want.dtype, (what, got.dtype, want.dtype)
    assert torch.equal(got.cpu(), want), what


def check_sort_edge_index(ns, S, to):
    n = S['num_nodes']
    for by_row in (True, False):
        want = S[f'simple_by_row={by_row}']
        ei, (af, ai) = ns.sort_edge_index(to(S['simple']), [to(S['attr_f']), to(S['attr_i'])], n,
                                          sort_by_row=by_row)
        _eq(ei, want['edge_index'], 'sorted edge_index')
        _eq(af, want['attr_f'], 'float attr follows its edge')
        _eq(ai, want['attr_i'], 'int attr follows its edge')
        out = ns.sort_edge_index(to(S['dup']), None, n, sort_by_row=by_row)
        _eq(out[0], S[f'dup_by_row={by_row}'], 'duplicates: sorted keys')
    out = ns.sort_edge_index(to(S['simple']), None)
    _eq(out[0], S['infer_num_nodes'], 'num_nodes inferred')


def check_coalesce(ns, C, to, atol=1e-5):
    n = C['num_nodes']
    dup, attr, w = to(C['dup']), to(C['attr']), to(C['w'])
    for red in ['sum', 'mean', 'min', 'max', 'mul']:
        ei, a = ns.coalesce(dup, attr, n, reduce=red)
        _eq(ei, C[red]['edge_index'], f'coalesce[{red}] edge_index')
        assert_close(a, C[red]['attr'], atol=atol, what=f'coalesce[{red}] attr')
    ei, (a, ww) = ns.coalesce(dup, [attr, w], n, reduce='sum', sort_by_row=False)
    _eq(ei, C['list_by_col']['edge_index'], 'by column')
    assert_close(a, C['list_by_col']['attr'], atol=atol, what='list attr 0')
    assert_close(ww, C['list_by_col']['w'], atol=atol, what='list attr 1')
    ei, a = ns.coalesce(dup, None, n)
    _eq(ei, C['none_attr']['edge_index'], 'attr=None')
    assert a is None
    s = C['is_sorted']
    ei, a = ns.coalesce(to(s['in_edge_index']), to(s['in_attr']), n, reduce='sum',
                        is_sorted=True)
    _eq(ei, s['edge_index'], 'is_sorted edge_index')
    assert_close(a, s['attr'], atol=atol, what='is_sorted attr')


def check_undirected(ns, U, to, atol=1e-5):
    n = U['num_nodes']
    ei_in, attr = to(U['edge_index']), to(U['attr'])
    for red in ['add', 'mean', 'max']:
        ei, a = ns.to_undirected(ei_in, attr, n, reduce=red)
        _eq(ei, U[red]['edge_index'], f'to_undirected[{red}]')
        assert_close(a, U[red]['attr'], atol=atol, what=f'to_undirected[{red}] attr')
    flags = U['is_undirected']
    und, und_attr = to(U['add']['edge_index']), to(U['add']['attr'])
    assert ns.is_undirected(ei_in, None, n) == flags['directed'] is False
    assert ns.is_undirected(und, None, n) == flags['undirected'] is True
    assert ns.is_undirected(und, und_attr, n) == flags['undirected_attr'] is True
    bad = to(torch.arange(und.size(1)).float())
    assert ns.is_undirected(und, bad, n) == flags['undirected_bad_attr'] is False
