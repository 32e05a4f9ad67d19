"""Pins oracle/pyg_oracle.py against the vectors the REAL reference produced
(tests/golden/make_golden.py) and against the reference's own known-answer tests."""
import torch

from oracle import pyg_oracle as O

ATOL = 1e-6


def close(a, b, atol=ATOL):
    assert a.shape == b.shape, (a.shape, b.shape)
    assert torch.allclose(a, b, atol=atol, rtol=1e-6), (a - b).abs().max()


def grad_of(fn, inputs, grad_out):
    leaves = [t.clone().requires_grad_(True) if t.is_floating_point() else t for t in inputs]
    out = fn(*leaves)
    fl = [t for t in leaves if t.is_floating_point()]
    return out.detach(), torch.autograd.grad(out, fl, grad_out, allow_unused=True)


def test_scatter_values_and_grads(golden):
    sc = golden['scatter']
    for red in ['sum', 'mean', 'min', 'max', 'mul']:
        out, (gs, ) = grad_of(lambda s: O.scatter(s, sc['index'], 0, sc['dim_size'], red),
                              [sc['src']], sc[red]['grad_out'])
        close(out, sc[red]['out'])
        close(gs, sc[red]['grad_src'])
    d = sc['dim1_mean']
    out, (gs, ) = grad_of(lambda s: O.scatter(s, d['index'], 1, 6, 'mean'), [d['src']],
                          d['grad_out'])
    close(out, d['out'])
    close(gs, d['grad_src'])
    v = sc['vec_sum']
    close(O.scatter(v['src'], v['index'], 0, None, 'sum'), v['out'])


def test_scatter_empty_groups_are_zero(golden):
    sc = golden['scatter']
    for red in ['sum', 'mean', 'min', 'max']:
        out = O.scatter(sc['src'], sc['index'], 0, sc['dim_size'], red)
        assert (out[5] == 0).all() and (out[12:] == 0).all()
    assert (O.scatter(sc['src'], sc['index'], 0, sc['dim_size'], 'mul')[12:] == 1).all()


def test_scatter_argmax_known_answer(golden):
    # test/utils/test_scatter.py:111-120
    k = golden['scatter']['argmax_known']
    assert O.scatter_argmax(k['src'], k['index'], dim_size=6).tolist() == [3, 5, 1, 4, 5, 5]
    assert torch.equal(O.scatter_argmax(k['src'], k['index'], dim_size=6), k['out'])
    r = golden['scatter']['argmax_rand']
    assert torch.equal(O.scatter_argmax(r['src'], r['index'], dim_size=r['dim_size']), r['out'])


def test_segment(golden):
    sg = golden['segment']
    for red in ['sum', 'mean', 'min', 'max']:
        out, (gs, ) = grad_of(lambda s: O.segment(s, sg['ptr'], red), [sg['src']],
                              sg[red]['grad_out'])
        close(out, sg[red]['out'])
        close(gs, sg[red]['grad_src'])
        assert (out[0] == 0).all()  # empty first segment -> 0 (test/utils/test_segment.py:16-31)


def test_softmax(golden):
    sm = golden['softmax']
    k = sm['known']  # test/utils/test_softmax.py:12-24
    assert O.softmax(k['src'], k['index']).tolist() == [0.5, 0.5, 1, 1]
    assert O.softmax(k['src'], None, k['ptr']).tolist() == [0.5, 0.5, 1, 1]
    i = sm['index']
    out, (gs, ) = grad_of(lambda s: O.softmax(s, i['index'], num_nodes=11), [i['src']],
                          i['grad_out'])
    close(out, i['out'])
    close(gs, i['grad_src'])
    p = sm['ptr']
    out, (gs, ) = grad_of(lambda s: O.softmax(s, None, p['ptr']), [i['src']], p['grad_out'])
    close(out, p['out'])
    close(gs, p['grad_src'])
    u = sm['unsorted']
    out, (gs, ) = grad_of(lambda s: O.softmax(s, u['index'], num_nodes=11), [u['src']],
                          u['grad_out'])
    close(out, u['out'])
    close(gs, u['grad_src'])


def test_integer_goldens(golden):
    ix = golden['index']
    s, p = O.index_sort(ix['keys'], stable=True)
    assert torch.equal(s, ix['sorted']) and torch.equal(p, ix['perm'])
    assert torch.equal(O.index2ptr(s, 40), ix['ptr'])
    assert torch.equal(O.ptr2index(ix['ptr']), ix['ptr2index'])
    assert O.index2ptr(s.int(), 40).dtype == torch.int32
    assert torch.equal(O.index2ptr(s.int(), 40), ix['keys32_ptr'])
    k = ix['known_index2ptr']  # test/test_index.py:85-96
    assert O.index2ptr(k['index'], 3).tolist() == [0, 1, 3, 4]
    e = ix['known_edge_index']  # test/test_edge_index.py:196-233
    ei = e['edge_index']
    ptr, col, perm = O.csr_from_coo(ei[0], ei[1], 3)
    assert ptr.tolist() == [0, 1, 3, 4] == e['indptr'].tolist()
    tptr, trow, tperm = O.csr_from_coo(ei[1], ei[0], 3)
    assert tptr.tolist() == [0, 1, 3, 4] == e['T_indptr'].tolist()
    assert tperm.tolist() in ([1, 0, 3, 2], [1, 3, 0, 2])


def test_spmm_and_loops(golden):
    gr, sp, lp = golden['graph'], golden['spmm'], golden['loops']
    for red in ['sum', 'mean', 'min', 'max']:
        out = O.spmm(sp['edge_index'], gr['x'], gr['N'], red)
        close(out, sp[red]['out'], 1e-5)
    out, (gx, ) = grad_of(lambda x: O.spmm(sp['edge_index'], x, gr['N'], 'mean'), [gr['x']],
                          sp['mean']['grad_out'])
    close(gx, sp['mean']['grad_x'], 1e-5)
    ei, ew = O.add_remaining_self_loops(gr['edge_index'], gr['edge_weight'], 2.0, gr['N'])
    assert torch.equal(ei, lp['remaining_ei'])
    close(ew, lp['remaining_ew'])
    ei, ew = O.gcn_norm(gr['edge_index'], gr['edge_weight'], gr['N'])
    assert torch.equal(ei, lp['norm_ei'])
    close(ew, lp['norm_ew'])
    ei, ew = O.gcn_norm(gr['edge_index'], None, gr['N'])
    assert torch.equal(ei, lp['norm0_ei'])
    close(ew, lp['norm0_ew'])
    ei, _ = O.remove_self_loops(gr['edge_index'])
    ei, _ = O.add_self_loops(ei, num_nodes=gr['N'])
    assert torch.equal(ei, lp['gat_ei'])


def _check_layer(case, fn, names):
    params = [case['state'][n] for n in names]
    gr_out = case['grad_out']

    def run(x, *ps):
        return fn(x, *ps)

    leaves = [case['x'].clone().requires_grad_(True)] + [p.clone().requires_grad_(True)
                                                          for p in params]
    out = run(*leaves)
    close(out, case['out'], 1e-5)
    grads = torch.autograd.grad(out, leaves, gr_out, allow_unused=True)
    close(grads[0], case['grad_x'], 1e-5)
    for n, gp in zip(names, grads[1:]):
        ref = case['grad_params'][n]
        if ref is not None:
            close(gp, ref, 1e-4)


def test_layers(golden):
    gr, L = golden['graph'], golden['layers']
    ei, et, ew = gr['edge_index'], gr['edge_type'], gr['edge_weight']
    for c in L.values():
        c['x'] = gr['x']
    _check_layer(L['sage_mean'], lambda x, wl, bl, wr: O.sage_conv(x, ei, wl, bl, wr, 'mean'),
                 ['lin_l.weight', 'lin_l.bias', 'lin_r.weight'])
    _check_layer(L['sage_max'], lambda x, wl, bl, wr: O.sage_conv(x, ei, wl, bl, wr, 'max'),
                 ['lin_l.weight', 'lin_l.bias', 'lin_r.weight'])
    _check_layer(L['sage_sum_noroot'], lambda x, wl: O.sage_conv(x, ei, wl, None, None, 'sum'),
                 ['lin_l.weight'])
    _check_layer(L['gcn'], lambda x, b, w: O.gcn_conv(x, ei, w, b), ['bias', 'lin.weight'])
    _check_layer(L['gcn_weighted'], lambda x, b, w: O.gcn_conv(x, ei, w, b, ew),
                 ['bias', 'lin.weight'])
    _check_layer(L['gcn_nonorm'], lambda x, b, w: O.gcn_conv(x, ei, w, b, ew, normalize=False),
                 ['bias', 'lin.weight'])
    gat_names = ['att_src', 'att_dst', 'bias', 'lin.weight']
    _check_layer(L['gat'], lambda x, a_s, a_d, b, w: O.gat_conv(x, ei, w, a_s, a_d, b, 4, 6),
                 gat_names)
    _check_layer(L['gat_mean_heads'],
                 lambda x, a_s, a_d, b, w: O.gat_conv(x, ei, w, a_s, a_d, b, 4, 6, concat=False),
                 gat_names)
    _check_layer(L['gat_noloops'],
                 lambda x, a_s, a_d, b, w: O.gat_conv(x, ei, w, a_s, a_d, b, 2, 6,
                                                      add_self_loops_=False), gat_names)
    gc = ['lin_rel.weight', 'lin_rel.bias', 'lin_root.weight']
    _check_layer(L['graph_add_weighted'],
                 lambda x, w, b, r: O.graph_conv(x, ei, w, b, r, ew, 'add'), gc)
    _check_layer(L['graph_mean'], lambda x, w, b, r: O.graph_conv(x, ei, w, b, r, None, 'mean'),
                 gc)
    _check_layer(L['graph_max'], lambda x, w, b, r: O.graph_conv(x, ei, w, b, r, None, 'max'), gc)
    _check_layer(L['rgcn'], lambda x, w, r, b: O.rgcn_conv(x, ei, et, w, r, b),
                 ['weight', 'root', 'bias'])
    _check_layer(L['rgcn_blocks'], lambda x, w, r, b: O.rgcn_conv_blocks(x, ei, et, w, r, b),
                 ['weight', 'root', 'bias'])


def test_gat_attention(golden):
    gr, ga = golden['graph'], golden['gat_attention']
    st = ga['state']
    out, ei, alpha = O.gat_conv(gr['x'], gr['edge_index'], st['lin.weight'], st['att_src'],
                                st['att_dst'], st['bias'], 4, 6, return_alpha=True)
    close(out, ga['out'], 1e-5)
    assert torch.equal(ei, ga['edge_index'])
    close(alpha, ga['alpha'], 1e-6)


def test_models(golden):
    gr, M = golden['graph'], golden['models']
    ei, x = gr['edge_index'], gr['x']
    st = M['graphsage']['state']
    params = [(st[f'convs.{i}.lin_l.weight'], st[f'convs.{i}.lin_l.bias'],
               st[f'convs.{i}.lin_r.weight']) for i in range(3)]
    xx = x.clone().requires_grad_(True)
    out = O.graphsage(xx, ei, params)
    close(out, M['graphsage']['out'], 1e-5)
    (gx, ) = torch.autograd.grad(out, [xx], M['graphsage']['grad_out'])
    close(gx, M['graphsage']['grad_x'], 1e-5)
    st = M['gcn']['state']
    params = [(st[f'convs.{i}.lin.weight'], st[f'convs.{i}.bias']) for i in range(2)]
    close(O.gcn(x, ei, params), M['gcn']['out'], 1e-5)
    st = M['gat']['state']
    params = [(st[f'convs.{i}.lin.weight'], st[f'convs.{i}.att_src'], st[f'convs.{i}.att_dst'],
               st[f'convs.{i}.bias']) for i in range(3)]
    close(O.gat(x, ei, params, heads=4), M['gat']['out'], 1e-5)


# ---- preprocessing either side of the path -------------------------------------
def test_sort_edge_index_coalesce_undirected_match_the_reference(golden_preproc):
    from tests import _preproc_cases as P
    P.check_sort_edge_index(O, golden_preproc['sort_edge_index'], lambda t: t)
    P.check_coalesce(O, golden_preproc['coalesce'], lambda t: t)
    P.check_undirected(O, golden_preproc['undirected'], lambda t: t)
    # reference docstring examples (utils/_coalesce.py:107-129, _sort_edge_index.py:92-101)
    ei = torch.tensor([[1, 1, 2, 3], [3, 3, 1, 2]])
    out, w = O.coalesce(ei, torch.ones(4))
    assert out.tolist() == [[1, 2, 3], [3, 1, 2]] and w.tolist() == [2., 1., 1.]
    assert O.coalesce(ei, None, sort_by_row=False)[0].tolist() == [[2, 3, 1], [1, 2, 3]]
    ei = torch.tensor([[2, 1, 1, 0], [1, 2, 0, 1]])
    out, a = O.sort_edge_index(ei, torch.tensor([[1], [2], [3], [4]]))
    assert out.tolist() == [[0, 1, 1, 2], [1, 0, 2, 1]] and a.view(-1).tolist() == [4, 3, 2, 1]


def test_fast_rgcn_and_index_inputs(golden, golden_rgcn):
    """FastRGCNConv == RGCNConv's math on float inputs; node-index inputs of both classes."""
    gr, L = golden['graph'], golden_rgcn['layers']
    ei, et, x_idx = gr['edge_index'], gr['edge_type'], golden_rgcn['x_idx']
    N = gr['x'].size(0)
    for c in ('fast', 'fast_add', 'fast_blocks', 'fast_bases'):
        L[c]['x'] = gr['x']
    names = ['weight', 'root', 'bias']
    _check_layer(L['fast'], lambda x, w, r, b: O.rgcn_conv(x, ei, et, w, r, b), names)
    _check_layer(L['fast_add'], lambda x, w, r, b: O.rgcn_conv(x, ei, et, w, r, b, 'add'), names)
    _check_layer(L['fast_blocks'], lambda x, w, r, b: O.rgcn_conv_blocks(x, ei, et, w, r, b),
                 names)
    # the pair-row evaluation used at BASELINE config 5's timed shape (tests/test_gpu_configs.py):
    # against the reference's own results (dense and block weights, mean and add) ...
    _check_layer(L['fast'], lambda x, w, r, b: O.rgcn_conv_blocks_pairs(x, ei, et, w, r, b), names)
    _check_layer(L['fast_add'],
                 lambda x, w, r, b: O.rgcn_conv_blocks_pairs(x, ei, et, w, r, b, 'add'), names)
    _check_layer(L['fast_blocks'],
                 lambda x, w, r, b: O.rgcn_conv_blocks_pairs(x, ei, et, w, r, b), names)
    # ... and against the loop restatement, incl. max and relations without edges
    g = torch.Generator().manual_seed(3)
    n, R = 40, 9
    ei2 = torch.randint(0, n, (2, 300), generator=g)
    et2 = torch.randint(0, R - 2, (300, ), generator=g)
    x2 = torch.randn(n, 12, generator=g, dtype=torch.float64)
    w2 = torch.randn(R, 3, 4, 5, generator=g, dtype=torch.float64)
    r2 = torch.randn(12, 15, generator=g, dtype=torch.float64)
    for aggr in ('mean', 'add', 'max'):
        close(O.rgcn_conv_blocks_pairs(x2, ei2, et2, w2, r2, None, aggr),
              O.rgcn_conv_blocks(x2.float(), ei2, et2, w2.float(), r2.float(), None,
                                 aggr).double(), 1e-5)
    _check_layer(L['fast_bases'],
                 lambda x, w, c, r, b: O.rgcn_conv(
                     x, ei, et, O.rgcn_weight_from_bases(c, w, 16, 10), r, b),
                 ['weight', 'comp', 'root', 'bias'])

    def check_index(case, fn, names):
        leaves = [case['state'][n].clone().requires_grad_(True) for n in names]
        out = fn(*leaves)
        close(out, case['out'], 1e-5)
        grads = torch.autograd.grad(out, leaves, case['grad_out'], allow_unused=True)
        for n, g in zip(names, grads):
            if case['grad_params'][n] is not None:
                close(g, case['grad_params'][n], 1e-4)

    check_index(L['rgcn_index'], lambda w, r, b: O.rgcn_conv_index(x_idx, ei, et, w, r, b), names)
    check_index(L['rgcn_index_max'],
                lambda w, r, b: O.rgcn_conv_index(x_idx, ei, et, w, r, b, 'max'), names)
    ar = torch.arange(N)
    check_index(L['rgcn_none'], lambda w, r, b: O.rgcn_conv_index(ar, ei, et, w, r, b), names)
    check_index(L['fast_none'],
                lambda w, r, b: O.rgcn_conv_index(ar, ei, et, w, r, b, by_node_id=True), names)
    check_index(L['fast_none_bases'],
                lambda w, c, r, b: O.rgcn_conv_index(
                    ar, ei, et, O.rgcn_weight_from_bases(c, w, N, 10), r, b, by_node_id=True),
                ['weight', 'comp', 'root', 'bias'])


def test_softmax_and_powermean_aggregation(golden_aggr):
    from tests._aggr_cases import CASES
    G = golden_aggr
    for name, (kind, kw, use_ptr, positive) in CASES.items():
        case = G['cases'][name]
        x = (G['x'].abs() + 0.1 if positive else G['x']).clone().requires_grad_(True)
        where = dict(ptr=G['ptr']) if use_ptr else dict(index=G['index'], dim_size=G['dim_size'])
        value = kw.get('t', kw.get('p', 1.0))
        param = None
        if kw.get('learn'):
            param = torch.full((kw.get('channels', 1), ), value).requires_grad_(True)
            value = param.view(-1, kw['channels']) if kw.get('channels', 1) != 1 else param
        if kind == 'softmax':
            out = O.softmax_aggregation(x, t=value, semi_grad=kw.get('semi_grad', False), **where)
        else:
            out = O.powermean_aggregation(x, p=value, **where)
        close(out, case['out'], 1e-5)
        leaves = [x] + ([param] if param is not None else [])
        grads = torch.autograd.grad(out, leaves, case['grad_out'])
        close(grads[0], case['grad_x'], 1e-5)
        if param is not None:
            close(grads[1], case['grad_param'], 1e-4)
