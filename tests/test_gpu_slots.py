"""Static-shape ("slot") sampled batches (csrc/minibatch.hip, pytorch_geometric_amd/slots.py): the
sampling contract of the reference's neighbour sampler (sampler/neighbor_sampler.py:550-577: per
frontier node a uniform min(deg, k)-subset of its in-neighbours, without replacement) in the
padded-id layout of a captured training step, the transposed CSRs of the backward, and the GraphSAGE
stack on top (one kernel per layer forward; dgrad GEMMs + transposed SpMM backward) against a plain
PyTorch evaluation of the same sampled computation graph (trim_to_layer,
utils/_trim_to_layer.py:167-215)."""
import pytest
import torch

from tests._util import assert_close, assert_close_scaled, gen

pytestmark = pytest.mark.gpu


def _graph(n, e, seed, hubs=True):
    g = gen(seed)
    src = torch.randint(0, n, (e, ), generator=g)
    dst = torch.randint(0, n, (e, ), generator=g)
    if hubs:  # a few popular destinations (deg >> k) and popular sources (many duplicates)
        dst[:e // 10] = torch.randint(0, 5, (e // 10, ), generator=g)
        src[e // 10:e // 5] = torch.randint(5, 25, (e // 10, ), generator=g)
    key = torch.unique(dst * n + src)            # no parallel edges: neighbours are distinct nodes
    return torch.stack([key % n, key // n])


def _decode(p, b):
    """CPU view of a batch: per valid slot (slot, dst row, src row, graph src)."""
    src_g = b.src_g.cpu()
    slots = (src_g >= 0).nonzero().squeeze(1)
    hop = torch.zeros_like(slots)
    for h in range(p.L):
        hop[slots >= p.ebase(h)] = h
    eb = torch.tensor([p.ebase(h) for h in range(p.L)])
    nb = torch.tensor(p.bases[:p.L])
    fan = torch.tensor(p.fanouts)
    dst = nb[hop] + (slots - eb[hop]) // fan[hop]
    return slots, hop, dst, b.src_id.cpu().long()[slots], src_g[slots]


@pytest.mark.parametrize('fan,B', [([4, 3, 2], 64), ([15, 10, 5], 32), ([3], 100), ([64, 2], 7)])
def test_slot_sampler_contract(dev, fan, B):
    import pytorch_geometric_amd as pga
    from pytorch_geometric_amd.slots import SlotPlan, SlotSampler
    n = 4000
    ei = _graph(n, 60_000, seed=B + len(fan))
    csc = pga.EdgeIndex(ei.to(dev), (n, n)).by_dst()
    colptr, row = csc.ptr.cpu(), csc.idx.cpu()
    nbrs = {}
    plan = SlotPlan(B, fan, dev)
    smp = SlotSampler(csc.ptr, csc.idx, n, plan, seed=5)
    g = gen(17)
    epoch = torch.zeros(1, dtype=torch.int64, device=dev)
    first = None
    for it in range(4):   # the node map is never reset: later batches must be as clean as the first
        seeds = torch.randperm(n, generator=g)[:B]
        epoch += 1
        b = smp.sample(seeds.to(dev), epoch)
        torch.cuda.synchronize()
        node_g, src_g = b.node_g.cpu(), b.src_g.cpu()
        row_end, inv = b.row_end.cpu().long(), b.inv_cnt.cpu()
        begin = plan.row_begin.cpu().long()
        assert torch.equal(node_g[:B], seeds)
        valid = node_g >= 0
        # every node of the batch sits in exactly one row
        assert node_g[valid].unique().numel() == int(valid.sum())
        cnt = row_end - begin
        deg = (colptr[1:] - colptr[:-1])
        for b_ in range(plan.L):
            k = fan[b_]
            rows = torch.arange(plan.bases[b_], plan.bases[b_ + 1])
            v = node_g[rows]
            want = torch.where(v >= 0, torch.minimum(deg[v.clamp(min=0)], torch.tensor(k)),
                               torch.tensor(0))
            assert torch.equal(cnt[rows], want), f'hop {b_}: counts'
        assert torch.allclose(inv, 1.0 / cnt.clamp(min=1).float())
        slots, hop, dst, srow, sg = _decode(plan, b)
        # filled slots are exactly the first cnt of every row
        filled = torch.zeros(plan.S, dtype=torch.bool)
        filled[slots] = True
        for r in torch.randint(0, plan.R_dst, (200, )).tolist():
            want = torch.zeros(fan[[i for i in range(plan.L) if r >= plan.bases[i]][-1]],
                               dtype=torch.bool)
            want[:int(cnt[r])] = True
            got = filled[int(begin[r]):int(begin[r]) + want.numel()]
            assert torch.equal(got, want)
        # every sampled edge exists, no neighbour twice per destination
        vd = node_g[dst]
        assert bool((vd >= 0).all())
        for i in torch.randint(0, slots.numel(), (min(400, slots.numel()), )).tolist():
            d = int(vd[i])
            if d not in nbrs:
                nbrs[d] = set(row[colptr[d]:colptr[d + 1]].tolist())
            assert int(sg[i]) in nbrs[d]
        key = dst * n + sg
        assert key.unique().numel() == key.numel(), 'a neighbour was drawn twice'
        # the source row holds the sampled node and is its EARLIEST occurrence
        assert torch.equal(node_g[srow], sg)
        assert bool((srow <= plan.B + slots).all())
        # transposed CSRs: the same (source row, destination row) pairs, grouped by source
        for c in range(plan.n_csr):
            sel = slots < plan.t_slots[c]
            ptr, col = b.t_ptr[c].cpu().long(), b.t_col[c].cpu().long()
            assert int(ptr[-1]) == int(sel.sum()) and ptr.numel() == plan.t_rows[c] + 1
            rows_t = torch.repeat_interleave(torch.arange(plan.t_rows[c]), ptr[1:] - ptr[:-1])
            got = torch.sort(rows_t * plan.R + col[:int(ptr[-1])]).values
            want = torch.sort(srow[sel] * plan.R + dst[sel]).values
            assert torch.equal(got, want), f'transposed CSR {c}'
        if it == 0:
            first = (seeds, src_g.clone())
    # reproducible: another sampler, same seed / epoch / seeds -> the same draws; a different epoch
    # draws differently
    smp2 = SlotSampler(csc.ptr, csc.idx, n, plan, seed=5)
    e1 = torch.ones(1, dtype=torch.int64, device=dev)
    again = smp2.sample(first[0].to(dev), e1).src_g.cpu()
    assert torch.equal(again, first[1])
    other = smp2.sample(first[0].to(dev), e1 + 6).src_g.cpu()
    assert not torch.equal(other, first[1])


def _reference_stack(plan, b, x_rows, params, aggr):
    """The sampled computation graph evaluated with plain PyTorch ops (float64 when the inputs
    are): layer l aggregates the valid slots of hops 0 .. L-l-1 into the rows of blocks 0 .. L-l-1."""
    slots, hop, dst, srow, _ = _decode(plan, b)
    h = x_rows
    L = plan.L
    for l in range(L):
        W_l, bias, W_r = params[l]
        m = plan.bases[L - l]
        sel = hop <= L - l - 1
        agg = torch.zeros(m, h.size(1), dtype=h.dtype).index_add_(0, dst[sel], h[srow[sel]])
        if aggr == 'mean':
            cnt = torch.zeros(m, dtype=h.dtype).index_add_(0, dst[sel],
                                                           torch.ones(int(sel.sum()), dtype=h.dtype))
            agg = agg / cnt.clamp(min=1).view(-1, 1)
        h = agg @ W_l.t() + h[:m] @ W_r.t() + bias
        if l < L - 1:
            h = h.relu()
    return h


@pytest.mark.parametrize('direct', [False, True])
@pytest.mark.parametrize('aggr', ['mean', 'sum'])
@pytest.mark.parametrize('dims,fan,B', [((16, 32, 32, 8), [4, 3, 2], 50),
                                        ((128, 256, 256, 172), [5, 4, 3], 33),
                                        ((12, 20, 6), [6, 2], 70)])
def test_slot_stack_matches_plain_pytorch(dev, aggr, dims, fan, B, direct):
    """``direct``: only the destination rows of the batch are copied; layer 0 gathers its
    neighbours from the graph's feature matrix by graph node id (SlotSampler.gather(direct=True))."""
    import pytorch_geometric_amd as pga
    from pytorch_geometric_amd.nn import GraphSAGE
    from pytorch_geometric_amd.slots import SlotPlan, SlotSampler, run_slot_stack
    n = 3000
    g = gen(sum(dims) + B)
    ei = _graph(n, 40_000, seed=B)
    csc = pga.EdgeIndex(ei.to(dev), (n, n)).by_dst()
    x = torch.randn(n, dims[0], generator=g)
    plan = SlotPlan(B, fan, dev)
    smp = SlotSampler(csc.ptr, csc.idx, n, plan, seed=3)
    seeds = torch.randperm(n, generator=g)[:B]
    epoch = torch.ones(1, dtype=torch.int64, device=dev)
    b = smp.sample(seeds.to(dev), epoch)
    x_dev = x.to(dev)
    b.x = smp.gather(x_dev, b, direct=direct)
    node_g = b.node_g.cpu()
    x_rows = torch.where((node_g >= 0).view(-1, 1), x[node_g.clamp(min=0)], torch.zeros(1))
    if direct:
        assert b.x.size(0) == plan.R_dst < plan.R and b.x_global is x_dev
    assert torch.equal(b.x[:, dims[0]:].cpu(), x_rows[:b.x.size(0)])      # holes: zero rows
    torch.manual_seed(4)
    model = GraphSAGE(dims[0], dims[1], num_layers=len(fan), out_channels=dims[-1], aggr=aggr)
    if len(dims) == 4:
        assert [c.lin_l.weight.size(0) for c in model.convs] == list(dims[1:])
    st = model.state_dict()
    L = len(fan)

    def make(dtype):
        return [tuple(st[f'convs.{i}.{k}'].detach().clone().to(dtype).requires_grad_(True)
                      for k in ('lin_l.weight', 'lin_l.bias', 'lin_r.weight')) for i in range(L)]

    go = torch.randn(B, dims[-1], generator=g)
    p32, p64 = make(torch.float32), make(torch.float64)
    ref = _reference_stack(plan, b, x_rows, p32, aggr)
    ref.backward(go)
    ex = _reference_stack(plan, b, x_rows.double(), p64, aggr)
    ex.backward(go.double())
    model = model.to(dev)
    out = run_slot_stack(model, b)
    out.backward(go.to(dev))
    assert_close(out, ref.detach(), rtol=1e-5, atol=2e-5, what='slot stack output')
    got = [t for conv in model.convs for t in (conv.lin_l.weight.grad, conv.lin_l.bias.grad,
                                               conv.lin_r.weight.grad)]
    for gt, w32, w64 in zip(got, (t for layer in p32 for t in layer),
                            (t for layer in p64 for t in layer)):
        # against fp64, with the plain fp32 evaluation's own distance as the yardstick
        scale = max(float(w64.grad.abs().max()), 1.0)
        e_gpu = float((gt.cpu().double() - w64.grad).abs().max()) / scale
        e_ref = float((w32.grad.double() - w64.grad).abs().max()) / scale
        assert e_gpu <= max(2 * e_ref, 2e-5), (e_gpu, e_ref)
    assert_close_scaled(got[0], p32[0][0].grad, tol=1e-4, what='layer 0 lin_l grad')


def test_slot_step_is_capturable_and_draws_fresh_batches(dev):
    """sampling + gather + forward + backward of a slot batch as ONE hipGraph; bumping the device
    epoch between replays draws a new batch; the replayed results equal the eager ones."""
    import pytorch_geometric_amd as pga
    from pytorch_geometric_amd.hipgraph import CapturedStep
    from pytorch_geometric_amd.loader import NeighborLoader
    from pytorch_geometric_amd.nn import GraphSAGE
    from pytorch_geometric_amd.slots import run_slot_stack
    n, B, fan = 5000, 64, [5, 4, 3]
    g = gen(8)
    ei = _graph(n, 80_000, seed=9).to(dev)
    x = torch.randn(n, 32, generator=g).to(dev)
    y = torch.randint(0, 7, (n, ), generator=g).to(dev)
    loader = NeighborLoader(x, ei, fan, batch_size=B, y=y, seed=1)
    torch.manual_seed(1)
    model = GraphSAGE(32, 48, num_layers=3, out_channels=7).to(dev)
    seeds = torch.randperm(n, generator=g)[:B].to(dev)
    epoch = torch.zeros(1, dtype=torch.int64, device=dev)
    loss_buf = torch.zeros((), device=dev)
    grads = [torch.zeros_like(p) for p in model.parameters()]

    def step():
        epoch.add_(1)
        b = loader.collate_slots(seeds, epoch)
        for p in model.parameters():
            p.grad = None
        out = run_slot_stack(model, b)
        loss = torch.nn.functional.cross_entropy(out, b.y)
        loss.backward()
        loss_buf.copy_(loss.detach())
        for gb, p in zip(grads, model.parameters()):
            gb.copy_(p.grad)

    cap = CapturedStep(step, warmup=2)       # (the capture itself records, it does not run)
    seen = []
    for _ in range(3):
        cap()
        torch.cuda.synchronize()
        seen.append((float(loss_buf), [gb.clone() for gb in grads]))
    assert len({round(s[0], 6) for s in seen}) == 3, 'replays did not draw new batches'
    # eager twin of the last replay: the same epoch again (claims are epoch-stamped, the sampler is
    # a pure function of (seeds, epoch): nothing to reset)
    last_epoch = int(epoch.item())
    epoch.fill_(last_epoch - 1)
    step()
    torch.cuda.synchronize()
    assert abs(float(loss_buf) - seen[-1][0]) <= 1e-5 * max(1.0, abs(seen[-1][0]))
    for a, c in zip(grads, seen[-1][1]):
        assert_close_scaled(a, c, tol=2e-5, what='replay vs eager gradient')


def test_slot_stack_refuses_unsupported_models_and_stale_batches(dev):
    """`run_slot_stack` computes relu(lin_l(aggr x_j) + lin_r(x_i)) and nothing else — a
    model with dropout / normalize / project / root_weight=False / mixed aggregations raises
    instead of being silently mis-evaluated; and a batch whose sampler has sampled again (its
    index buffers are the sampler's own) is refused by forward and backward."""
    import pytorch_geometric_amd as pga
    from pytorch_geometric_amd.nn import GraphSAGE
    from pytorch_geometric_amd.slots import SlotPlan, SlotSampler, run_slot_stack
    n, B, fan = 2000, 40, [4, 3]
    g = gen(3)
    ei = _graph(n, 30_000, seed=2)
    csc = pga.EdgeIndex(ei.to(dev), (n, n)).by_dst()
    x = torch.randn(n, 16, generator=g).to(dev)
    plan = SlotPlan(B, fan, dev)
    smp = SlotSampler(csc.ptr, csc.idx, n, plan, seed=1)
    epoch = torch.ones(1, dtype=torch.int64, device=dev)
    seeds = torch.randperm(n, generator=g)[:B].to(dev)

    def batch():
        epoch.add_(1)
        b = smp.sample(seeds, epoch)
        b.x = smp.gather(x, b)
        return b

    def model(**kw):
        torch.manual_seed(0)
        return GraphSAGE(16, 24, num_layers=2, out_channels=5, **kw).to(dev)

    b = batch()
    for kw, msg in ((dict(dropout=0.5), 'dropout'), (dict(normalize=True), 'normalize'),
                    (dict(project=True), 'project'), (dict(root_weight=False), 'root_weight'),
                    (dict(aggr='max'), 'aggr'), (dict(act='elu'), 'ReLU')):
        with pytest.raises(ValueError, match=msg):
            run_slot_stack(model(**kw), b)
    m = model(dropout=0.5).eval()          # dropout is the identity in eval mode: fine
    run_slot_stack(m, b)
    m = model()
    m.convs[1].aggr = 'sum'
    with pytest.raises(ValueError, match='one aggregation'):
        run_slot_stack(m, b)
    with pytest.raises(ValueError, match='3 layers on a batch of 2 hops'):
        torch.manual_seed(0)
        run_slot_stack(GraphSAGE(16, 24, num_layers=3, out_channels=5).to(dev), b)
    # stale batches
    m = model()
    out = run_slot_stack(m, b)
    nxt = batch()                          # the sampler's buffers now hold the NEXT batch
    with pytest.raises(RuntimeError, match='sampled again'):
        out.sum().backward()
    with pytest.raises(RuntimeError, match='sampled again'):
        run_slot_stack(m, b)
    out = run_slot_stack(m, nxt)           # the current batch is fine, forward and backward
    out.sum().backward()
    # a second sampler is how the next batch is prefetched before backward()
    smp2 = SlotSampler(csc.ptr, csc.idx, n, plan, seed=1)
    cur = batch()
    out = run_slot_stack(m, cur)
    epoch.add_(1)
    ahead = smp2.sample(seeds, epoch)
    ahead.x = smp2.gather(x, ahead)
    out.sum().backward()
    run_slot_stack(m, ahead).sum().backward()
