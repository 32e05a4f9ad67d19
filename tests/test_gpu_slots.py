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


# (fan-outs 16 / 17 and 32 / 33: the boundaries between 16, 32 and 64 lanes per frontier position)
@pytest.mark.parametrize('fan,B', [([4, 3, 2], 64), ([15, 10, 5], 32), ([3], 100), ([64, 2], 7),
                                   ([16, 17], 20), ([32, 33], 5), ([24, 1], 33)])
def test_slot_sampler_contract(dev, fan, B):
    import pytorch_geometric_amd as pga
    from pytorch_geometric_amd.slots import SlotPlan, SlotSampler
    n = 4000
    # (degrees above the fan-out somewhere, or nothing is ever DRAWN: 15 on average, 75 for the wide ones)
    ei = _graph(n, 60_000 if max(fan) <= 16 or 64 in fan else 300_000, seed=B + len(fan))
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


def _trainer_setup(dev, n=5000, B=64, fan=(5, 4, 3), feat=32, hidden=48, classes=7, bias=True,
                   index_dtype=torch.int64):
    from pytorch_geometric_amd.loader import NeighborLoader
    from pytorch_geometric_amd.nn import GraphSAGE
    g = gen(8)
    ei = _graph(n, 80_000, seed=9).to(dev).to(index_dtype)
    x = torch.randn(n, feat, generator=g).to(dev)
    y = torch.randint(0, classes, (n, ), generator=g).to(dev)

    def make():
        loader = NeighborLoader(x, ei, list(fan), batch_size=B, y=y, seed=1)
        torch.manual_seed(1)
        model = GraphSAGE(feat, hidden, num_layers=len(fan), out_channels=classes,
                          bias=bias).to(dev)
        return loader, model

    seed_sets = [torch.randperm(n, generator=g)[:B].to(dev).to(index_dtype) for _ in range(5)]
    return make, seed_sets


@pytest.mark.parametrize('capture,pipeline,bias', [(False, False, True), (True, False, True),
                                                   (True, False, False), (False, True, True),
                                                   (True, True, True)])
def test_slot_trainer_equals_autograd_cross_entropy_and_torch_adam(dev, capture, pipeline, bias):
    """slots.SlotTrainer (flat parameters in the kernels' layout, one-launch loss, one-launch Adam
    that also refreshes the transposed weights; pipelined: the next batch drawn beside the current
    one's training) against the step it replaces — run_slot_stack through autograd,
    F.cross_entropy, torch.optim.Adam (examples/multi_gpu/distributed_sampling.py:104-117) — on the
    same batches: losses, gradients and parameters after every step; captured, the recording's
    warm-up leaves no trace."""
    from pytorch_geometric_amd.slots import SlotTrainer, run_slot_stack
    make, seed_sets = _trainer_setup(dev, bias=bias)
    loader_r, model_r = make()
    loader_t, model_t = make()
    opt = torch.optim.Adam(model_r.parameters(), lr=1e-2)
    trainer = SlotTrainer(model_t, loader_t, lr=1e-2, capture=capture, pipeline=pipeline)
    # the model's parameters are views of the flat buffer now, values unchanged
    for (k, a), (_, b) in zip(model_t.named_parameters(), model_r.named_parameters()):
        assert torch.equal(a, b), k
        assert trainer.flat.data_ptr() <= a.data_ptr() < trainer.flat.data_ptr() + 4 * trainer.n
    epoch = torch.zeros(1, dtype=torch.int64, device=dev)

    def reference_step(i, seeds, ep, loss_t):
        epoch.fill_(ep)
        b = loader_r.collate_slots(seeds, epoch)
        opt.zero_grad()
        loss_r = torch.nn.functional.cross_entropy(run_slot_stack(model_r, b), b.y)
        loss_r.backward()
        lr_ = float(loss_r.detach())
        assert abs(float(loss_t) - lr_) <= 1e-5 * max(1.0, abs(lr_)), (i, float(loss_t), lr_)
        for (k, a), (_, c) in zip(model_t.named_parameters(), model_r.named_parameters()):
            assert_close_scaled(a.grad, c.grad, tol=2e-5, what=f'step {i}: grad {k}')
        opt.step()
        for (k, a), (_, c) in zip(model_t.named_parameters(), model_r.named_parameters()):
            # Adam divides by sqrt(v): a gradient element near 0 amplifies a 1e-7 difference of
            # the gradients into a visible one of the update (|update| <= lr): compare at lr scale
            assert float((a - c).abs().max()) <= 2e-2 * 1e-2, (i, k, float((a - c).abs().max()))
            assert float((a - c).abs().mean()) <= 1e-3 * 1e-2, (i, k)
            with torch.no_grad():   # every step is compared from the same parameters (the two
                c.copy_(a)          # trajectories would drift apart through that amplification)

    drawn = []   # (seeds, epoch of the draw) in the order the trainer drew them
    for i, seeds in enumerate(seed_sets):
        loss_t = trainer.step(seeds)
        drawn.append((seeds, int(trainer.epoch.item())))
        if not pipeline:
            reference_step(i, *drawn[i], loss_t)
        elif i > 0:               # this call trained on the batch of the previous call's seeds
            reference_step(i - 1, *drawn[i - 1], loss_t)
    if pipeline:
        loss_t = trainer.finish()
        reference_step(len(seed_sets) - 1, *drawn[-1], loss_t)
    assert int(trainer.opt_step.item()) == len(seed_sets)
    # the transposed copies the input-gradient GEMMs read are current
    for l in range(1, trainer.L):
        assert torch.equal(trainer.wt[l], trainer.wmat[l].t())
    trainer.check_labels()
    # the re-bound model still evaluates (views) and checkpoints
    sd = model_t.state_dict()
    assert all(torch.isfinite(v).all() for v in sd.values())


def test_cross_entropy_step_kernel_vs_torch(dev):
    """pygamd_cross_entropy_step against F.cross_entropy + autograd: labels through an index,
    ragged class counts, an out-of-range label (flag, zero row)."""
    import ctypes

    from pytorch_geometric_amd import _lib, _native
    from pytorch_geometric_amd._lib import check
    lib = _lib.load()
    g = gen(5)
    for B, C, ld in ((1, 1, 1), (7, 3, 8), (1024, 172, 172), (333, 700, 704)):
        logits = (torch.randn(B, ld, generator=g) * 3).to(dev)
        view = logits[:, :C]
        y_all = torch.randint(0, C, (5000, ), generator=g).to(dev)
        idx = torch.randint(0, 5000, (B, ), generator=g).to(dev)
        grad = torch.full((B, ld), 7.0, device=dev)
        loss = torch.zeros((), device=dev)
        err = torch.zeros(1, dtype=torch.int32, device=dev)
        nb = ctypes.c_size_t(0)
        check(lib.pygamd_cross_entropy_step_workspace_bytes(B, ctypes.byref(nb)))
        ws = torch.zeros(nb.value, dtype=torch.uint8, device=dev)
        counter = torch.full((1, ), 40, dtype=torch.int64, device=dev)
        for rep in range(2):   # the ticket re-arms itself
            check(lib.pygamd_cross_entropy_step(
                _native._p(view), ld, None, B, B, C, _native._p(y_all), _native._p(idx),
                _native._p(grad),
                ld, _native._p(loss), _native._p(ws), nb.value, _native._p(err),
                _native._p(counter), _native._stream(logits)))
        ref_in = view.detach().clone().requires_grad_(True)
        ref = torch.nn.functional.cross_entropy(ref_in, y_all[idx])
        ref.backward()
        assert abs(float(loss) - float(ref)) <= 1e-6 * max(1.0, abs(float(ref))), (B, C)
        assert_close(grad[:, :C], ref_in.grad, rtol=1e-5, atol=1e-7, what=f'CE grad {B}x{C}')
        assert torch.all(grad[:, C:] == 7.0) and int(err) == 0 and int(counter) == 42
    # an out-of-range label: flagged, its row contributes nothing, the mean still divides by B
    y_bad = y_all.clone()
    y_bad[idx[0]] = 700
    check(lib.pygamd_cross_entropy_step(
        _native._p(view), ld, None, B, B, C, _native._p(y_bad), _native._p(idx), _native._p(grad),
        ld,
        _native._p(loss), _native._p(ws), nb.value, _native._p(err), None,
        _native._stream(logits)))
    good = (y_bad[idx] < C)
    rows = torch.nn.functional.cross_entropy(view, y_bad[idx].clamp(max=C - 1), reduction='none')
    want = float((rows * good).sum() / B)
    assert int(err) == 1 and abs(float(loss) - want) <= 1e-6 * max(1.0, abs(want))
    assert torch.all(grad[~good, :C] == 0)


def test_adam_step_kernel_vs_torch(dev):
    """pygamd_adam_step against torch.optim.Adam over several steps (with and without weight
    decay, a gradient scale), and its transposed copies."""
    import ctypes

    from pytorch_geometric_amd import _lib, _native
    from pytorch_geometric_amd._lib import check
    lib = _lib.load()
    g = gen(6)
    n = 3 * 40 + 8 + 5 * 12 + 4
    for wd, scale in ((0.0, 1.0), (0.01, 0.5)):
        p0 = torch.randn(n, generator=g).to(dev)
        flat, m, v = p0.clone(), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
        ref = p0.clone().requires_grad_(True)
        opt = torch.optim.Adam([ref], lr=3e-3, betas=(0.8, 0.95), eps=1e-6, weight_decay=wd)
        step = torch.full((1, ), 10, dtype=torch.int64, device=dev)
        wt = torch.zeros(3 * 40 + 5 * 12, device=dev)
        segs = ((0, 3, 40, 0), (128, 5, 12, 120))
        arr = lambda ty, k: (ty * 2)(*[s[k] for s in segs])   # noqa: E731
        for it in range(6):
            grad = torch.randn(n, generator=g).to(dev)
            step.add_(1)
            check(lib.pygamd_adam_step(
                _native._p(flat), _native._p(grad), _native._p(m), _native._p(v), n,
                _native._p(step), 10, 3e-3, 0.8, 0.95, 1e-6, wd, scale, _native._p(wt), 2,
                arr(ctypes.c_int64, 0), arr(ctypes.c_int32, 1), arr(ctypes.c_int32, 2),
                arr(ctypes.c_int64, 3), _native._stream(flat)))
            ref.grad = grad * scale
            opt.step()
            assert_close(flat, ref.detach(), rtol=1e-5, atol=1e-6, what=f'Adam step {it}')
        assert torch.equal(wt[:120].view(40, 3), flat[:120].view(3, 40).t())
        assert torch.equal(wt[120:].view(12, 5), flat[128:188].view(5, 12).t())


def test_slot_trainer_on_an_int32_graph(dev):
    """int32 `edge_index` (the reference accepts it, test_message_passing.py:686-703): the sampler
    runs on int32 pointers, the seeds are int32, the labels are still read through them."""
    from pytorch_geometric_amd.slots import SlotTrainer, run_slot_stack
    make, seed_sets = _trainer_setup(dev, fan=(4, 3), index_dtype=torch.int32)
    loader_r, model_r = make()
    loader_t, model_t = make()
    trainer = SlotTrainer(model_t, loader_t, lr=1e-2, capture=True)
    assert trainer.seeds.dtype == torch.int32
    epoch = torch.zeros(1, dtype=torch.int64, device=dev)
    losses = [float(trainer.step(seeds)) for seeds in seed_sets[:3]]   # captured: replays run
    assert all(v == v and v > 0 for v in losses) and len(set(losses)) == 3
    # one clean comparison: a fresh pair, one step
    loader_r, model_r = make()
    loader_t, model_t = make()
    trainer = SlotTrainer(model_t, loader_t, lr=1e-2, capture=False)
    loss_t = float(trainer.step(seed_sets[0]))
    epoch.fill_(int(trainer.epoch.item()))
    b = loader_r.collate_slots(seed_sets[0], epoch)
    loss_r = torch.nn.functional.cross_entropy(run_slot_stack(model_r, b), b.y)
    loss_r.backward()
    assert abs(loss_t - float(loss_r.detach())) <= 1e-5 * max(1.0, abs(float(loss_r.detach())))
    for (k, a), (_, c) in zip(model_t.named_parameters(), model_r.named_parameters()):
        assert_close_scaled(a.grad, c.grad, tol=2e-5, what=f'int32 graph: grad {k}')
    trainer.check_labels()
