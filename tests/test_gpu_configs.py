"""BASELINE.json configs 1, 3 and 5 at their full (synthetic-shape) sizes: HIP path vs the CPU
oracle on identical seeded inputs, forward values and input/parameter gradients.  (Config 2 at full
size is covered by tests/test_gpu_scale.py through properties and by bench.py; config 4 needs the
sampler)"""
import pytest
import torch

from oracle import pyg_oracle as O
from tests._util import (assert_close, assert_close_outliers, assert_close_rows,
                         assert_close_scaled, gen)

pytestmark = pytest.mark.gpu


def test_config1_gcn_cora_shape(dev):
    """GCNConv 2-layer on a Cora-shaped graph (N=2,708, E=10,556 mirrored pairs, F=1,433, 7
    classes), examples/gcn.py plumbing: GCN(1433, 16, 2 layers, out 7), cached normalisation."""
    from pytorch_geometric_amd.nn import GCN
    g = gen(0)
    n, pairs = 2708, 5278
    u = torch.randint(0, n, (pairs, ), generator=g)
    v = torch.randint(0, n, (pairs, ), generator=g)
    ei = torch.stack([torch.cat([u, v]), torch.cat([v, u])])
    x = torch.rand(n, 1433, generator=g)
    x = x / x.sum(1, keepdim=True)  # NormalizeFeatures
    go = torch.randn(n, 7, generator=g)
    torch.manual_seed(0)
    model = GCN(1433, 16, num_layers=2, out_channels=7, cached=True)
    st = {k: v.clone() for k, v in model.state_dict().items()}
    params = [(st[f'convs.{i}.lin.weight'].requires_grad_(True),
               st[f'convs.{i}.bias'].requires_grad_(True)) for i in range(2)]
    ref = O.gcn(x, ei, params)
    ref.backward(go)
    model = model.to(dev).eval()
    out = model(x.to(dev), ei.to(dev))
    out.backward(go.to(dev))
    out2 = model(x.to(dev), ei.to(dev))  # second call hits the cached normalisation + handle
    assert_close(out, ref.detach(), what='gcn out')
    assert_close(out2, ref.detach(), what='gcn out (cached)')
    for i in range(2):
        assert_close(model.convs[i].lin.weight.grad, params[i][0].grad, atol=2e-5,
                     what=f'lin{i} grad')
        assert_close(model.convs[i].bias.grad, params[i][1].grad, atol=2e-5, what=f'b{i} grad')


def test_config3_gat_arxiv_shape(dev):
    """GATConv 3-layer heads=8 on an ogbn-arxiv-shaped graph (N=169,343, E=1,166,243): edge
    softmax + multi-head weighted aggregation, widths 8x32 / 8x32 / 8x40 (mean over heads)."""
    from pytorch_geometric_amd.nn import GAT
    g = gen(2)
    n, e = 169_343, 1_166_243
    ei = torch.randint(0, n, (2, e), generator=g)
    x = torch.randn(n, 128, generator=g)
    torch.manual_seed(2)
    model = GAT(128, 256, num_layers=3, out_channels=40, heads=8)
    st = {k: v.clone() for k, v in model.state_dict().items()}
    params = [(st[f'convs.{i}.lin.weight'], st[f'convs.{i}.att_src'], st[f'convs.{i}.att_dst'],
               st[f'convs.{i}.bias']) for i in range(3)]
    go = torch.randn(n, 40, generator=g)
    xr = x.clone().requires_grad_(True)
    ref = O.gat(xr, ei, params, heads=8)
    ref.backward(go)
    model = model.to(dev).eval()
    xg = x.to(dev).requires_grad_(True)
    out = model(xg, ei.to(dev))
    out.backward(go.to(dev))
    assert_close(out, ref.detach(), atol=2e-5, what='gat out')
    assert_close_outliers(xg.grad, xr.grad, what='gat grad_x')  # ReLU-boundary flips, see _util
    # one layer (no ReLU in between): every element tight, values and all gradients
    from pytorch_geometric_amd.nn import GATConv
    torch.manual_seed(3)
    conv = GATConv(128, 32, heads=8)
    ps = [p.detach().clone().requires_grad_(True) for p in
          (conv.lin.weight, conv.att_src, conv.att_dst, conv.bias)]
    xr = x.clone().requires_grad_(True)
    go = torch.randn(n, 256, generator=g)
    ref = O.gat_conv(xr, ei, ps[0], ps[1], ps[2], ps[3], 8, 32)
    ref.backward(go)
    conv = conv.to(dev)
    xg = x.to(dev).requires_grad_(True)
    out = conv(xg, ei.to(dev))
    out.backward(go.to(dev))
    assert_close(out, ref.detach(), atol=2e-5, what='gat layer out')
    # (leaky_relu runs once per edge and head, 9.3 M evaluations: a pre-activation within fp32
    # rounding of 0 is expected about once — see assert_close_rows)
    assert_close_rows(xg.grad, xr.grad, max_bad_rows=32, what='gat layer grad_x')
    # parameter gradients: sums over 169 k nodes / 9.3 M (edge, head) terms with heavy cancellation
    # (|sum| ~ sqrt(N) x |term|): two correct fp32 evaluations differ by their summation order, and
    # a leaky_relu pre-activation within rounding of 0 takes the other slope.  Judged against an
    # fp64 evaluation, with the CPU fp32 reference's own distance to it as the yardstick.
    p64 = [p.detach().double().requires_grad_(True) for p in ps]
    x64 = x.double().requires_grad_(True)
    O.gat_conv(x64, ei, p64[0], p64[1], p64[2], p64[3], 8, 32).backward(go.double())
    for got, r32, r64, what in ((conv.att_src.grad, ps[1].grad, p64[1].grad, 'att_src'),
                                (conv.att_dst.grad, ps[2].grad, p64[2].grad, 'att_dst'),
                                (conv.lin.weight.grad, ps[0].grad, p64[0].grad, 'W')):
        scale = max(float(r64.abs().max()), 1.0)
        e_gpu = float((got.cpu().double() - r64).abs().max()) / scale
        e_ref = float((r32.double() - r64).abs().max()) / scale
        assert e_gpu <= max(2 * e_ref, 2e-5), f'gat layer grad {what}: {e_gpu:.2e} vs ref {e_ref:.2e}'


def test_config5_rgcn_fb15k237_shape(dev):
    """RGCNConv 2-layer on an FB15k-237-shaped graph (N=14,541, E=544,230, 474 relations with a
    Zipf histogram): dense-weight variant RGCNConv(64, 64, 474) and the block-diagonal variant of
    examples/rgcn_link_pred.py (num_blocks=5; width 100 here to keep the CPU oracle in seconds)."""
    from pytorch_geometric_amd.nn import RGCNConv
    g = gen(4)
    n, e, R = 14_541, 544_230, 474
    ei = torch.randint(0, n, (2, e), generator=g)
    et = (torch.rand(e, generator=g).pow(4) * R).long().clamp(max=R - 1)
    go = torch.randn(n, 64, generator=g)
    x = torch.randn(n, 64, generator=g)
    torch.manual_seed(4)
    convs = [RGCNConv(64, 64, R), RGCNConv(64, 64, R)]
    xr = x.clone().requires_grad_(True)
    h = xr
    ws = []
    for i, c in enumerate(convs):
        w = [c.weight.detach().clone().requires_grad_(True),
             c.root.detach().clone().requires_grad_(True),
             c.bias.detach().clone().requires_grad_(True)]
        ws.append(w)
        h = O.rgcn_conv(h, ei, et, *w)
        if i == 0:
            h = h.relu()
    h.backward(go)
    xg = x.to(dev).requires_grad_(True)
    hg = xg
    eid, etd = ei.to(dev), et.to(dev)
    for i, c in enumerate(convs):
        c.to(dev)
        hg = c(hg, eid, etd)
        if i == 0:
            hg = hg.relu()
    hg.backward(go.to(dev))
    assert_close(hg, h.detach(), atol=5e-5, what='rgcn out')
    assert_close_outliers(xg.grad, xr.grad, what='rgcn grad_x')
    assert_close_outliers(convs[0].weight.grad, ws[0][0].grad, what='rgcn grad W')
    assert_close_scaled(convs[1].weight.grad, ws[1][0].grad, what='rgcn grad W (last layer)')
    # block-diagonal decomposition
    torch.manual_seed(5)
    conv = RGCNConv(100, 100, R, num_blocks=5)
    x2 = torch.randn(n, 100, generator=g)
    ref = O.rgcn_conv_blocks(x2, ei, et, conv.weight.detach(), conv.root.detach(),
                             conv.bias.detach())
    out = conv.to(dev)(x2.to(dev), eid, etd)
    assert_close(out, ref, atol=5e-5, what='rgcn blocks out')


def test_config5_at_the_timed_shape(dev):
    """BASELINE config 5 exactly as `bench_configs.config5` / scripts/time_rgcn.py time it
    (VERDICT r5 #3): an embedding Parameter [14,541, 500] -> RGCNConv(500, 500, 474, num_blocks=5)
    -> ReLU -> RGCNConv(500, 500, 474, num_blocks=5) on the FB15k-237 shape — output, the gradient
    of the embedding and of every parameter of both layers (block weights [474, 5, 100, 100], root,
    bias), i.e. the 100 x 100-block grouped GEMM forward, its input gradient and its weight
    gradient, against the oracle evaluated in float64 (`rgcn_conv_blocks_pairs`: the reference's
    per-relation loop restricted to the rows that can be non-zero, pinned on the reference's own
    vectors in tests/test_oracle_golden.py).  Tolerance: 2e-5 of each tensor's largest magnitude
    (north_star's 1e-5 on the aggregation, doubled for two stacked layers), no outlier allowance."""
    from pytorch_geometric_amd.nn import RGCNConv
    g = gen(4)
    n, e, R = 14_541, 544_230, 474
    ei = torch.randint(0, n, (2, e), generator=g)
    et = (torch.rand(e, generator=g).pow(4) * R).long().clamp(max=R - 1)
    emb = torch.randn(n, 500, generator=g)
    go = torch.randn(n, 500, generator=g)
    torch.manual_seed(6)
    convs = [RGCNConv(500, 500, R, num_blocks=5), RGCNConv(500, 500, R, num_blocks=5)]
    for c in convs:
        with torch.no_grad():
            c.bias.uniform_(-0.1, 0.1)   # (zeros at init: give the bias path something to do)
    # device
    emb_d = torch.nn.Parameter(emb.to(dev))
    eid, etd = ei.to(dev), et.to(dev)
    for c in convs:
        c.to(dev)
    hid = convs[0](emb_d, eid, etd).relu()
    out = convs[1](hid, eid, etd)
    out.backward(go.to(dev))
    # float64 oracle ON THE DEVICE'S ACTIVATION PATTERN: of the 7.3 M hidden units a handful sit
    # within float32 rounding of 0 and would take the other ReLU branch in float64 — one such unit
    # moves ~4 k elements of the embedding gradient by 1e-3 (seen: 4.7e-4 of the elements), which
    # says nothing about a kernel.  The oracle therefore applies the mask the device produced, and
    # the mask itself is checked: it may differ from the float64 one only where the pre-activation
    # is within 1e-5 of the tensor's scale of zero.
    e64 = emb.double().requires_grad_(True)
    p64 = [[p.detach().cpu().double().requires_grad_(True) for p in (c.weight, c.root, c.bias)]
           for c in convs]
    pre = O.rgcn_conv_blocks_pairs(e64, ei, et, *p64[0])
    mask = (hid.detach() > 0).cpu()
    flipped = mask != (pre.detach() > 0)
    assert int(flipped.sum()) <= 64, f'{int(flipped.sum())} hidden units on the other ReLU branch'
    if bool(flipped.any()):
        assert float(pre.detach()[flipped].abs().max()) <= 1e-5 * float(pre.detach().abs().max())
    h = O.rgcn_conv_blocks_pairs(pre * mask.double(), ei, et, *p64[1])
    h.backward(go.double())

    def check(got, ref, what, tol=2e-5):
        scale = max(float(ref.abs().max()), 1.0)
        err = float((got.detach().cpu().double() - ref).abs().max()) / scale
        assert bool(torch.isfinite(got).all()) and err <= tol, f'{what}: {err:.2e} of the scale'

    check(hid, (pre * mask.double()).detach(), 'hidden layer')
    check(out, h.detach(), 'out')
    check(emb_d.grad, e64.grad, 'grad embedding')
    for li, (c, ps) in enumerate(zip(convs, p64)):
        check(c.weight.grad, ps[0].grad, f'layer {li} grad block weights')
        check(c.root.grad, ps[1].grad, f'layer {li} grad root')
        check(c.bias.grad, ps[2].grad, f'layer {li} grad bias')
    # relations without edges keep an exactly-zero weight gradient
    empty = torch.bincount(et, minlength=R) == 0
    if bool(empty.any()):
        assert float(convs[0].weight.grad[empty.to(dev)].abs().max()) == 0.0
