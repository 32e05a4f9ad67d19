"""The OTHER BASELINE.json configs beside the headline, for the driver's record:
configs 1 (GCN, Cora shape), 3 (GAT, ogbn-arxiv shape), 5 (RGCN, FB15k-237 shape) — each with this
package's own classes AND with the REFERENCE's classes behind ``backend.install()`` — config 2 with
the reference's ``GraphSAGE`` + ``install()``, and config 4 (captured mini-batch step) on one GPU.

``bench.py`` calls :func:`run` AFTER its timed region, on rank 0 of a one-GPU run only; every value
in the returned dict is a scalar or a short string (the driver's record keeps scalars and the
first 120 characters of strings).  Nothing here is the headline.

Per config:
  ms_per_step / edges_per_s       own classes, fwd + bwd (``out.sum().backward()``), eager
  ref_ms_per_step / ref_vs_own    torch_geometric's classes + install(), same shapes
  parity_err / parity_ok          forward of the reference's classes + install() on the GPU against
                                  THE SAME module objects on the CPU (the reference's own path, where
                                  the backend steps aside), max |diff| / max(1, max |ref|), ok <= 2e-5
  alg_GB                          SURVEY.md §8(d) bytes of one step (formulas in the functions)
  frac_hbm                        alg_GB / ms_per_step / 8 TB/s   (own classes)
  dominant_kernel / dominant_ms   longest device symbol of the own-class step (torch.profiler)
The reference is the staged copy the cpu_baseline leg uses (``oracle/make_ref.py``): it supplies
the HOST-side classes; every device computation behind them is this package's HIP library.
"""
import copy
import time

import torch

HBM_PEAK = 8.0e12


def _timeit(fn, warm=3, steps=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def _dominant(fn, steps=3):
    """(kernel symbol, ms per step) of the device kernel with the largest summed time."""
    try:
        from torch.profiler import ProfilerActivity, profile
        fn()
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            for _ in range(steps):
                fn()
            torch.cuda.synchronize()
        best, best_t = None, 0.0
        for ev in prof.key_averages():
            t = float(getattr(ev, 'self_device_time_total', 0.0)
                      or getattr(ev, 'self_cuda_time_total', 0.0) or 0.0)
            if t > best_t and 'Memcpy' not in ev.key and 'Memset' not in ev.key:
                best, best_t = ev.key, t
        if best is None:
            return None, None
        name = best.split('(')[0].replace('void ', '').strip()
        return name[:100], round(best_t / steps / 1e3, 4)
    except Exception as exc:  # (a side figure: never at the price of the bench line)
        return f'profiler unavailable: {type(exc).__name__}'[:100], None


def _err(got, ref):
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    return float((got - ref).abs().max()) / max(1.0, float(ref.abs().max()))


def _entry(name, own_step, ref_step, parity, alg_bytes, edges, out):
    ms = _timeit(own_step)
    e = {'ms_per_step': round(ms, 4), 'edges_per_s': round(edges / (ms * 1e-3), 1),
         'alg_GB': round(alg_bytes / 1e9, 4),
         'frac_hbm': round(alg_bytes / (ms * 1e-3) / HBM_PEAK, 4)}
    e['dominant_kernel'], e['dominant_ms'] = _dominant(own_step)
    if ref_step is not None:
        rms = _timeit(ref_step)
        e['ref_ms_per_step'] = round(rms, 4)
        e['ref_vs_own'] = round(rms / ms, 3)
    if parity is not None:
        t0 = time.perf_counter()
        err = parity()
        e['parity_err'] = float(f'{err:.3e}')
        e['parity_ok'] = bool(err <= 2e-5)
        e['parity_s'] = round(time.perf_counter() - t0, 1)
    out[name] = e


def _spmm_bytes(E, n_dst, F, b=8, heads=0):
    """One fused CSR aggregation pass (§8(d)): every edge reads one source row + one index (+ its
    per-head weights), plus the row pointer and the output rows."""
    return E * (4 * F + b + 4 * heads) + (n_dst + 1) * b + n_dst * 4 * F


def config1(dev, pyg, out):
    """GCN 2-layer on the Cora shape (N = 2,708, E = 10,556 + 2,708 loops, 1,433 -> 16 -> 7)."""
    from pytorch_geometric_amd.nn import GCN
    g = torch.Generator().manual_seed(0)
    n, pairs = 2708, 5278
    u, v = torch.randint(0, n, (pairs, ), generator=g), torch.randint(0, n, (pairs, ), generator=g)
    ei_c = torch.stack([torch.cat([u, v]), torch.cat([v, u])])
    x_c = torch.rand(n, 1433, generator=g)
    x_c = x_c / x_c.sum(1, keepdim=True)
    x, ei = x_c.to(dev), ei_c.to(dev)
    torch.manual_seed(0)
    own = GCN(1433, 16, num_layers=2, out_channels=7, cached=True).to(dev)

    def step(model=own):
        model.zero_grad()
        model(x, ei).sum().backward()

    ref_step = parity = None
    if pyg is not None:
        from torch_geometric.nn import GCN as RefGCN
        torch.manual_seed(0)
        ref_cpu = RefGCN(1433, 16, num_layers=2, out_channels=7, cached=True)
        ref = copy.deepcopy(ref_cpu).to(dev)
        ref_step = lambda: step(ref)  # noqa: E731

        def parity():
            with torch.no_grad():
                return _err(copy.deepcopy(ref_cpu).to(dev)(x, ei), ref_cpu(x_c, ei_c))

    Ep = ei.size(1) + n
    # aggregation at widths 16 and 7 forward, 7 backward (x takes no gradient); the 2,708 x 1,433
    # feature matrix read by the first transform and by its weight gradient
    alg = (_spmm_bytes(Ep, n, 16, heads=1) + 2 * _spmm_bytes(Ep, n, 7, heads=1)
           + 2 * n * 1433 * 4)
    _entry('config1_gcn_cora', step, ref_step, parity, alg, 2 * Ep, out)
    # the same step as ONE hipGraph (launch-bound: ~40 small kernels)
    try:
        for p in own.parameters():
            p.grad = torch.zeros_like(p)

        def gstep():
            own.zero_grad(set_to_none=False)
            own(x, ei).sum().backward()

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                gstep()
        torch.cuda.current_stream().wait_stream(side)
        cg = torch.cuda.CUDAGraph()
        with torch.cuda.graph(cg):
            gstep()
        out['config1_gcn_cora']['ms_per_step_hipgraph'] = round(_timeit(cg.replay, 5, 50), 4)
    except Exception as exc:
        out['config1_gcn_cora']['ms_per_step_hipgraph'] = None
        out['config1_gcn_cora']['hipgraph_error'] = str(exc)[:100]


def config3(dev, pyg, out):
    """GAT 3-layer, heads = 8, on the ogbn-arxiv shape (N = 169,343, E = 1,166,243 + N loops)."""
    from pytorch_geometric_amd.nn import GAT
    g = torch.Generator().manual_seed(2)
    n, e = 169_343, 1_166_243
    ei_c = torch.randint(0, n, (2, e), generator=g)
    x_c = torch.randn(n, 128, generator=g)
    x, ei = x_c.to(dev), ei_c.to(dev)
    torch.manual_seed(0)
    own = GAT(128, 256, num_layers=3, out_channels=40, heads=8).to(dev)

    def step(model=own):
        model.zero_grad()
        model(x, ei).sum().backward()

    ref_step = parity = None
    if pyg is not None:
        from torch_geometric.nn import GAT as RefGAT
        torch.manual_seed(0)
        ref_cpu = RefGAT(128, 256, num_layers=3, out_channels=40, heads=8)
        ref = copy.deepcopy(ref_cpu).to(dev)
        ref_step = lambda: step(ref)  # noqa: E731

        def parity():
            with torch.no_grad():
                return _err(copy.deepcopy(ref_cpu).to(dev)(x, ei), ref_cpu(x_c, ei_c))

    Ep = e + n  # (self-loops re-added by the layer; the synthetic graph has ~7 of its own)
    H = 8
    alg = 0
    for F in (256, 256, 320):
        soft = Ep * (4 * H * 2 + 8) + n * 4 * H * 2          # edge softmax, one direction
        agg = _spmm_bytes(Ep, n, F, heads=H)
        alg += soft + agg                                      # forward
        alg += soft + agg                                      # backward (same order)
    _entry('config3_gat_arxiv', step, ref_step, parity, alg, 3 * Ep, out)


def config5(dev, pyg, out):
    """RGCNConv(500, 500, 474, num_blocks=5) x 2 on an embedding Parameter, FB15k-237 shape
    (N = 14,541, E = 544,230; examples/rgcn_link_pred.py:27-47)."""
    from pytorch_geometric_amd.nn import RGCNConv
    g = torch.Generator().manual_seed(4)
    n, e, R = 14_541, 544_230, 474
    ei_c = torch.randint(0, n, (2, e), generator=g)
    et_c = (torch.rand(e, generator=g).pow(4) * R).long().clamp(max=R - 1)
    ei, et = ei_c.to(dev), et_c.to(dev)
    torch.manual_seed(0)
    emb_c = torch.randn(n, 500)
    emb = torch.nn.Parameter(emb_c.to(dev))
    c1 = RGCNConv(500, 500, R, num_blocks=5).to(dev)
    c2 = RGCNConv(500, 500, R, num_blocks=5).to(dev)

    def step(a=c1, b=c2):
        a.zero_grad()
        b.zero_grad()
        emb.grad = None
        b(a(emb, ei, et).relu(), ei, et).sum().backward()

    ref_step = parity = None
    if pyg is not None:
        from torch_geometric.nn import RGCNConv as RefRGCN
        torch.manual_seed(0)
        r1c, r2c = RefRGCN(500, 500, R, num_blocks=5), RefRGCN(500, 500, R, num_blocks=5)
        r1, r2 = copy.deepcopy(r1c).to(dev), copy.deepcopy(r2c).to(dev)
        ref_step = lambda: step(r1, r2)  # noqa: E731

        def parity():  # ONE layer: the reference's CPU loop is 474 masked propagates per layer
            with torch.no_grad():
                return _err(r1(emb.detach(), ei, et), r1c(emb_c, ei_c, et_c))

    S = int(torch.unique(et_c * n + ei_c[1]).numel())   # (relation, destination) pairs
    W = R * 5 * 100 * 100 * 4
    layer = (_spmm_bytes(e, S, 500) + (2 * S * 500 * 4 + W) + _spmm_bytes(S, n, 500)
             + 2 * n * 500 * 4)                               # + root transform
    alg = 2 * layer + 2 * (layer + S * 500 * 4 + W)           # forward + backward (d in, d W)
    _entry('config5_rgcn_fb15k237', step, ref_step, parity, alg, 2 * e, out)
    out['config5_rgcn_fb15k237']['pair_segments'] = S


def config2_reference(dev, pyg, x, ei, train_idx, y_train, num_classes, own_ms, out):
    """The headline step written with the REFERENCE's ``GraphSAGE`` + install(): same graph
    tensors (the sorted handle is found again by identity), same loss, Adam included."""
    import torch.nn.functional as F
    from torch_geometric.nn import GraphSAGE as RefSAGE
    torch.manual_seed(0)
    model = RefSAGE(100, 256, num_layers=3, out_channels=num_classes).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)

    def step():
        opt.zero_grad(set_to_none=False)
        loss = F.cross_entropy(model(x, ei)[train_idx], y_train)
        loss.backward()
        opt.step()

    ms = _timeit(step, warm=2, steps=5)
    out['ms_per_step_reference_graphsage_installed'] = round(ms, 3)
    out['reference_graphsage_vs_own'] = round(ms / own_ms, 4)
    del model, opt
    torch.cuda.empty_cache()


def config4(dev, out, steps=60, warmup=10):
    """The captured mini-batch step of ``bench.py --mode minibatch --capture`` on one GPU."""
    import argparse

    import bench
    got = []
    prev = bench.EMIT
    bench.EMIT = got.append
    try:
        ns = argparse.Namespace(scale=1.0, capture=True, steps=steps, warmup=warmup, prefetch=2)
        bench.run_minibatch(ns, 0, 0, 1, dev)
    finally:
        bench.EMIT = prev
    r = got[0]
    cfg = r['config']
    out['config4_minibatch_papers100m'] = {
        'ms_per_step': round(r['ms_per_step'], 4), 'edges_per_s': round(r['value'], 1),
        'alg_GB': round(r['roofline']['algorithmic_bytes_per_batch'] / 1e9, 4),
        'frac_hbm': r['roofline']['frac'],
        'launches_per_batch': cfg.get('launches_per_batch'),
        'real_edges_per_batch': round(sum(cfg['real_edges_per_batch_per_hop']), 1),
        'captured': cfg['captured'][:100], 'graph_build_s': cfg['graph_build_s'],
        'hbm_gb_allocated': cfg['hbm_gb_allocated']}


def run(dev, headline=None, budget_s=40.0):
    """``headline``: (x, ei, train_idx, y_train, num_classes, own_ms) of bench.py's full-batch run,
    or None.  Each leg is skipped (and says so) once ``budget_s`` is spent."""
    from pytorch_geometric_amd import backend
    t0 = time.perf_counter()
    out = {}
    pyg = None
    try:
        from oracle import make_ref
        pyg = make_ref.import_reference()
        backend.install()
    except ImportError as exc:  # no staged reference on this box: own classes only
        out['reference'] = f'unavailable ({exc})'[:100]

    def left():
        return budget_s - (time.perf_counter() - t0)

    try:
        legs = [('config1', lambda: config1(dev, pyg, out)),
                ('config3', lambda: config3(dev, pyg, out)),
                ('config5', lambda: config5(dev, pyg, out))]
        if headline is not None and pyg is not None:
            legs.insert(0, ('config2_reference',
                            lambda: config2_reference(dev, pyg, *headline, out)))
        legs.append(('config4', lambda: config4(dev, out)))
        for name, leg in legs:
            if left() <= 0:
                out[name + '_skipped'] = 'time budget spent'
                continue
            try:
                leg()
            except Exception as exc:
                out[name + '_error'] = f'{type(exc).__name__}: {exc}'[:110]
            torch.cuda.empty_cache()
    finally:
        if pyg is not None:
            backend.uninstall()
    out['seconds'] = round(time.perf_counter() - t0, 1)
    return out


if __name__ == '__main__':  # python bench_configs.py: the block alone (no headline leg)
    import json

    import pytorch_geometric_amd as pga
    pga.load_library()
    print(json.dumps(run(torch.device('cuda:0'), budget_s=120.0), indent=1))
