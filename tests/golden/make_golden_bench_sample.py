"""Cached reference results of ONE bench step at 1/128 of the products shape, for the parity leg
bench.py runs on rank 0 when world > 1 (no time for a live CPU run of the reference there).

Run in the build container (imports the real reference from /root/reference):
    python tests/golden/make_golden_bench_sample.py
Inputs are NOT stored: `products_like(seed=1, scale=1/128)` and `torch.manual_seed(0)` +
`GraphSAGE(100, 256, 3, c)` reproduce them bit for bit (checksums below catch a drift).  Stored:
the reference's fp32 loss, 1024 output rows, and — per parameter tensor — the fp64 gradient (rounded
to fp32: 6e-8 relative, far below the 2e-5 tolerance) and the distance of the reference's OWN fp32
gradient from it, which is the yardstick of `parity_at_cpu_scale`."""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
SCALE = 1 / 128


def checksum(t: torch.Tensor) -> float:
    return float(t.double().abs().sum())


def main():
    sys.path.insert(0, os.environ.get('PYG_REFERENCE', '/root/reference'))
    from torch_geometric.nn import GraphSAGE as RefSAGE
    from pytorch_geometric_amd.datasets import products_like
    x, y, ei, c = products_like(seed=1, scale=SCALE)
    g = torch.Generator().manual_seed(7)
    train_idx = torch.randperm(x.size(0), generator=g)[:max(int(0.0803 * x.size(0)), 1)]
    torch.manual_seed(0)
    model = RefSAGE(100, 256, num_layers=3, out_channels=c)
    state = {k: v.detach().clone() for k, v in model.state_dict().items()}
    out = model(x, ei)
    loss = F.cross_entropy(out[train_idx], y[train_idx])
    loss.backward()
    grads = {k: p.grad.clone() for k, p in model.named_parameters()}
    m64 = RefSAGE(100, 256, num_layers=3, out_channels=c).double()
    m64.load_state_dict({k: v.double() for k, v in state.items()})
    F.cross_entropy(m64(x.double(), ei)[train_idx], y[train_idx]).backward()
    g64 = {k: p.grad.clone() for k, p in m64.named_parameters()}
    ref_vs_64 = {k: float((grads[k].double() - g64[k]).abs().max() / g64[k].abs().max())
                 for k in grads}
    ref_vs_64_l2 = {k: float((grads[k].double() - g64[k]).norm() / g64[k].norm()) for k in grads}
    rows = torch.arange(0, x.size(0), max(x.size(0) // 1024, 1))[:1024]
    blob = {
        'scale': SCALE, 'classes': c, 'n': x.size(0), 'e': ei.size(1),
        'checksums': {'x': checksum(x), 'ei': float(ei.double().sum()),
                      'state': sum(checksum(v) for v in state.values())},
        'loss': loss.detach(), 'rows': rows, 'out_rows': out.detach()[rows].clone(),
        'out_absmax': float(out.detach().abs().max()),
        'grads64_as_f32': {k: v.float() for k, v in g64.items()},
        'ref_vs_fp64': ref_vs_64, 'ref_vs_fp64_l2': ref_vs_64_l2,
        'made_with': f'torch {torch.__version__}, torch_geometric (reference) GraphSAGE, CPU',
    }
    path = os.path.join(ROOT, 'tests', 'golden', 'golden_bench_sample_v1.pt')
    torch.save(blob, path)
    print(path, os.path.getsize(path), 'bytes; ref_vs_fp64 max', max(ref_vs_64.values()))


if __name__ == '__main__':
    main()
