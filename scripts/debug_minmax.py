import torch, sys
sys.path.insert(0, '.')
import pytorch_geometric_amd as pga
from pytorch_geometric_amd import _native
dev = torch.device('cuda:0')
torch.manual_seed(0)
for F in (16, 8, 100, 4):
    for n in (1, 2, 5, 17, 40, 70):
        x = torch.randn(n, F)
        ptr = torch.tensor([0, n])
        for red in ('max', 'min'):
            out = _native.spmm_csr(ptr.to(dev), None, x.to(dev), red, n_rows=1).cpu()
            ref = x.max(0).values if red == 'max' else x.min(0).values
            bad = (out[0] != ref).nonzero().view(-1).tolist()
            if bad:
                # which row did each wrong column come from?
                src = [(x[:, c] == out[0, c]).nonzero().view(-1).tolist() for c in bad[:8]]
                want = [(x[:, c] == ref[c]).nonzero().view(-1).tolist() for c in bad[:8]]
                print(f'F={F} n={n} {red}: bad cols {bad[:16]} got-from-rows {src} want-rows {want} out {out[0, bad[:4]].tolist()}')
            else:
                print(f'F={F} n={n} {red}: ok')
