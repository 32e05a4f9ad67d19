"""One weight-gradient shape, a few launches per schedule (for counter passes):
python scripts/wgrad_one.py [rows K1 K2 N]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_geometric_amd import _native  # noqa: E402

rows, K1, K2, N = (int(a) for a in sys.argv[1:5]) if len(sys.argv) >= 5 else (2_449_029, 256, 256, 256)
dev = torch.device('cuda:0')
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(rows, K1, device=dev, generator=g)
x2 = torch.randn(rows, K2, device=dev, generator=g)
go = torch.randn(rows, N, device=dev, generator=g)
out = torch.empty(N, K1 + K2, device=dev)
_native.set_gemm_mode('split')
for variant in (1, 0):
    _native.lab_set_wgrad_variant(variant)
    for _ in range(3):
        _native.linear_wgrad(go, x, out=out, bias_grad=True, x2=x2)
    torch.cuda.synchronize()
_native.lab_set_wgrad_variant(0)
