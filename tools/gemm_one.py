"""One GEMM shape, repeated (for rocprofv3 --pmc passes).  Usage: python tools/gemm_one.py M N K [iters] [geglu]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sduss_amd import ops  # noqa: E402

m, n, k = (int(x) for x in sys.argv[1:4])
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 10
geglu = len(sys.argv) > 5
g = torch.Generator(device="cuda:0").manual_seed(0)
a = torch.randn(m, k, device="cuda:0", generator=g).to(torch.bfloat16)
w = (torch.randn(n, k, device="cuda:0", generator=g) * k ** -0.5).to(torch.bfloat16)
bias = torch.randn(n, device="cuda:0", generator=g)
for _ in range(iters):
    ops.gemm(a, w, bias, geglu=geglu)
torch.cuda.synchronize()
