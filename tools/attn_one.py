"""One attention shape, repeated (for rocprofv3 --pmc passes).  Usage: python tools/attn_one.py B H Lq Lk [iters]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sduss_amd import ops  # noqa: E402

b, h, lq, lk = (int(x) for x in sys.argv[1:5])
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 5
g = torch.Generator(device="cuda:0").manual_seed(0)
c = h * 64
q = torch.randn(b * lq, c, device="cuda:0", generator=g).to(torch.bfloat16)
k = torch.randn(b * lk, c, device="cuda:0", generator=g).to(torch.bfloat16)
vt = torch.randn(b, c, ops.vt_ld(lk), device="cuda:0", generator=g).to(torch.bfloat16)
for _ in range(iters):
    ops.attention(q, k, vt, h, lq, lk)
torch.cuda.synchronize()
