"""Same-box A/B of the fused q | k | v projection (M 8192, N 3840, K 1280; SD3.5: M 32768, N 4608, K 1536): V tiles in the transposed MFMA orientation
(8-byte V^T stores, round 5) against MX_EPI_VT_PLAIN (2-byte stores), and a plain bias-only launch of the same shape."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from sduss_amd import ops  # noqa: E402


def bench(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


g = torch.Generator(device="cuda").manual_seed(0)
for m, n, k, rpb in ((8192, 3840, 1280, 1024), (32768, 4608, 1536, 4096)):
    a = torch.randn(m, k, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(n, k, device="cuda", generator=g) * k ** -0.5).to(torch.bfloat16)
    bias = torch.randn(n, device="cuda", generator=g)
    for rep in range(2):
        t_new = bench(lambda: ops.gemm_qkv(a, w, n // 3, 3, rpb, q_scale=ops.ATTN_QSCALE, bias=bias))
        t_old = bench(lambda: ops.gemm_qkv(a, w, n // 3, 3, rpb, q_scale=ops.ATTN_QSCALE, bias=bias, vt_plain=True)  # (needs the patch))
        t_plain = bench(lambda: ops.gemm(a, w, bias))
        print(f"M{m} N{n} K{k}: transposed V tiles {t_new:7.1f} us | 2-byte V^T stores {t_old:7.1f} us | plain bias-only launch {t_plain:7.1f} us")
