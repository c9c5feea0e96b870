"""a few GEMM shapes through whatever library MXDENOISE_LIB points at (ablation builds): prints us and us per K tile per CU"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from sduss_amd import ops
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
def bench(fn, iters=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3
out = []
for m, n, k in ((32768, 1536, 6144), (32768, 4608, 1536), (8192, 10240, 1280)):
    a = torch.randn(m, k, device=dev, generator=g).to(torch.bfloat16)
    w = (torch.randn(n, k, device=dev, generator=g) * k ** -0.5).to(torch.bfloat16)
    t = bench(lambda: ops.gemm(a, w, None))
    per = (m // 256) * (n // 256) * (k // 64) / 256
    out.append(f"M{m} N{n} K{k}: {t:7.1f} us = {t / per:.2f} us per K tile per CU")
print(os.path.basename(os.environ.get("MXDENOISE_LIB", "libmxdenoise.so")), " | ".join(out))
