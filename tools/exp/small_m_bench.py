"""M <= 16 GEMMs of the two plans (embedding MLPs, stacked time_emb_proj / AdaLN modulation): us per launch.  Run twice on one lease:
   python tools/exp/small_m_bench.py ; MX_SMALL_M=0 python tools/exp/small_m_bench.py      (0 = the generic 128-row tile kernel)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from sduss_amd import ops  # noqa: E402


def bench(fn, iters=50):
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
print("MX_SMALL_M =", os.environ.get("MX_SMALL_M", "(default: weight-stream form)"))
# (M, N, K, f32 out): SDXL time_embedding.linear_1 / linear_2, add_embedding.linear_1, temb_proj_all; SD3.5 timestep / pooled MLPs, a slice of adaln_all
for m, n, k, f32 in ((8, 1280, 320, False), (8, 1280, 1280, False), (8, 1280, 2816, False), (8, 13760, 1280, True), (2, 1280, 2816, False), (2, 13760, 1280, True),
                     (8, 1536, 256, False), (8, 1536, 2048, False), (8, 1536, 1536, False), (8, 110592, 1536, True), (8, 442368, 1536, True)):
    x = torch.randn(m, k, device="cuda", generator=g).bfloat16()
    w = (torch.randn(n, k, device="cuda", generator=g) * k ** -0.5).bfloat16()
    b = torch.randn(n, device="cuda", generator=g)
    t = bench(lambda: ops.gemm(x, w, b, silu=not f32, out_f32=f32))
    print(f"M{m} N{n} K{k}{' f32' if f32 else ''}: {t:7.1f} us   {n * k * 2 / t / 1e6:6.2f} TB/s of weights", flush=True)
