"""The shipped 256 x 256 kernel's whole-launch rate at long K (where the epilogue is amortised: ~ its K-loop rate) on the SAME box as tools/exp/wave_tile_bench.hip's idealised loops.
Usage on the GPU box: /tmp/wtb && python tools/exp/loop_rate_same_box.py"""
import os, sys, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sduss_amd import ops


def t(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


g = torch.Generator(device="cuda").manual_seed(0)
for (m, n, k) in [(8192, 10240, 1280), (8192, 10240, 6144), (8192, 10240, 12288), (8192, 1280, 12288)]:
    a = torch.randn(m, k, device="cuda", generator=g).bfloat16(); w = (torch.randn(n, k, device="cuda", generator=g) * k ** -0.5).bfloat16()
    bias = torch.randn(n, device="cuda", generator=g)
    dt = statistics.median([t(lambda: ops.gemm(a, w, bias)) for _ in range(5)])
    print(f"shipped kernel, plain epilogue, M{m} N{n} K{k}: {dt * 1e6:8.1f} us = {2.0 * m * n * k / dt / 1e12:6.0f} TFLOP/s", flush=True)
