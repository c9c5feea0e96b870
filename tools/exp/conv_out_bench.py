"""conv_out (320 -> 4 channels, 3x3) and conv_in (4 latent channels zero-padded to 64 -> 320) at the headline batch and for one request: us per launch.
Run twice on one lease:   python tools/exp/conv_out_bench.py ; MX_CONV_SMALL_N=0 MX_CONV_SMALL_CIN=0 python tools/exp/conv_out_bench.py     (0 = the tile kernels)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from sduss_amd import ops  # noqa: E402


def bench(fn, iters=30):
    for _ in range(3):
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
print("MX_CONV_SMALL_N =", os.environ.get("MX_CONV_SMALL_N", "(default: small-N form)"))
for b, hw, cin, cout in ((8, 128, 320, 4), (2, 128, 320, 4), (8, 64, 320, 4), (8, 128, 320, 16)):
    x = torch.randn(b, hw, hw, cin, device="cuda", generator=g).bfloat16()
    w = (torch.randn(cout, 9 * cin, device="cuda", generator=g) * (9 * cin) ** -0.5).bfloat16()
    bias = torch.randn(cout, device="cuda", generator=g)
    t = bench(lambda: ops.conv3x3(x, w, bias))
    print(f"B{b} {hw}x{hw} {cin}->{cout}: {t:7.1f} us   input read at {x.numel() * 2 / t / 1e6:5.2f} TB/s", flush=True)

print("MX_CONV_SMALL_CIN =", os.environ.get("MX_CONV_SMALL_CIN", "(default: small-Cin form)"))
for b, hw in ((8, 128), (2, 128), (8, 64)):
    x = torch.zeros(b, hw, hw, 64, device="cuda", dtype=torch.bfloat16)
    x[..., :4] = torch.randn(b, hw, hw, 4, device="cuda", generator=g).bfloat16()
    w = (torch.randn(320, 9 * 64, device="cuda", generator=g) * 36 ** -0.5).bfloat16()
    bias = torch.randn(320, device="cuda", generator=g)
    t = bench(lambda: ops.conv3x3(x, w, bias, cin_valid=8))
    print(f"conv_in B{b} {hw}x{hw} 4(64)->320: {t:7.1f} us   output written at {b * hw * hw * 320 * 2 / t / 1e6:5.2f} TB/s", flush=True)
