"""conv_out (320 -> 4 channels, 3x3) at the headline batch and for one request: us per launch.  Run twice on one lease:
   python tools/exp/conv_out_bench.py ; MX_CONV_SMALL_N=0 python tools/exp/conv_out_bench.py     (0 = the generic tile kernel)"""
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
