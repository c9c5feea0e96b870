"""GroupNorm + SiLU of a resnet's conv1 output: the three-launch form (statistics pass, fold, apply) against the form that takes the statistics as partial sums the conv left
(per 64 rows and channel: fold, apply), per call, same box (round 4).  Usage on the GPU box: python tools/exp/gn_partials_bench.py"""
import os, sys, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sduss_amd import ops


def t(fn, iters=40):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


g = torch.Generator(device="cuda").manual_seed(0)
for (b, hw, c) in [(8, 128, 320), (8, 64, 640), (8, 32, 1280)]:
    x = torch.randn(b, hw, hw, c, device="cuda", generator=g).bfloat16()
    ga, be = torch.ones(c, device="cuda"), torch.zeros(c, device="cuda")
    xf = x.float().reshape(-1, 64, c)
    part = torch.stack([xf.sum(1), (xf * xf).sum(1)], dim=-1).contiguous()
    y0 = ops.groupnorm_nhwc(x, ga, be, 32, 1e-5, True, 0).float(); y1 = ops.groupnorm_nhwc_from_partials(x, ga, be, 32, 1e-5, True, part).float()
    d = ((y0 - y1).abs().max() / y0.abs().max()).item()
    a3, a2 = [], []
    for _ in range(5):
        a3.append(t(lambda: ops.groupnorm_nhwc(x, ga, be, 32, 1e-5, True, 0))); a2.append(t(lambda: ops.groupnorm_nhwc_from_partials(x, ga, be, 32, 1e-5, True, part)))
    print(f"B{b} {hw}x{hw} C{c}: statistics + fold + apply {statistics.median(a3):6.1f} us | fold + apply from {b * hw * hw // 64} x {c} partials {statistics.median(a2):6.1f} us  (max diff {d:.1e}; both include a torch.empty workspace)", flush=True)
