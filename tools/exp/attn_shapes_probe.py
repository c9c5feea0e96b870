"""64-row-per-wave attention kernel vs the 32-row kernels over shapes around the SD3 joint sequence (which one of: ragged tail, head count /
row stride, sequence length decides that L = 4429 x 24 heads gains nothing).  Usage: python tools/exp/attn_shapes_probe.py"""
import os, sys, subprocess, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sduss_amd import ops
SHAPES = [(8, 24, 4429), (8, 24, 4480), (8, 24, 4096), (8, 10, 4429), (8, 10, 4096), (8, 8, 4096), (8, 16, 4096), (8, 32, 4096), (4, 24, 4429), (8, 24, 2048)]
def main():
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(0)
    for b, h, l in SHAPES:
        c = h * 64
        q = (torch.randn(b * l, c, device=dev, generator=g) * ops.ATTN_QSCALE).to(torch.bfloat16)
        k = torch.randn(b * l, c, device=dev, generator=g).to(torch.bfloat16)
        vt = torch.randn(b, c, ops.vt_ld(l), device=dev, generator=g).to(torch.bfloat16)
        for _ in range(2): ops.attention(q, k, vt, h, l, l, prescaled=True)
        torch.cuda.synchronize()
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10): ops.attention(q, k, vt, h, l, l, prescaled=True)
        e.record(); torch.cuda.synchronize()
        t = a.elapsed_time(e) / 10 * 1e-3
        print(f"B{b} H{h} L{l}: {t * 1e6:8.1f} us {4.0 * b * h * l * l * 64 / t / 1e12:7.1f} TFLOP/s", flush=True)
main()
