"""Attention microbenchmark over the step shapes (random bf16 data), through the C ABI.
Usage on the GPU box: python tools/attn_bench.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sduss_amd import ops  # noqa: E402

SHAPES = [(8, 10, 4096, 4096), (8, 20, 1024, 1024), (8, 24, 4429, 4429), (8, 20, 1024, 77), (8, 10, 4096, 77)]   # (B, H, Lq, Lk)


if os.environ.get("ATTN_SHAPES") == "cross":      # the short-key launches of a step: 4 requests / 1 request, levels 2 and 1
    SHAPES = [(8, 20, 1024, 77), (2, 20, 1024, 77), (8, 10, 4096, 77), (2, 10, 4096, 77), (8, 20, 576, 77), (8, 20, 256, 77)]


if os.environ.get("ATTN_SHAPES") == "l1024":      # L = 1024 self-attention at workgroup counts around whole rounds of 3 per CU (768): 384, 640, 768, 1280, 1536, 2304, 2560
    SHAPES = [(8, 6, 1024, 1024), (8, 10, 1024, 1024), (8, 12, 1024, 1024), (8, 20, 1024, 1024), (8, 24, 1024, 1024), (8, 36, 1024, 1024), (16, 20, 1024, 1024),
              (8, 20, 2048, 2048), (8, 20, 512, 512)]


def main():
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(0)
    iters = int(os.environ.get("ITERS", "10"))
    for b, h, lq, lk in SHAPES:
        c = h * 64
        q = torch.randn(b * lq, c, device=dev, generator=g).to(torch.bfloat16)
        k = torch.randn(b * lk, c, device=dev, generator=g).to(torch.bfloat16)
        ldvt = ops.vt_ld(lk)
        vt = torch.randn(b, c, ldvt, device=dev, generator=g).to(torch.bfloat16)
        qs = (q.float() * ops.ATTN_QSCALE).to(torch.bfloat16)        # what the producing GEMM epilogue hands to the prescaled entry point
        for pre in (False, True):
            q = qs if pre else q
            for _ in range(2):
                ops.attention(q, k, vt, h, lq, lk, prescaled=pre)
            torch.cuda.synchronize()
            a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(iters):
                ops.attention(q, k, vt, h, lq, lk, prescaled=pre)
            e.record()
            torch.cuda.synchronize()
            t = a.elapsed_time(e) / iters * 1e-3
            fl = 4.0 * b * h * lq * lk * 64
            print(f"attn{' prescaled' if pre else '          '} B{b} H{h} Lq{lq} Lk{lk}  {t * 1e6:9.1f} us  {fl / t / 1e12:7.1f} TFLOP/s")


if __name__ == "__main__":
    main()
