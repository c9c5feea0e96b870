"""The inner-boundary op (esymred_mp.groupnorm / mock_groupnorm, norm_silu_concat.cu) at the reference's shapes: SDXL 1024^2, 4 requests
under CFG, patch 256 px -> 128 patches of 32 x 32 (level 0, C 320), 16 x 16 (C 640), 8 x 8 (C 1280); fp16 as the reference runs it.
Prints time and the fraction of the 8 TB/s HBM peak (algorithmic bytes: input read twice -- moments, apply -- and the padded output
written once).  Usage on the GPU box: python tools/gn_halo_bench.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import patch_ref  # noqa: E402  (index construction only)
from sduss_amd import esymred_mp  # noqa: E402

HBM_PEAK = 8.0e12


def bench(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


def main():
    dev = "cuda:0"
    for dtype in (torch.float16, torch.bfloat16):
        for c, hw_img, p in ((320, 128, 32), (640, 64, 16), (1280, 32, 8)):
            nlat = 8
            samples = {str(hw_img * 8): torch.zeros(nlat, 1, hw_img, hw_img)}
            pidx, lat_off, _ro, _patches, pmap = patch_ref.split_sample(samples, p * 8)
            n = pmap.numel()
            x = torch.randn(n, c, p, p, device=dev).to(dtype)
            ga = torch.randn(c, device=dev).to(dtype); be = torch.randn(c, device=dev).to(dtype)
            lo = torch.tensor(lat_off, dtype=torch.int32, device=dev); pm = pmap.to(dev); pi = pidx.to(dev)
            es = x.element_size()
            t = bench(lambda: esymred_mp.groupnorm(x, ga, be, n, c, p, p, c // 32, 1e-5, True, lo, pm, pi))
            by = x.numel() * es * 2 + n * c * (p + 2) * (p + 2) * es
            t2 = bench(lambda: esymred_mp.mock_groupnorm(x, n, c, p, p, c // 32, pi))
            by2 = x.numel() * es + n * c * (p + 2) * (p + 2) * es
            print(f"{str(dtype):15s} N{n} C{c} {p}x{p}: groupnorm+halo {t * 1e6:8.1f} us {by / t / 1e9:7.0f} GB/s ({by / t / HBM_PEAK:.3f} of HBM peak) | "
                  f"halo only {t2 * 1e6:8.1f} us {by2 / t2 / 1e9:7.0f} GB/s ({by2 / t2 / HBM_PEAK:.3f})")


if __name__ == "__main__":
    main()
