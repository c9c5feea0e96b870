"""Split-K microbenchmark: the one-request GEMM shapes of the SDXL step, timed through mx_gemm unsplit (mx_gemm_desc.splitk = 1), split in two
and in four (forced), and with the library's own choice.  Usage: python tools/exp/splitk_bench.py"""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sduss_amd import lib, ops  # noqa: E402


def main():
    l = lib.load()
    dev = torch.device("cuda:0")
    print(f"{'M':>6s} {'N':>6s} {'K':>6s}  {'unsplit us':>10s} {'2 slices':>9s} {'4 slices':>9s}   auto (slices, us)")
    for m, n, k, res in ((2048, 1280, 1280, True), (2048, 1280, 5120, True), (2048, 3840, 1280, False), (8192, 640, 2560, True),
                         (512, 1280, 1280, True), (512, 1280, 5120, True), (1152, 1280, 5120, True), (3712, 1280, 1280, True), (3712, 1280, 5120, True)):
        a = torch.randn(m, k, device=dev).to(torch.bfloat16); w = (torch.randn(n, k, device=dev) * k ** -0.5).to(torch.bfloat16)
        b = torch.randn(n, device=dev); r = torch.randn(m, n, device=dev).to(torch.bfloat16) if res else None
        d = lib.GemmDesc()
        d.M, d.N, d.K, d.lda, d.ldc, d.ldr = m, n, k, k, n, n
        d.a = d.w = d.c = 4096
        d.residual = 4096 if res else 0

        def t(sk):
            for _ in range(5):
                ops.gemm(a, w, b, residual=r, splitk=sk)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(200):
                ops.gemm(a, w, b, residual=r, splitk=sk)
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / 200 * 1e6
        auto = l.mx_gemm_splitk(C.byref(d), 0)
        print(f"{m:6d} {n:6d} {k:6d}  {t(1):10.1f} {t(2):9.1f} {t(4):9.1f}   {auto} {t(0):7.1f}", flush=True)


if __name__ == "__main__":
    main()
