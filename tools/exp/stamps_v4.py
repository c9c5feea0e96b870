"""Diagnostic: where a K tile of the ping-pong 256x256 GEMM spends its shader cycles (MX_EXP=7 build, s_memtime sums per wave).
   MXDENOISE_LIB=build/exp/libmx_v4e7.so python tools/exp/stamps_v4.py M N K"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from sduss_amd import lib, ops  # noqa: E402

m, n, k = (int(x) for x in sys.argv[1:4])
g = torch.Generator(device="cuda:0").manual_seed(0)
a = torch.randn(m, k, device="cuda:0", generator=g).to(torch.bfloat16)
w = (torch.randn(n, k, device="cuda:0", generator=g) * k ** -0.5).to(torch.bfloat16)
for _ in range(5):
    ops.gemm(a, w, None)
torch.cuda.synchronize()
l = lib.load()
buf = np.zeros(256 * 8 * 16, dtype=np.uint64)
fn = l.mx_debug_v4_sums
fn.argtypes = [C.c_void_p]
assert fn(buf.ctypes.data) == 0
s = buf.reshape(256, 8, 16)[:, :, :8].astype(np.float64)
tiles = (m + 255) // 256 * (n // 256)
kt_per_cu = tiles / 256 * (k // 64)
names = ["LA issue (16 reads, 4 DMA)", "barrier + fragment wait", "MA (32 MFMA)", "barrier", "LB issue (8 reads, 4 DMA, vmcnt)", "barrier + fragment wait", "MB (32 MFMA)", "barrier -> next LA"]
print(f"M{m} N{n} K{k}: {kt_per_cu:.0f} K tiles per CU; mean shader-clock ticks per K tile (s_memtime), by wave row")
for row, waves in (("row 0 (waves 0-3)", slice(0, 4)), ("row 1 (waves 4-7)", slice(4, 8))):
    per = s[:, waves, :].mean(axis=(0, 1)) / kt_per_cu
    print(f"  {row}: total {per.sum():7.0f}")
    for nm, v in zip(names, per):
        print(f"     {nm:34s} {v:7.0f}")
