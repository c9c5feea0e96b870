"""Diagnostic: per-tile time line of the persistent 256x256 GEMM (MX_EXP=7 build with s_memrealtime stamps).
   build:  hipcc ... -DMX_EXP=7 (tools/exp/build_stamps.sh) ; run: MXDENOISE_LIB=build/exp/libmx_exp7.so python tools/exp/stamps_v3.py M N K [geglu]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from sduss_amd import lib, ops  # noqa: E402

m, n, k = (int(x) for x in sys.argv[1:4])
geglu = len(sys.argv) > 4 and sys.argv[4] == "geglu"
res = len(sys.argv) > 4 and sys.argv[4] == "res"
g = torch.Generator(device="cuda:0").manual_seed(0)
a = torch.randn(m, k, device="cuda:0", generator=g).to(torch.bfloat16)
w = (torch.randn(n, k, device="cuda:0", generator=g) * k ** -0.5).to(torch.bfloat16)
bias = torch.randn(n, device="cuda:0", generator=g)
r = torch.randn(m, n, device="cuda:0", generator=g).to(torch.bfloat16) if res else None
for _ in range(5):
    ops.gemm(a, w, bias, geglu=geglu, residual=r)
torch.cuda.synchronize()
l = lib.load()
buf = np.zeros(256 * 2 * 64, dtype=np.uint64)
fn = l.mx_debug_v3_stamps
fn.argtypes = [C.c_void_p]
assert fn(buf.ctypes.data) == 0
st = buf.reshape(256, 2, 64).astype(np.float64) / 100.0     # microseconds
tiles = (m + 255) // 256 * (n // 256)
per_cu = (tiles + 255) // 256
t0 = st[:, :, 0].min()
print(f"M{m} N{n} K{k} tiles {tiles} ({per_cu} per CU), nk {k // 64}; all times in us relative to the first workgroup start")
for wv in (0, 1):
    print(f" wave {'0' if wv == 0 else '7'}:")
    print(f"  kernel start spread: {st[:, wv, 0].max() - t0:.2f}")
    for t in range(min(per_cu, 12)):
        s_k0 = st[:, wv, 1 + 3 * t] - t0; s_k1 = st[:, wv, 2 + 3 * t] - t0; s_e = st[:, wv, 3 + 3 * t] - t0
        ok = st[:, wv, 3 + 3 * t] > 0
        if not ok.any():
            break
        print(f"  tile {t}: first barrier at {np.median(s_k0[ok]):8.2f} (spread {s_k0[ok].max() - s_k0[ok].min():5.2f})  K loop {np.median((s_k1 - s_k0)[ok]):7.2f}  "
              f"epilogue {np.median((s_e - s_k1)[ok]):6.2f} (max {((s_e - s_k1)[ok]).max():6.2f})   n={ok.sum()}")
    last = st[:, wv, :].max(axis=1) - t0
    print(f"  last stamp: median {np.median(last):.2f} max {last.max():.2f}")
