"""Diagnostic: where a launch of the ping-pong GEMMs spends its wall clock OUTSIDE the K loop (MX_EXP=8 build, s_memrealtime stamps at
100 MHz per workgroup: kernel entry, first K tile, end of the K loop, end of the epilogue's issue).
   build: tools/exp/build_timeline.sh ; run: MXDENOISE_LIB=build/exp/libmx_exp8.so python tools/exp/timeline_v4.py
Prints, per shape: the spread of the workgroups' start times, the time to the first K tile, the K loop and the epilogue per tile, and
where the last workgroup ends -- next to the launch's duration by hipEvents."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from sduss_amd import lib, ops  # noqa: E402

SHAPES = [  # (label, M, N, K, geglu, residual)  -- label starting with "QKV": the fused q | k | v projection with its V^T epilogue
    ("QKV epilogue (256x256)", 8192, 3840, 1280, False, False),
    ("to_qkv  (256x256)", 8192, 3840, 1280, False, False),
    ("LN to_qkv plain epilogue, LayerNorm from finalised statistics (256x256)", 8192, 3840, 1280, False, False),
    ("LN QKV epilogue + LayerNorm from finalised statistics (256x256)", 8192, 3840, 1280, False, False),
    ("GEGLU   (256x256)", 8192, 10240, 1280, True, False),
    ("GEGLU 64x64 level (256x256)", 32768, 5120, 640, True, False),
    ("ff.out  (256x160)", 8192, 1280, 5120, False, True),
    ("to_out  (256x160)", 8192, 1280, 1280, False, True),
    ("to_out  no residual", 8192, 1280, 1280, False, False),
]


def event_us(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def main():
    l = lib.load()
    g = torch.Generator(device="cuda:0").manual_seed(0)
    only_v4 = len(sys.argv) > 1 and sys.argv[1] == "v4"
    for label, m, n, k, geglu, res in SHAPES:
        if only_v4 and not (n % 256 == 0 and n >= 2560):
            continue
        a = torch.randn(m, k, device="cuda:0", generator=g).to(torch.bfloat16)
        w = (torch.randn(n, k, device="cuda:0", generator=g) * k ** -0.5).to(torch.bfloat16)
        bias = torch.randn(n, device="cuda:0", generator=g)
        r = torch.randn(m, n, device="cuda:0", generator=g).to(torch.bfloat16) if res else None
        run = lambda: ops.gemm(a, w, bias, geglu=geglu, residual=r)  # noqa: E731
        if label.startswith("QKV"):
            run = lambda: ops.gemm_qkv(a, w, n // 3, 3, 1024, q_scale=0.125, bias=bias)  # noqa: E731
        if label.startswith("LN"):
            x = a.float()
            fin = torch.stack([x.mean(dim=1), torch.rsqrt(x.var(dim=1, unbiased=False) + 1e-5)], dim=1).contiguous()
            colsum = w.float().sum(dim=1).contiguous()
            if "QKV" in label:
                run = lambda: ops.gemm_qkv(a, w, n // 3, 3, 1024, q_scale=0.125, bias=bias, ln_final=fin, ln_colsum=colsum)  # noqa: E731
            else:
                run = lambda: ops.gemm(a, w, bias, ln_final=fin, ln_colsum=colsum)  # noqa: E731
        us = event_us(run)
        torch.cuda.synchronize()
        run(); torch.cuda.synchronize()
        v4 = n % 256 == 0 and n >= 2560
        print(f"{label}: M{m} N{n} K{k}  {us:.1f} us per launch (hipEvents, back to back)")
        if v4:
            buf = np.zeros(256 * 2 * 64, dtype=np.uint64)
            fn = l.mx_debug_v4_stamps; fn.argtypes = [C.c_void_p]
            assert fn(buf.ctypes.data) == 0
            raw = buf.reshape(256, 2, 64).astype(np.float64)
            st = raw / 100.0
            tiles = (m + 255) // 256 * (n // 256)
            per_cu = (tiles + 255) // 256
            for wv, name in ((0, "wave 0 (row 0)"), (1, "wave 7 (row 1)")):
                t0 = st[:, wv, 0].min()
                print(f"  {name}: workgroup starts spread over {st[:, wv, 0].max() - t0:.2f} us")
                for t in range(per_cu):
                    ok = st[:, wv, 3 + 3 * t] > 0
                    if not ok.any():
                        break
                    k0 = st[ok, wv, 1 + 3 * t] - t0; k1 = st[ok, wv, 2 + 3 * t] - t0; e = st[ok, wv, 3 + 3 * t] - t0
                    if 3 + 3 * t >= 32:
                        break
                    # shader clock over the K loop: s_memtime ticks (slots 32..) per microsecond of s_memrealtime
                    ck = (raw[ok, wv, 32 + 2 + 3 * t] - raw[ok, wv, 32 + 1 + 3 * t]) / np.maximum(k1 - k0, 1e-9)
                    print(f"    tile {t} (n={int(ok.sum()):3d}): first K tile at {np.median(k0):7.2f} (min {k0.min():6.2f} max {k0.max():6.2f})  K loop {np.median(k1 - k0):6.2f} "
                          f"(max {np.max(k1 - k0):6.2f})  epilogue issue {np.median(e - k1):5.2f} (max {np.max(e - k1):5.2f})  ends at {np.median(e):7.2f} (max {e.max():7.2f})  shader clock in the K loop {np.median(ck):6.0f} MHz")
        else:
            buf = np.zeros(1024 * 2 * 4, dtype=np.uint64)
            fn = l.mx_debug_v5_stamps; fn.argtypes = [C.c_void_p]
            assert fn(buf.ctypes.data) == 0
            tiles = min(1024, (m + 255) // 256 * (n // 160))
            st = buf.reshape(1024, 2, 4)[:tiles].astype(np.float64) / 100.0
            for wv, name in ((0, "wave 0 (group A)"), (1, "wave 7 (group B)")):
                t0 = st[:, wv, 0].min()
                s0 = st[:, wv, 0] - t0; k0 = st[:, wv, 1] - t0; k1 = st[:, wv, 2] - t0; e = st[:, wv, 3] - t0
                print(f"  {name}: workgroup starts spread over {s0.max():.2f} us (median {np.median(s0):.2f}); first K tile at {np.median(k0):.2f} (max {k0.max():.2f}); "
                      f"K loop {np.median(k1 - k0):.2f} (min {np.min(k1 - k0):.2f} max {np.max(k1 - k0):.2f}); epilogue issue {np.median(e - k1):.2f} (max {np.max(e - k1):.2f}); "
                      f"ends at {np.median(e):.2f} (max {e.max():.2f})")


if __name__ == "__main__":
    main()
