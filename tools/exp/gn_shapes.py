"""NHWC GroupNorm(+SiLU) at the SDXL step's shapes (batch 8 and 2): time per call and HBM rate (3 passes over the tensor).
(the MX_GN_TILE_PIX switch of the round-2 A/B is gone from the library: the tile is 256 pixels, halved until the launch has two blocks per CU).  Usage: python tools/exp/gn_shapes.py"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sduss_amd import lib
l = lib.load()
for b in (8, 2):
    tot = 0.0
    for (hw, c, n) in [(128, 320, 9), (128, 640, 2), (128, 960, 1), (64, 640, 9), (64, 320, 1), (64, 1280, 2), (64, 1920, 1), (64, 960, 1), (32, 1280, 12), (32, 640, 1), (32, 2560, 2), (32, 1920, 1)]:
        x = torch.randn(b, hw, hw, c, device="cuda").bfloat16(); y = torch.empty_like(x)
        g = torch.ones(c, device="cuda"); be = torch.zeros(c, device="cuda")
        ws = torch.empty(l.mx_groupnorm_nhwc_workspace_bytes(b, hw, hw, c), dtype=torch.uint8, device="cuda")
        f = lambda: lib.check(l.mx_groupnorm_nhwc(lib.current_stream(), x.data_ptr(), y.data_ptr(), g.data_ptr(), be.data_ptr(), b, hw, hw, c, 32, 1e-5, 1, 0, ws.data_ptr()))
        for _ in range(3): f()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20): f()
        e.record(); torch.cuda.synchronize()
        us = s.elapsed_time(e) / 20 * 1e3
        tot += us * n
        print(f"B{b} {hw}x{hw}x{c}: {us:7.1f} us  {3 * x.numel() * 2 / us / 1e6:6.2f} TB/s  (x{n} per step)")
    print(f"B{b}: {tot / 1e3:.2f} ms per step in GroupNorm")
