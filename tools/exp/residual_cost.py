"""What the residual costs a single-round 256 x 160 launch (round 4): the same GEMM with and without the residual operand, and with the row statistics / their
finalisation on top (to_out / ff.net.2 as the step launches them).  "residual (no prefetch)" / "residual" differed in an experimental build whose two past-the-end
LDS-DMA groups fetched the tile's residual (into LDS stages nobody reads: an L2 prefetch for the epilogue's loads) instead of the zero page -- 0.5-1.7 us per launch,
nothing on the step (profiles/r04_u_residual_prefetch.txt); not kept, with the shipped library both columns time the same launch.  The "+stats" columns include a
1-MB torch fill of the statistics buffer per call.
Usage on the GPU box: python tools/exp/residual_cost.py"""
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
for (m, n, k) in [(8192, 1280, 1280), (8192, 1280, 5120), (32768, 640, 640), (32768, 640, 2560)]:
    x = torch.randn(m, k, device="cuda", generator=g).bfloat16()
    w = (torch.randn(n, k, device="cuda", generator=g) * k ** -0.5).bfloat16(); b = torch.randn(n, device="cuda", generator=g)
    res = torch.randn(m, n, device="cuda", generator=g).bfloat16()
    fb = (torch.zeros(m, 2, device="cuda"), torch.zeros((m + 255) // 256, dtype=torch.int32, device="cuda"))
    rows = {"plain": [], "residual (no prefetch)": [], "residual": [], "residual+stats": [], "residual+stats+final": []}
    for _ in range(5):
        rows["plain"].append(t(lambda: ops.gemm(x, w, b)))
        os.environ["MX_NO_RES_PF"] = "1"
        rows["residual (no prefetch)"].append(t(lambda: ops.gemm(x, w, b, residual=res)))
        os.environ.pop("MX_NO_RES_PF")
        rows["residual"].append(t(lambda: ops.gemm(x, w, b, residual=res)))
        rows["residual+stats"].append(t(lambda: ops.gemm(x, w, b, residual=res, want_stats=True)))
        rows["residual+stats+final"].append(t(lambda: ops.gemm(x, w, b, residual=res, want_stats=True, want_final=True, final_buffers=fb)))
    print(f"M{m} N{n} K{k}: " + " | ".join(f"{kk} {statistics.median(v):6.1f} us" for kk, v in rows.items()), flush=True)
