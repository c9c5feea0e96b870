"""attn2.to_q + 77-key cross-attention: the fused launch (MX_EPI_XATTN, round 4) against the two launches it replaces, per layer, same box; and their difference.
The fused form is NOT in the shipped library: it was correct on its first run (max difference 3e-3 of range against the two launches, NaN-filled V^T pads
survived) and tied -- 50.5 us against 48.8 at the headline batch (160 tiles of 256 x 256 leave 96 CUs idle, and the in-kernel attention is VALU-bound like
the standalone one), 44-46 us against 25-33 at one or two requests.  The complete change is kept as tools/exp/xattn_fused.patch (apply with `git apply`
on the commit that added this file to re-run); result: profiles/r04_w_xattn_fused.txt.
Usage on the GPU box (patched build): python tools/exp/xattn_fused.py"""
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
for (bsz, heads, L, lk) in [(8, 20, 1024, 77), (4, 20, 1024, 77), (2, 20, 1024, 77)]:
    c = heads * 64
    m = bsz * L
    x = (torch.randn(m, c, device="cuda", generator=g)).bfloat16()
    w = (torch.randn(c, c, device="cuda", generator=g) * c ** -0.5).bfloat16()
    k = torch.randn(bsz * lk, c, device="cuda", generator=g).bfloat16()
    v = torch.randn(bsz, lk, c, device="cuda", generator=g).bfloat16()
    vt = ops.pack_vt(v, pad=float("nan"))
    two = lambda: ops.attention(ops.gemm(x, w, None, out_scale=ops.ATTN_QSCALE), k, vt, heads, L, lk, prescaled=True)
    one = lambda: ops.gemm_cross_attention(x, w, None, k, vt, L, lk)
    a, b = two().float(), one().float()
    err = ((a - b).abs().max() / a.abs().max()).item()
    t2, t1 = [], []
    for _ in range(5):
        t2.append(t(two)); t1.append(t(one))
    print(f"B{bsz} H{heads} L{L} Lk{lk}: to_q + attention {statistics.median(t2):6.1f} us | fused {statistics.median(t1):6.1f} us | max diff {err:.2e} of range, finite {bool(torch.isfinite(b).all())}", flush=True)
