"""77-key cross-attention, per launch, same box: the short-key kernel at 64 / 128 / 256 queries per wave (MX_XQPW, an experiment hook of round 4)
and the general register-staged kernel (MX_XQPW=0).  The hook lived in an experimental build of attention.hip (template parameter QPW of
attn_cross_kernel, chosen per launch) that was not kept: profiles/r04_n_cross_attn_bench.txt.  With the shipped library every column times the same launch."""
import os, sys, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sduss_amd import ops


def t(fn, iters=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


g = torch.Generator(device="cuda").manual_seed(0)
shapes = [(8, 20, 1024, 77), (8, 10, 4096, 77), (2, 20, 1024, 77), (2, 10, 4096, 77), (4, 20, 576, 77), (12, 20, 1024, 77)]
for (b, h, lq, lk) in shapes:
    c = h * 64
    q = (torch.randn(b * lq, c, device="cuda", generator=g) * 0.3).bfloat16()
    k = torch.randn(b * lk, c, device="cuda", generator=g).bfloat16()
    v = torch.randn(b, lk, c, device="cuda", generator=g).bfloat16()
    vt = ops.pack_vt(v)
    base = None
    row = []
    for mode in ["0", "64", "128", "256", "auto"]:
        if mode == "auto": os.environ.pop("MX_XQPW", None)
        else: os.environ["MX_XQPW"] = mode
        o = ops.attention(q, k, vt, h, lq, lk, prescaled=True).float()
        if base is None: base = o
        d = ((o - base).abs().max() / base.abs().max()).item()
        us = statistics.median([t(lambda: ops.attention(q, k, vt, h, lq, lk, prescaled=True)) for _ in range(5)])
        row.append(f"{mode}: {us:6.1f} us (d {d:.1e})")
    os.environ.pop("MX_XQPW", None)
    print(f"B{b} H{h} Lq{lq} Lk{lk}  Q+O {4 * b * lq * c / 1e6:6.1f} MB | " + " | ".join(row), flush=True)
