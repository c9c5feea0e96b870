"""Folded LayerNorm on the 256 x 256 kernel vs the normalisation pass in front of it (round 3): per launch, same box.
   pass:  mx_layernorm (no affine) + GEGLU / QKV-shaped GEMM on the normalised copy
   fold:  the same GEMM on the un-normalised rows with ln_stats = the 16 slabs the producer GEMM (N = 1280, 256 x 160 tiles) left behind
Usage on the GPU box: python tools/exp/ln_fold_256.py
The shipped library keeps ln_stats launches off the 256 x 256 kernel (pick_tile); the measurement in profiles/r03_h_ln_fold_on_256x256.txt was made with
an LN instantiation of it (template flag calling gemm_ln_init at every tile start, epilogue STATS on) that lost or tied and was not kept."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sduss_amd import ops


def t(fn, iters=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


g = torch.Generator(device="cuda").manual_seed(0)
for (m, c, n, geglu) in [(8192, 1280, 10240, True), (8192, 1280, 3840, False), (32768, 640, 5120, True), (32768, 640, 1920, False)]:
    x = torch.randn(m, c, device="cuda", generator=g).bfloat16()
    w0 = (torch.randn(c, c, device="cuda", generator=g) * c ** -0.5).bfloat16(); b0 = torch.randn(c, device="cuda", generator=g)
    res = torch.randn(m, c, device="cuda", generator=g).bfloat16()
    y, st = ops.gemm(x, w0, b0, residual=res, want_stats=True)          # the producer: y and its row statistics
    w = (torch.randn(n, c, device="cuda", generator=g) * c ** -0.5).bfloat16(); bias = torch.randn(n, device="cuda", generator=g)
    cs = w.float().sum(dim=1).contiguous()
    ones, zeros = torch.ones(c, device="cuda"), torch.zeros(c, device="cuda")
    yn = ops.layernorm(y, ones, zeros)
    # interleaved rounds (the box's clock drifts with load): medians of five
    import statistics
    lp, pl, fo = [], [], []
    for _ in range(5):
        lp.append(t(lambda: ops.layernorm(y, ones, zeros), 20))
        pl.append(t(lambda: ops.gemm(yn, w, bias, geglu=geglu), 20))
        fo.append(t(lambda: ops.gemm(y, w, bias, geglu=geglu, ln_stats=st, ln_colsum=cs), 20))
    ln_pass, plain, fold = statistics.median(lp), statistics.median(pl), statistics.median(fo)
    a = ops.gemm(yn, w, bias, geglu=geglu).float(); b = ops.gemm(y, w, bias, geglu=geglu, ln_stats=st, ln_colsum=cs).float()
    rel = ((a - b).norm() / a.norm()).item()
    print(f"M{m} C{c} N{n} {'geglu' if geglu else 'plain'}: LayerNorm pass {ln_pass:6.1f} us + GEMM {plain:7.1f} us = {ln_pass + plain:7.1f} | folded ({st[1]} slabs) {fold:7.1f} us"
          f"   (rel L2 between the two results {rel:.2e})")
    if not geglu and n == 3 * c:                # the fused q | k | v projection with the V^T epilogue
        L = 1024 if c == 1280 else 4096
        q1, q2 = [], []
        for _ in range(5):
            q1.append(t(lambda: ops.gemm_qkv(yn, w, c, 3, L, q_scale=0.125, bias=bias), 20))
            q2.append(t(lambda: ops.gemm_qkv(y, w, c, 3, L, q_scale=0.125, bias=bias, ln_stats=st, ln_colsum=cs), 20))
        print(f"      as to_qkv (V^T epilogue): pass {ln_pass:6.1f} + {statistics.median(q1):7.1f} = {ln_pass + statistics.median(q1):7.1f} | folded {statistics.median(q2):7.1f} us")
