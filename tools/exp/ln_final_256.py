"""Folded LayerNorm from FINALISED row statistics on the 256 x 256 kernel (round 4) vs the normalisation pass in front of the plain launch: per launch, same box.
   pass:   mx_layernorm (no affine) + GEGLU / QKV-shaped GEMM on the normalised copy;  producer = the N = C GEMM with stats_out
   final:  producer with stats_out + ln_final_out (its last workgroup per 256-row panel folds the slabs) + the same consumer with ln_final
Usage on the GPU box: python tools/exp/ln_final_256.py"""
import os, sys, statistics, torch
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
for (m, c, n, geglu, kp) in [(8192, 1280, 10240, True, 1280), (8192, 1280, 3840, False, 5120), (32768, 640, 5120, True, 640), (32768, 640, 1920, False, 2560)]:
    xin = torch.randn(m, kp, device="cuda", generator=g).bfloat16()
    w0 = (torch.randn(c, kp, device="cuda", generator=g) * kp ** -0.5).bfloat16(); b0 = torch.randn(c, device="cuda", generator=g)
    res = torch.randn(m, c, device="cuda", generator=g).bfloat16()
    y, st, fin = ops.gemm(xin, w0, b0, residual=res, want_stats=True, want_final=True)
    w = (torch.randn(n, c, device="cuda", generator=g) * c ** -0.5).bfloat16(); bias = torch.randn(n, device="cuda", generator=g)
    cs = w.float().sum(dim=1).contiguous()
    if fin is None:
        print(f"M{m} C{c}: the producer cannot finalise"); continue
    if n % 256 != 0:
        print(f"M{m} C{c} N{n}: the consumer does not run on the 256 x 256 kernel (it reads the slabs)"); continue
    yf = y.float()
    want_mean, want_rstd = yf.mean(dim=1), (yf.var(dim=1, unbiased=False) + 1e-5).rsqrt()
    e1 = ((fin[:, 0] - want_mean).abs().max() / want_mean.abs().max()).item(); e2 = ((fin[:, 1] - want_rstd).abs().max() / want_rstd.abs().max()).item()
    ones, zeros = torch.ones(c, device="cuda"), torch.zeros(c, device="cuda")
    yn = ops.layernorm(y, ones, zeros)
    qkv = (not geglu) and n == 3 * c
    L = 1024 if c == 1280 else 4096
    if qkv:
        plain_fn = lambda: ops.gemm_qkv(yn, w, c, 3, L, q_scale=0.125, bias=bias)
        final_fn = lambda: ops.gemm_qkv(y, w, c, 3, L, q_scale=0.125, bias=bias, ln_final=fin, ln_colsum=cs)
    else:
        plain_fn = lambda: ops.gemm(yn, w, bias, geglu=geglu)
        final_fn = lambda: ops.gemm(y, w, bias, geglu=geglu, ln_final=fin, ln_colsum=cs)
    prod_plain = lambda: ops.gemm(xin, w0, b0, residual=res, want_stats=True)
    fb = (torch.zeros(m, 2, device="cuda"), torch.zeros((m + 255) // 256, dtype=torch.int32, device="cuda"))
    prod_final = lambda: ops.gemm(xin, w0, b0, residual=res, want_stats=True, want_final=True, final_buffers=fb)
    lp, pl, fo, pp, pf = [], [], [], [], []
    for _ in range(5):
        lp.append(t(lambda: ops.layernorm(y, ones, zeros), 20)); pl.append(t(plain_fn, 20)); fo.append(t(final_fn, 20))
        pp.append(t(prod_plain, 20)); pf.append(t(prod_final, 20))
    med = statistics.median
    a = plain_fn(); b = final_fn()
    a = (a[0] if isinstance(a, tuple) else a).float(); b = (b[0] if isinstance(b, tuple) else b).float()
    rel = ((a - b).norm() / a.norm()).item()
    print(f"M{m} C{c} N{n} {'geglu' if geglu else 'qkv' if qkv else 'plain'}: pass {med(lp):6.1f} + GEMM {med(pl):7.1f} = {med(lp) + med(pl):7.1f} us | final {med(fo):7.1f} us | "
          f"producer (K {kp}; the statistics buffer's allocation in both) {med(pp):6.1f} -> {med(pf):6.1f} us | fin err mean {e1:.1e} rstd {e2:.1e} | rel L2 pass vs final {rel:.2e}", flush=True)
