"""Where the fused q | k | v launch (M 8192, N 3840, K 1280) spends its time beyond a plain bias-only launch of the same shape: same process, preallocated outputs
(ops.gemm_qkv zeroes a fresh V^T buffer per call, which a timing must not include).  Variants: plain | QKV epilogue | QKV epilogue + LayerNorm from finalised statistics."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from sduss_amd import lib, ops  # noqa: E402


def bench(fn, iters=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def main():
    l = lib.load()
    g = torch.Generator(device="cuda").manual_seed(0)
    m, n, k, rpb = 8192, 3840, 1280, 1024
    a = torch.randn(m, k, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(n, k, device="cuda", generator=g) * k ** -0.5).to(torch.bfloat16)
    bias = torch.randn(n, device="cuda", generator=g)
    colsum = w.float().sum(dim=1).contiguous()
    x = a.float()
    fin = torch.stack([x.mean(dim=1), torch.rsqrt(x.var(dim=1, unbiased=False) + 1e-5)], dim=1).contiguous()
    c_plain = torch.empty(m, n, dtype=torch.bfloat16, device="cuda")
    c = torch.empty(m, n // 3 * 2, dtype=torch.bfloat16, device="cuda")
    vt = torch.zeros(m // rpb, n // 3, ops.vt_ld(rpb), dtype=torch.bfloat16, device="cuda")
    stream = lib.current_stream()

    def desc(kind):
        d = lib.GemmDesc()
        d.a, d.w, d.bias = a.data_ptr(), w.data_ptr(), bias.data_ptr()
        d.M, d.N, d.K, d.lda = m, n, k, k
        if kind == "plain":
            d.c, d.ldc = c_plain.data_ptr(), n
        else:
            d.c, d.ldc, d.vt, d.ldvt = c.data_ptr(), c.shape[1], vt.data_ptr(), vt.shape[2]
            d.rows_per_batch, d.flags, d.seg, d.period, d.out_scale = rpb, lib.EPI_QKV, n // 3, 3, ops.ATTN_QSCALE
        if kind in ("qkv_ln", "plain_ln"):
            d.ln_final, d.ln_colsum, d.ln_eps = fin.data_ptr(), colsum.data_ptr(), 1e-5
        return d
    ds = {kd: desc(kd) for kd in ("plain", "plain_ln", "qkv", "qkv_ln")}
    for rep in range(3):
        row = []
        for kd, d in ds.items():
            t = bench(lambda: lib.check(l.mx_gemm(stream, C.byref(d)), "mx_gemm"))
            row.append(f"{kd} {t:6.1f} us")
        print(" | ".join(row))


if __name__ == "__main__":
    main()
