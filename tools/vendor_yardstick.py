"""Same-box yardstick: the vendor libraries on the hot shapes of the SDXL / SD3.5 step, next to this library's own kernels.

Test / bench infrastructure only -- nothing under sduss_amd/ imports this file, and the product path never calls torch.matmul or SDPA.
For each of the six GEMM shapes that carry ~70 % of the SDXL step it times, in the same process and on the same data:
  * torch's linear (hipBLASLt / rocBLAS, bf16, bias + residual as separate torch ops AND as a bare matmul: the bare figure is the yardstick,
    the library's own kernel also pays for its fused epilogue),
  * mx_gemm through the C ABI (ops.gemm with the same epilogue the step uses: bias + residual, or GEGLU),
and for the three attention shapes F.scaled_dot_product_attention (all backends torch enables on this build) next to mx_attention.
Usage on the GPU box: python tools/vendor_yardstick.py [> profiles/rNN_vendor_yardstick.txt]"""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sduss_amd import ops  # noqa: E402

GEMMS = [  # (label, M, N, K, kind)
    ("ff.net.0 GEGLU     ", 8192, 10240, 1280, "geglu"),
    ("to_out / to_q      ", 8192, 1280, 1280, "res"),
    ("ff.net.2           ", 8192, 1280, 5120, "res"),
    ("to_qkv             ", 8192, 3840, 1280, "plain"),
    ("ff.net.0 GEGLU @64 ", 32768, 5120, 640, "geglu"),
    ("to_out / to_q @64  ", 32768, 640, 640, "res"),
]
ATTN = [(8, 20, 1024, 1024), (8, 10, 4096, 4096), (8, 24, 4429, 4429), (8, 20, 1024, 77)]   # (B, H, Lq, Lk)


def bench(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


def main():
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(0)
    print(f"# torch {torch.__version__}  device {torch.cuda.get_device_name(0)}")
    print(f"# preferred BLAS backend: {torch.backends.cuda.preferred_blas_library()}")
    print("# GEMM: us per launch (TFLOP/s).  vendor bare = torch.matmul(a, w.T) alone; vendor + epilogue = F.linear + residual add (or GEGLU) as torch ops")
    print(f"{'shape':20s} {'M':>6s} {'N':>6s} {'K':>6s} | {'mx (fused epilogue)':>22s} | {'vendor bare':>20s} | {'vendor + epilogue':>20s} | mx / vendor-bare time")
    for label, m, n, k, kind in GEMMS:
        a = torch.randn(m, k, device=dev, generator=g).to(torch.bfloat16)
        w = (torch.randn(n, k, device=dev, generator=g) * k ** -0.5).to(torch.bfloat16)
        bias = torch.randn(n, device=dev, generator=g)
        bias16 = bias.to(torch.bfloat16)
        fl = 2.0 * m * n * k
        if kind == "geglu":
            t_mx = bench(lambda: ops.gemm(a, w, bias, geglu=True))

            def vendor_full():
                y = F.linear(a, w, bias16)
                h, gate = y.chunk(2, dim=-1)
                return h * F.gelu(gate)
        elif kind == "res":
            r = torch.randn(m, n, device=dev, generator=g).to(torch.bfloat16)
            t_mx = bench(lambda: ops.gemm(a, w, bias, residual=r))

            def vendor_full():
                return F.linear(a, w, bias16) + r
        else:
            t_mx = bench(lambda: ops.gemm(a, w, bias))

            def vendor_full():
                return F.linear(a, w, bias16)
        wt = w.t()
        t_bare = bench(lambda: torch.matmul(a, wt))
        t_full = bench(vendor_full)
        f = lambda t: f"{t * 1e6:8.1f} us ({fl / t / 1e12:6.0f})"   # noqa: E731
        print(f"{label:20s} {m:6d} {n:6d} {k:6d} | {f(t_mx):>22s} | {f(t_bare):>20s} | {f(t_full):>20s} | {t_mx / t_bare:5.2f}")

    print("# attention: us per launch (TFLOP/s, 4 B H Lq Lk 64 FLOP).  vendor = F.scaled_dot_product_attention on (B, H, L, 64) bf16")
    for b, h, lq, lk in ATTN:
        c = h * 64
        q = torch.randn(b * lq, c, device=dev, generator=g).to(torch.bfloat16)
        kk = torch.randn(b * lk, c, device=dev, generator=g).to(torch.bfloat16)
        v = torch.randn(b * lk, c, device=dev, generator=g).to(torch.bfloat16)
        vt = ops.pack_vt(v.view(b, lk, c))
        qs = (q.float() * ops.ATTN_QSCALE).to(torch.bfloat16)
        t_mx = bench(lambda: ops.attention(qs, kk, vt, h, lq, lk, prescaled=True), iters=10)
        q4 = q.view(b, lq, h, 64).transpose(1, 2).contiguous()
        k4 = kk.view(b, lk, h, 64).transpose(1, 2).contiguous()
        v4 = v.view(b, lk, h, 64).transpose(1, 2).contiguous()
        fl = 4.0 * b * h * lq * lk * 64
        try:
            t_v = bench(lambda: F.scaled_dot_product_attention(q4, k4, v4), iters=10)
            vs = f"{t_v * 1e6:8.1f} us ({fl / t_v / 1e12:6.0f})"
            ratio = f"{t_mx / t_v:5.2f}"
        except Exception as e:  # noqa: BLE001 -- a backend torch cannot run on this build is reported, not fatal
            vs, ratio = f"failed: {type(e).__name__}", "  n/a"
        print(f"attn B{b} H{h:2d} Lq{lq:5d} Lk{lk:5d} | mx {t_mx * 1e6:8.1f} us ({fl / t_mx / 1e12:6.0f}) | vendor SDPA {vs} | mx / vendor time {ratio}")


if __name__ == "__main__":
    main()
