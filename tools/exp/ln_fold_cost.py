"""What the folded-LayerNorm hooks cost per launch: plain GEMM vs producer (stats_out) vs consumer (ln_stats), same shapes, same box.
Usage on the GPU box: python tools/exp/ln_fold_cost.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sduss_amd import ops
from sduss_amd.weights import _geglu_interleave

def t(fn, iters=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3

g = torch.Generator(device="cuda").manual_seed(0)
for (m, n, k, geglu) in [(2048, 1280, 5120, False), (2048, 1280, 1280, False), (8192, 1280, 1280, False), (8192, 1280, 5120, False), (8192, 10240, 1280, True), (2048, 10240, 1280, True),
                         (8192, 3840, 1280, False), (2048, 3840, 1280, False)]:
    a = torch.randn(m, k, device="cuda", generator=g).bfloat16(); w = (torch.randn(n, k, device="cuda", generator=g) * k ** -0.5).bfloat16()
    bias = torch.randn(n, device="cuda", generator=g); res = torch.randn(m, n // 2 if geglu else n, device="cuda", generator=g).bfloat16()
    cs = torch.randn(n, device="cuda", generator=g)
    st = ops.row_stats(a)
    if geglu:
        plain = t(lambda: ops.gemm(a, w, bias, geglu=True))
        cons = t(lambda: ops.gemm(a, w, bias, geglu=True, ln_stats=st, ln_colsum=cs))
        print(f"M{m} N{n} K{k} geglu: plain {plain:7.1f} us | ln consumer (1 slab) {cons:7.1f} us")
    else:
        plain = t(lambda: ops.gemm(a, w, bias, residual=res))
        prod = t(lambda: ops.gemm(a, w, bias, residual=res, want_stats=True))
        y, st16 = ops.gemm(a, w, bias, residual=res, want_stats=True)
        cons1 = t(lambda: ops.gemm(a, w, bias, ln_stats=st, ln_colsum=cs))
        print(f"M{m} N{n} K{k}: plain+res {plain:7.1f} us | producer (stats_out, {st16[1]} slabs; includes the torch.full of the buffer) {prod:7.1f} us | ln consumer (1 slab) {cons1:7.1f} us")
        if n == k:
            cons16 = t(lambda: ops.gemm(y, w, bias, ln_stats=st16, ln_colsum=cs))
            print(f"      ln consumer ({st16[1]} slabs) {cons16:7.1f} us")
