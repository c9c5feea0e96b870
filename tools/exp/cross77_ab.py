"""The 77-key cross-attention: the general register-staged kernel (what mx_attention_prescaled takes below Lq 2048) against the short-key kernel in its compile-time
Lk = 77 form (mx_attention_cross_prescaled), same process, preallocated operands."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from sduss_amd import lib, ops  # noqa: E402


def bench(fn, iters=50):
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


l = lib.load()
g = torch.Generator(device="cuda").manual_seed(0)
for b, h, lq, lk in ((8, 20, 1024, 77), (8, 10, 4096, 77), (2, 20, 1024, 77), (8, 20, 1024, 64)):
    c = h * 64
    q = (torch.randn(b * lq, c, device="cuda", generator=g) * ops.ATTN_QSCALE).to(torch.bfloat16)
    k = torch.randn(b * lk, c, device="cuda", generator=g).to(torch.bfloat16)
    vt = ops.pack_vt(torch.randn(b, lk, c, device="cuda", generator=g).to(torch.bfloat16))
    o1, o2 = torch.empty_like(q), torch.empty_like(q)
    st = lib.current_stream()
    args = lambda o: (st, q.data_ptr(), c, k.data_ptr(), c, vt.data_ptr(), vt.shape[2], vt.shape[1] * vt.shape[2], o.data_ptr(), c, b, h, lq, lk)
    for rep in range(2):
        t_auto = bench(lambda: lib.check(l.mx_attention_prescaled(*args(o1)), "attn"))
        t_cross = bench(lambda: lib.check(l.mx_attention_cross_prescaled(*args(o2)), "cross"))
    d = (o1.float() - o2.float()).abs().max().item() / o1.float().abs().max().item()
    print(f"B{b} H{h} Lq{lq} Lk{lk}: mx_attention_prescaled {t_auto:6.1f} us | short-key kernel (forced) {t_cross:6.1f} us | max diff {d:.1e} of range")
