"""The attention tail of a transformer layer (to_out + residual -> to_q with norm2 folded -> 77-key cross-attention -> to_out + residual): ONE chained
launch (mx_attn_tail) against the four separate launches on the same descriptors, back to back on the same box, at the two step shapes of the headline
batch.  Prints per-layer times and checks bit equality.  Usage on the GPU box: python tools/attn_tail_bench.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from sduss_amd import ops  # noqa: E402
from test_attn_tail_gpu import _problem  # noqa: E402


def bench(fn, iters=30, warm=5):
    for _ in range(warm):
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
    for b, heads, L in ((8, 20, 1024), (8, 10, 4096), (16, 20, 1024)):
        _host, args = _problem(b, heads, L)
        ref = ops.attn_tail(**args, chained=False)
        got = ops.attn_tail(**args, chained=True)
        sync = got[5]
        same = torch.equal(ref[0], got[0])
        # (ops.attn_tail allocates its outputs per call: ~4 torch allocations, the same for both forms)
        t_sep = bench(lambda: ops.attn_tail(**args, chained=False))
        t_ch = bench(lambda: ops.attn_tail(**args, chained=True, sync=sync))
        c = heads * 64
        fl = 3 * 2.0 * b * L * c * c + 4.0 * b * L * c * 77
        print(f"B{b} H{heads} L{L} (M {b * L}, C {c}): four launches {t_sep:7.1f} us | chained {t_ch:7.1f} us ({fl / t_ch / 1e6:6.0f} TFLOP/s) | "
              f"{t_sep / t_ch:4.2f}x | bit-equal {same} | status {ops.attn_tail_status(sync):#x}")


if __name__ == "__main__":
    main()
