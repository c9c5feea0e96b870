"""Diagnostic: where the chained attention-tail launch (mx_attn_tail) spends its wall clock, per work item (MX_TAIL_STAMPS build, s_memrealtime at 100 MHz:
ticket taken, wait for the panel's previous stage over, stage body issued, stores drained + workgroup barrier, signal sent).
   build: tools/exp/build_tail_stamps.sh ; run: MXDENOISE_LIB=sduss_amd/libmxdenoise_tailstamps.so python tools/exp/tail_timeline.py
Prints per stage: when its items start and end (relative to the launch's first ticket), the wait before the body, the body, the drain, next to the
launch's duration by hipEvents and the four separate launches'."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from sduss_amd import lib, ops  # noqa: E402
from test_attn_tail_gpu import _problem  # noqa: E402

STAGES = ["to_out1", "to_q", "xattn", "to_out2"]


def main():
    l = lib.load()
    fn = l.mx_debug_tail_stamps
    fn.argtypes = [C.c_void_p]
    for b, heads, L in ((8, 20, 1024), (8, 10, 4096)):
        _host, args = _problem(b, heads, L)
        got = ops.attn_tail(**args, chained=True)
        sync = got[5]
        for _ in range(3):
            ops.attn_tail(**args, chained=True, sync=sync)
        torch.cuda.synchronize()
        buf = np.zeros(256 * 16 * 6, dtype=np.uint64)
        assert fn(buf.ctypes.data) == 0
        st = buf.reshape(256, 16, 6).astype(np.int64)
        t0 = st[:, 0, 1].min()
        us = lambda x: (x - t0) / 100.0
        print(f"B{b} H{heads} L{L}: workgroups' first tickets spread over {us(st[:, 0, 1].max()):.2f} us")
        n_items = (st[:, :, 1] > 0).sum(axis=1)
        print(f"  items per workgroup: min {n_items.min()} max {n_items.max()}")
        for s_, name in enumerate(STAGES):
            sel = [(w, i) for w in range(256) for i in range(16) if st[w, i, 1] > 0 and ((st[w, i, 0] >> 8) & 0xff) == s_]
            if not sel:
                continue
            a = np.array([[us(st[w, i, k]) for k in range(1, 6)] for w, i in sel])
            wait, body, drain, sig = a[:, 1] - a[:, 0], a[:, 2] - a[:, 1], a[:, 3] - a[:, 2], a[:, 4] - a[:, 3]
            print(f"  {name:8s} {len(sel):4d} items: ticket at {a[:, 0].mean():7.2f} (min {a[:, 0].min():7.2f} max {a[:, 0].max():7.2f}) | wait {wait.mean():5.2f} (max {wait.max():5.2f}) | "
                  f"body {body.mean():6.2f} (min {body.min():6.2f} max {body.max():6.2f}) | drain + barrier {drain.mean():5.2f} (max {drain.max():5.2f}) | signal {sig.mean():4.2f} | "
                  f"ends at {a[:, 4].mean():7.2f} (max {a[:, 4].max():7.2f})")


if __name__ == "__main__":
    main()
