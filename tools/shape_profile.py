"""Per-shape kernel timing of one SDXL step (B=4 requests, 1024^2) from the library's per-launch hipEvent records.
Usage on the GPU box: python tools/shape_profile.py > gpurun_out/shapes.log"""
import ctypes as C
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sduss_amd import lib  # noqa: E402
from sduss_amd.config import UNetConfig  # noqa: E402
from sduss_amd.pipeline import SDXLDenoiser, synthetic_request  # noqa: E402
from sduss_amd.unet import MxUNet  # noqa: E402
from sduss_amd.weights import synthetic_params  # noqa: E402

KINDS = ["gemm128", "gemm64", "conv128", "conv64", "attn", "gnorm", "gemm_v2_160", "conv_v2_160", "gemm_v2_128", "conv_v2_128", "gemm_256", "attn_cross", "attn_tail"]


def main():
    """argv[1]: requests at 1024 px (default 4), or a mix "a/b/c" of 512 / 768 / 1024 px requests; argv[2] = "serial": the mix as one launch
    sequence per resolution, back to back (default: ONE mixed sequence)"""
    arg = sys.argv[1] if len(sys.argv) > 1 else "4"
    mix = tuple(int(v) for v in arg.split("/")) if "/" in arg else (0, 0, int(arg))
    dev = torch.device("cuda:0")
    cfg = UNetConfig.sdxl_base()
    net = MxUNet(cfg, synthetic_params(cfg, device=dev), device=dev)
    den = SDXLDenoiser(net)
    if len(sys.argv) > 2 and sys.argv[2] == "serial":
        net.mixed_one_sequence = False
        den.concurrent_resolutions = False
    shared = {}
    rid = 0
    batch = {}
    for res, n in zip((512, 768, 1024), mix):
        if n:
            batch[str(res)] = [synthetic_request(rid + i, res, 50, cfg, den, dev, shared=shared) for i in range(n)]
            rid += n
    for _ in range(2):
        den.denoising_step(batch)
    torch.cuda.synchronize()
    l = lib.load()
    l.mx_profile_enable(1)
    den.denoising_step(batch)
    torch.cuda.synchronize()
    buf = (C.c_double * 64)()
    lib.check(l.mx_profile_collect(buf))
    l.mx_profile_enable(0)
    rec = (C.c_double * (6 * 4096))()
    n = l.mx_profile_records(rec, 4096)
    agg = collections.OrderedDict()
    for i in range(n):
        kind, m, nn, k, ms, fl = (rec[6 * i + j] for j in range(6))
        key = (KINDS[int(kind)], int(m), int(nn), int(k))
        a = agg.setdefault(key, [0, 0.0, 0.0])
        a[0] += 1; a[1] += ms; a[2] += fl
    tot = sum(a[1] for a in agg.values())
    print(f"{'kernel':11s} {'M':>7s} {'N':>6s} {'K':>6s} {'n':>4s} {'ms':>8s} {'%':>6s} {'us/launch':>10s} {'TFLOP/s':>8s}")
    for (kind, m, nn, k), (cnt, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        tf = fl / (ms * 1e-3) / 1e12 if fl else 0.0
        print(f"{kind:11s} {m:7d} {nn:6d} {k:6d} {cnt:4d} {ms:8.3f} {100 * ms / tot:6.2f} {1e3 * ms / cnt:10.1f} {tf:8.1f}")
    print(f"total profiled {tot:.2f} ms")
    per_kind = collections.OrderedDict()
    for (kind, _m, _n, _k), (cnt, ms, _fl) in agg.items():
        a = per_kind.setdefault(kind, [0, 0.0]); a[0] += cnt; a[1] += ms
    print("per kind: " + ", ".join(f"{k} {v[0]} launches {v[1]:.2f} ms" for k, v in per_kind.items()))


if __name__ == "__main__":
    main()
