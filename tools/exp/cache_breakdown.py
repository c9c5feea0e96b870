"""Where the patch-unit block cache's overhead goes (verdict item 4: "make it profitable or prove it cannot be").  One process per mode, each under
rocprofv3 --kernel-trace --stats (tools/exp/cache_breakdown.sh): `exact` = mx_unet_forward_mixed of 8 x 1024^2 (sliced, no cache); `f1.0` / `f0.5` = the cached
entry with that fraction of the patches asking in every block.  Prints wall ms per forward (host clock, fenced) so that wall - summed kernel time = what
the host round trips cost (the GPU idles while the decision travels)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from block_cache_bench import Fraction  # noqa: E402
from sduss_amd.block_cache import PatchSkipCache  # noqa: E402
from sduss_amd.config import UNetConfig  # noqa: E402
from sduss_amd.unet import MxUNet  # noqa: E402
from sduss_amd.weights import synthetic_params  # noqa: E402


def main():
    mode = sys.argv[1]
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    dev = torch.device("cuda:0")
    cfg = UNetConfig.sdxl_base()
    net = MxUNet(cfg, synthetic_params(cfg, device=dev), device=dev)
    g = torch.Generator(device=dev).manual_seed(3)
    xs = [torch.randn(8, 4, 128, 128, device=dev, generator=g).to(torch.bfloat16)]
    t = torch.full((8,), 801.0, device=dev)
    e = torch.randn(8, 77, cfg.cross_attention_dim, device=dev, generator=g).to(torch.bfloat16)
    te = torch.randn(8, cfg.text_embed_dim, device=dev, generator=g).to(torch.bfloat16)
    ti = torch.tensor([[1024, 1024, 0, 0, 1024, 1024]], device=dev, dtype=torch.float32).repeat(8, 1)
    ids = [f"r{i}" for i in range(8)]
    if mode == "exact":
        fn = lambda: net.forward_mixed(xs, t, e, te, ti, gn_patch=32)  # noqa: E731
    else:
        pred = Fraction(1.0)
        pc = PatchSkipCache(pred, forced_after=1 << 30)
        net.forward_mixed_cached(pc, xs, ids, t, e, te, ti, gn_patch=32)
        pred.f = float(mode[1:])
        fn = lambda: net.forward_mixed_cached(pc, xs, ids, t, e, te, ti, gn_patch=32)  # noqa: E731
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    print(f"WALL {mode} {(time.perf_counter() - t0) / n * 1e3:.2f} ms per forward over {n} forwards (+ 2 or 3 warm-up forwards in the kernel statistics)")


if __name__ == "__main__":
    main()
