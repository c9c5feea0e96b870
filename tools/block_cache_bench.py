"""What the block-skip cache costs and saves at SDXL width (mx_unet_forward_cached, random-init weights, synthetic inputs).
Per batch size: the exact step (mx_unet_forward, graph replay), the cached entry with every block run (the price of the seven
comparisons, the host round trips and the un-captured launch sequence), with the up blocks reused, and with every block reused.
Usage on the GPU box: python tools/block_cache_bench.py > gpurun_out/block_cache_bench.log"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sduss_amd.block_cache import BlockSkipCache  # noqa: E402
from sduss_amd.config import UNetConfig  # noqa: E402
from sduss_amd.unet import MxUNet  # noqa: E402
from sduss_amd.weights import synthetic_params  # noqa: E402


class Blocks:
    """run exactly the blocks of `mask`"""
    def __init__(self, mask):
        self.mask = mask

    def predict(self, f):
        f = np.asarray(f)
        return np.full(len(f), (self.mask >> int(f[0, 0])) & 1)


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    dev = torch.device("cuda:0")
    cfg = UNetConfig.sdxl_base()
    net = MxUNet(cfg, synthetic_params(cfg, device=dev), device=dev)
    print(f"{'batch':>5s} {'px':>5s} {'exact ms':>9s} {'all run':>9s} {'down+mid':>9s} {'down only':>9s} {'none':>9s}")
    for batch, px in ((2, 1024), (8, 1024), (2, 512), (16, 512)):
        hw = px // 8
        g = torch.Generator(device=dev).manual_seed(batch)
        s = torch.randn(batch, 4, hw, hw, device=dev, generator=g).to(torch.bfloat16)
        t = torch.full((batch,), 801.0, device=dev)
        e = torch.randn(batch, 77, cfg.cross_attention_dim, device=dev, generator=g).to(torch.bfloat16)
        te = torch.randn(batch, cfg.text_embed_dim, device=dev, generator=g).to(torch.bfloat16)
        ti = torch.tensor([[px, px, 0, 0, px, px]], device=dev, dtype=torch.float32).repeat(batch, 1)
        row = [timed(lambda: net.forward_one(s, t, e, te, ti))]
        for mask in (0x7f, 0x0f, 0x07, 0x00):
            pred = Blocks(0x7f)
            bc = BlockSkipCache(pred, forced_after=1 << 30)      # the timing wants the same decision every step
            net.forward_one_cached(bc, s, t, e, te, ti, batch_key=1)
            pred.mask = mask
            row.append(timed(lambda: net.forward_one_cached(bc, s, t, e, te, ti, batch_key=1)))
            assert bc.history[-1] == mask
        print(f"{batch:5d} {px:5d} " + " ".join(f"{v:9.2f}" for v in row), flush=True)
        state_mb = bc.state.numel() / 2 ** 20
        print(f"      cache state {state_mb:.0f} MiB", flush=True)


def main_sd3():
    from sduss_amd.block_cache import FORCED_RUN_AFTER_SD3  # noqa: F401
    from sduss_amd.config import MMDiTConfig
    from sduss_amd.transformer_sd3 import MxSD3Transformer
    from sduss_amd.weights import synthetic_mmdit_params
    dev = torch.device("cuda:0")
    cfg = MMDiTConfig.sd35_medium()
    net = MxSD3Transformer(cfg, synthetic_mmdit_params(cfg, device=dev), device=dev)
    n = cfg.num_layers
    full = (1 << n) - 1
    print(f"SD3.5-medium, {n} blocks")
    print(f"{'batch':>5s} {'px':>5s} {'exact ms':>9s} {'all run':>9s} {'even only':>9s} {'none':>9s}")
    for batch, px in ((2, 1024), (8, 1024), (2, 512)):
        hw = px // 8
        g = torch.Generator(device=dev).manual_seed(batch)
        lat = torch.randn(batch, cfg.in_channels, hw, hw, device=dev, generator=g).to(torch.bfloat16)
        t = torch.full((batch,), 801.0, device=dev)
        e = torch.randn(batch, 333, cfg.joint_attention_dim, device=dev, generator=g).to(torch.bfloat16)
        p = torch.randn(batch, cfg.pooled_projection_dim, device=dev, generator=g).to(torch.bfloat16)
        row = [timed(lambda: net.forward_one(lat, t, e, p))]
        for mask in (full, full & 0x555555555555, 0):
            pred = Blocks(full)
            bc = BlockSkipCache(pred, forced_after=1 << 30)
            net.forward_one(lat, t, e, p, cache=bc, batch_key=1)
            pred.mask = mask
            row.append(timed(lambda: net.forward_one(lat, t, e, p, cache=bc, batch_key=1)))
            assert bc.history[-1] == mask
        print(f"{batch:5d} {px:5d} " + " ".join(f"{v:9.2f}" for v in row), flush=True)
        print(f"      cache state {bc.state.numel() / 2 ** 20:.0f} MiB", flush=True)
        del bc
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
    main_sd3()
