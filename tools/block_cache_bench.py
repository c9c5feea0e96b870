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


class Fraction:
    """ask a fixed fraction of the patches of every block, the same ones every step: the first round(f * n) rows of the call (so whole samples
    drop out first, the way a predictor with per-request thresholds behaves), or, spread=True, every k-th patch (every sample keeps some asking
    patches: what costs the per-sample attention problems their batching)"""
    def __init__(self, f, spread=False):
        self.f, self.spread = f, spread

    def predict(self, feats):
        feats = np.asarray(feats)
        n = len(feats)
        out = np.zeros(n, dtype=np.int64)
        k = int(round(self.f * n))
        if self.spread and k > 0:
            out[np.unique(np.linspace(0, n - 1, k).round().astype(int))] = 1
        else:
            out[:k] = 1
        out[feats[:, 2] > 1e18] = 1
        return out


def main_patch_unit():
    """The cache at the reference's unit (mx_unet_forward_cached_mixed, is_sliced=True / patch 256): step time against the fraction of patches that
    ask, SDXL-base.  `exact` = mx_unet_forward_mixed of the same batch (sliced, no cache, graph replay).  Every block runs with the given asking
    fraction: the convolutions, attn1's core + to_out and the whole attn2 scale with it; GroupNorm, LayerNorms, q|k|v, the feed-forward and the
    state copies do not (as in the reference)."""
    from sduss_amd.block_cache import PatchSkipCache
    dev = torch.device("cuda:0")
    cfg = UNetConfig.sdxl_base()
    net = MxUNet(cfg, synthetic_params(cfg, device=dev), device=dev)
    print("patch-unit cache (is_sliced=True, patch 256 px): ms per forward against the asking fraction")
    print(f"{'batch':>14s} {'exact':>8s} " + " ".join(f"{'f=' + str(f):>8s}" for f in (1.0, 0.75, 0.5, 0.25, 0.125, 0.0)) + "   (f = 0.5 spread over all samples)   state MiB")
    for comp in (((2, 1024),), ((8, 1024),), ((2, 512), (2, 768), (2, 1024))):
        g = torch.Generator(device=dev).manual_seed(3)
        xs = [torch.randn(b, 4, px // 8, px // 8, device=dev, generator=g).to(torch.bfloat16) for b, px in comp]
        btot = sum(b for b, _ in comp)
        t = torch.full((btot,), 801.0, device=dev)
        e = torch.randn(btot, 77, cfg.cross_attention_dim, device=dev, generator=g).to(torch.bfloat16)
        te = torch.randn(btot, cfg.text_embed_dim, device=dev, generator=g).to(torch.bfloat16)
        ti = torch.cat([torch.tensor([[px, px, 0, 0, px, px]], device=dev, dtype=torch.float32).repeat(b, 1) for b, px in comp])
        ids = [f"r{i}" for i in range(btot)]
        row = [timed(lambda: net.forward_mixed(xs, t, e, te, ti, gn_patch=32))]
        for f, spread in ((1.0, False), (0.75, False), (0.5, False), (0.25, False), (0.125, False), (0.0, False), (0.5, True)):
            pred = Fraction(1.0)
            pc = PatchSkipCache(pred, forced_after=1 << 30)
            net.forward_mixed_cached(pc, xs, ids, t, e, te, ti, gn_patch=32)
            pred.f, pred.spread = f, spread
            row.append(timed(lambda: net.forward_mixed_cached(pc, xs, ids, t, e, te, ti, gn_patch=32), n=6))
            state = pc.state.numel() / 2 ** 20
            del pc
            torch.cuda.empty_cache()
        name = "+".join(f"{b}x{px}" for b, px in comp)
        print(f"{name:>14s} " + " ".join(f"{v:8.2f}" for v in row[:7]) + f"   {row[7]:8.2f}   {state:8.0f}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "patch":
        main_patch_unit()
    else:
        main()
        main_sd3()
        main_patch_unit()
