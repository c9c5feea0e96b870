"""Latency of the VAE decoder step plan (mx_vae_decode) at the SDXL widths, random-init weights.
Usage on the GPU box: python tools/vae_bench.py > gpurun_out/vae_bench.log"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sduss_amd.vae import MxVAEDecoder, VAEConfig  # noqa: E402


def shapes(cfg):
    out = {}
    lc, top = cfg.latent_channels, cfg.block_out_channels[-1]
    out["post_quant_conv.weight"] = (lc, lc, 1, 1); out["post_quant_conv.bias"] = (lc,)
    out["decoder.conv_in.weight"] = (top, lc, 3, 3); out["decoder.conv_in.bias"] = (top,)

    def resnet(p, cin, cout):
        out[f"{p}.norm1.weight"] = (cin,); out[f"{p}.norm1.bias"] = (cin,)
        out[f"{p}.conv1.weight"] = (cout, cin, 3, 3); out[f"{p}.conv1.bias"] = (cout,)
        out[f"{p}.norm2.weight"] = (cout,); out[f"{p}.norm2.bias"] = (cout,)
        out[f"{p}.conv2.weight"] = (cout, cout, 3, 3); out[f"{p}.conv2.bias"] = (cout,)
        if cin != cout:
            out[f"{p}.conv_shortcut.weight"] = (cout, cin, 1, 1); out[f"{p}.conv_shortcut.bias"] = (cout,)
    resnet("decoder.mid_block.resnets.0", top, top)
    a = "decoder.mid_block.attentions.0"
    out[f"{a}.group_norm.weight"] = (top,); out[f"{a}.group_norm.bias"] = (top,)
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        out[f"{a}.{n}.weight"] = (top, top); out[f"{a}.{n}.bias"] = (top,)
    resnet("decoder.mid_block.resnets.1", top, top)
    c, n = top, len(cfg.block_out_channels)
    for i in range(n):
        cout = cfg.block_out_channels[n - 1 - i]
        for j in range(cfg.layers_per_block + 1):
            resnet(f"decoder.up_blocks.{i}.resnets.{j}", c, cout); c = cout
        if i != n - 1:
            out[f"decoder.up_blocks.{i}.upsamplers.0.conv.weight"] = (c, c, 3, 3); out[f"decoder.up_blocks.{i}.upsamplers.0.conv.bias"] = (c,)
    out["decoder.conv_norm_out.weight"] = (c,); out["decoder.conv_norm_out.bias"] = (c,)
    out["decoder.conv_out.weight"] = (cfg.out_channels, c, 3, 3); out["decoder.conv_out.bias"] = (cfg.out_channels,)
    return out


def main():
    cfg = VAEConfig.sdxl()
    g = torch.Generator().manual_seed(1)
    P = {}
    for k, s in shapes(cfg).items():
        if len(s) > 1:
            fan = 1
            for d in s[1:]:
                fan *= d
            P[k] = torch.randn(s, generator=g) * fan ** -0.5
        else:
            P[k] = (1.0 if k.endswith("weight") else 0.0) + 0.05 * torch.randn(s, generator=g)
    vae = MxVAEDecoder(cfg, P, device="cuda:0", out_dtype=torch.bfloat16)
    for res, batch in ((512, 1), (1024, 1), (1024, 4)):
        lat = torch.randn(batch, 4, res // 8, res // 8, device="cuda:0", dtype=torch.bfloat16)
        for _ in range(2):
            out = vae.decode(lat)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 5
        for _ in range(n):
            out = vae.decode(lat)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        print(f"vae decode {res}x{res} batch {batch}: {ms:.2f} ms ({ms / batch:.2f} ms/image), finite {bool(torch.isfinite(out.float()).all())}", flush=True)


if __name__ == "__main__":
    main()
