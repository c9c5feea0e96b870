"""CPU restatement (torch fp32) of the SDXL VAE DECODER as the reference's post_inference runs it.  TEST INFRASTRUCTURE (oracle/__init__.py).
PARITY UNPINNED: the arithmetic lives in un-vendored diffusers 0.32.1 (``AutoencoderKL.decode``), absent from this image.

Call site restated: ``ESyMReDStableDiffusionXLPipeline.post_inference``
(sduss/model_executor/diffusers/pipelines/stable_diffusion_xl/pipeline_stable_diffusion_xl_esymred.py:406-463):
latents are upcast to fp32 (:48-52, :432-433), divided by ``vae.config.scaling_factor`` (:440), decoded, and handed to
``image_processor.postprocess`` (:455).  The decoder is the published diffusers ``Decoder``: post_quant_conv (1x1) -> conv_in ->
UNetMidBlock2D (ResnetBlock2D, Attention with one 512-wide head over GroupNorm'd tokens, ResnetBlock2D) -> four UpDecoderBlock2D
(layers_per_block + 1 resnets without time embedding, nearest-2x upsample + 3x3 conv between blocks) -> GroupNorm -> SiLU -> conv_out.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Tuple

import torch
import torch.nn.functional as F


@dataclass(frozen=True)
class VAEConfig:
    latent_channels: int = 4
    out_channels: int = 3
    block_out_channels: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2
    norm_num_groups: int = 32
    norm_eps: float = 1e-6
    scaling_factor: float = 0.13025          # SDXL vae config.json
    shift_factor: float = 0.0                # SD3: latents / scaling_factor + shift_factor (pipeline_stable_diffusion_3_esymred.py:408)
    use_post_quant_conv: bool = True         # SD3's AutoencoderKL has no quant convs

    @staticmethod
    def sdxl() -> "VAEConfig":
        return VAEConfig()

    @staticmethod
    def tiny() -> "VAEConfig":
        return VAEConfig(block_out_channels=(64, 64, 128), layers_per_block=1)

    @staticmethod
    def tiny_sd3() -> "VAEConfig":
        return VAEConfig(latent_channels=16, block_out_channels=(64, 64, 128), layers_per_block=1, scaling_factor=1.5305, shift_factor=0.0609,
                         use_post_quant_conv=False)


def param_shapes(cfg: VAEConfig) -> Dict[str, Tuple[int, ...]]:
    s: Dict[str, Tuple[int, ...]] = {}
    lc = cfg.latent_channels
    if cfg.use_post_quant_conv:
        s["post_quant_conv.weight"] = (lc, lc, 1, 1); s["post_quant_conv.bias"] = (lc,)
    top = cfg.block_out_channels[-1]
    s["decoder.conv_in.weight"] = (top, lc, 3, 3); s["decoder.conv_in.bias"] = (top,)

    def resnet(p, cin, cout):
        s[f"{p}.norm1.weight"] = (cin,); s[f"{p}.norm1.bias"] = (cin,)
        s[f"{p}.conv1.weight"] = (cout, cin, 3, 3); s[f"{p}.conv1.bias"] = (cout,)
        s[f"{p}.norm2.weight"] = (cout,); s[f"{p}.norm2.bias"] = (cout,)
        s[f"{p}.conv2.weight"] = (cout, cout, 3, 3); s[f"{p}.conv2.bias"] = (cout,)
        if cin != cout:
            s[f"{p}.conv_shortcut.weight"] = (cout, cin, 1, 1); s[f"{p}.conv_shortcut.bias"] = (cout,)
    resnet("decoder.mid_block.resnets.0", top, top)
    a = "decoder.mid_block.attentions.0"
    s[f"{a}.group_norm.weight"] = (top,); s[f"{a}.group_norm.bias"] = (top,)
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        s[f"{a}.{n}.weight"] = (top, top); s[f"{a}.{n}.bias"] = (top,)
    resnet("decoder.mid_block.resnets.1", top, top)
    c = top
    n = len(cfg.block_out_channels)
    for i in range(n):
        cout = cfg.block_out_channels[n - 1 - i]
        for j in range(cfg.layers_per_block + 1):
            resnet(f"decoder.up_blocks.{i}.resnets.{j}", c, cout)
            c = cout
        if i != n - 1:
            s[f"decoder.up_blocks.{i}.upsamplers.0.conv.weight"] = (c, c, 3, 3); s[f"decoder.up_blocks.{i}.upsamplers.0.conv.bias"] = (c,)
    s["decoder.conv_norm_out.weight"] = (c,); s["decoder.conv_norm_out.bias"] = (c,)
    s["decoder.conv_out.weight"] = (cfg.out_channels, c, 3, 3); s["decoder.conv_out.bias"] = (cfg.out_channels,)
    return s


def init_params(cfg: VAEConfig, seed: int = 10086) -> Dict[str, torch.Tensor]:
    """seeded, bf16-representable weights (fan-in scaled) so the HIP side and the oracle share them bit for bit"""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for name, shape in param_shapes(cfg).items():
        if name.endswith(".weight") and len(shape) > 1:
            fan = 1
            for d in shape[1:]:
                fan *= d
            t = torch.randn(shape, generator=g) * fan ** -0.5
        elif name.endswith(".weight"):
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        else:
            t = 0.05 * torch.randn(shape, generator=g)
        out[name] = t.to(torch.bfloat16).to(torch.float32)
    return out


def _resnet(P, p, x, cfg: VAEConfig):
    h = F.silu(F.group_norm(x, cfg.norm_num_groups, P[f"{p}.norm1.weight"], P[f"{p}.norm1.bias"], cfg.norm_eps))
    h = F.conv2d(h, P[f"{p}.conv1.weight"], P[f"{p}.conv1.bias"], padding=1)
    h = F.silu(F.group_norm(h, cfg.norm_num_groups, P[f"{p}.norm2.weight"], P[f"{p}.norm2.bias"], cfg.norm_eps))
    h = F.conv2d(h, P[f"{p}.conv2.weight"], P[f"{p}.conv2.bias"], padding=1)
    if f"{p}.conv_shortcut.weight" in P:
        x = F.conv2d(x, P[f"{p}.conv_shortcut.weight"], P[f"{p}.conv_shortcut.bias"])
    return x + h


def _attention(P, p, x, cfg: VAEConfig):
    b, c, h, w = x.shape
    n = F.group_norm(x, cfg.norm_num_groups, P[f"{p}.group_norm.weight"], P[f"{p}.group_norm.bias"], cfg.norm_eps)
    t = n.reshape(b, c, h * w).transpose(1, 2)                      # [b, L, C]
    q = F.linear(t, P[f"{p}.to_q.weight"], P[f"{p}.to_q.bias"])
    k = F.linear(t, P[f"{p}.to_k.weight"], P[f"{p}.to_k.bias"])
    v = F.linear(t, P[f"{p}.to_v.weight"], P[f"{p}.to_v.bias"])
    a = torch.softmax(q @ k.transpose(1, 2) * c ** -0.5, dim=-1) @ v   # one head of width C
    o = F.linear(a, P[f"{p}.to_out.0.weight"], P[f"{p}.to_out.0.bias"])
    return x + o.transpose(1, 2).reshape(b, c, h, w)


def decode(P: Dict[str, torch.Tensor], cfg: VAEConfig, latents: torch.Tensor) -> torch.Tensor:
    """latents [B, 4, H, W] as the denoising loop leaves them (NOT yet divided by the scaling factor) -> images [B, 3, 8H, 8W]"""
    P = {k: v.to(torch.float32) for k, v in P.items()}
    z = latents.to(torch.float32) / cfg.scaling_factor + cfg.shift_factor  # SDXL :440, SD3 :408
    if cfg.use_post_quant_conv:
        z = F.conv2d(z, P["post_quant_conv.weight"], P["post_quant_conv.bias"])
    x = F.conv2d(z, P["decoder.conv_in.weight"], P["decoder.conv_in.bias"], padding=1)
    x = _resnet(P, "decoder.mid_block.resnets.0", x, cfg)
    x = _attention(P, "decoder.mid_block.attentions.0", x, cfg)
    x = _resnet(P, "decoder.mid_block.resnets.1", x, cfg)
    n = len(cfg.block_out_channels)
    for i in range(n):
        for j in range(cfg.layers_per_block + 1):
            x = _resnet(P, f"decoder.up_blocks.{i}.resnets.{j}", x, cfg)
        if i != n - 1:
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
            x = F.conv2d(x, P[f"decoder.up_blocks.{i}.upsamplers.0.conv.weight"], P[f"decoder.up_blocks.{i}.upsamplers.0.conv.bias"], padding=1)
    x = F.silu(F.group_norm(x, cfg.norm_num_groups, P["decoder.conv_norm_out.weight"], P["decoder.conv_norm_out.bias"], cfg.norm_eps))
    return F.conv2d(x, P["decoder.conv_out.weight"], P["decoder.conv_out.bias"], padding=1)


def postprocess(images: torch.Tensor) -> torch.Tensor:
    """diffusers VaeImageProcessor.postprocess up to the tensor stage: (x / 2 + 0.5).clamp(0, 1)"""
    return (images / 2 + 0.5).clamp(0, 1)
