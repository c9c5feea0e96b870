"""The VAE decoder behind the C ABI (``mx_vae_decode``): what ``post_inference`` calls as ``self.vae.decode``
(sduss/model_executor/diffusers/pipelines/stable_diffusion_xl/pipeline_stable_diffusion_xl_esymred.py:406-463).
SURVEY.md section 8f rank 2 ("next" row): the step after the denoising loop, on the same HIP kernels as the UNet."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Dict, List, Tuple

import torch

from . import lib as _lib
from .weights import PackedWeights, _conv_pack

PAD = 64   # latent channels are zero-padded to one K tile


@dataclass(frozen=True)
class VAEConfig:
    latent_channels: int = 4
    out_channels: int = 3
    block_out_channels: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2
    norm_num_groups: int = 32
    norm_eps: float = 1e-6
    scaling_factor: float = 0.13025
    shift_factor: float = 0.0          # SD3: decode(latents / scaling_factor + shift_factor)
    use_post_quant_conv: bool = True   # SD3's AutoencoderKL has none

    @staticmethod
    def sdxl() -> "VAEConfig":
        return VAEConfig()

    @staticmethod
    def sd3() -> "VAEConfig":
        """the 16-channel VAE of SD3 / SD3.5 (vae/config.json: scaling 1.5305, shift 0.0609, no quant convs); same decoder graph
        (pipeline_stable_diffusion_3_esymred.py:408-409)"""
        return VAEConfig(latent_channels=16, scaling_factor=1.5305, shift_factor=0.0609, use_post_quant_conv=False)

    @staticmethod
    def tiny() -> "VAEConfig":
        return VAEConfig(block_out_channels=(64, 64, 128), layers_per_block=1)

    @staticmethod
    def from_hf_json(path: str) -> "VAEConfig":
        import json
        with open(path) as f:
            c = json.load(f)
        return VAEConfig(latent_channels=c["latent_channels"], out_channels=c["out_channels"], block_out_channels=tuple(c["block_out_channels"]),
                         layers_per_block=c["layers_per_block"], norm_num_groups=c["norm_num_groups"], scaling_factor=c.get("scaling_factor", 0.18215),
                         shift_factor=c.get("shift_factor") or 0.0, use_post_quant_conv=c.get("use_post_quant_conv", True))


def pack_vae(cfg: VAEConfig, P: Dict[str, torch.Tensor]) -> List[Tuple[str, torch.Tensor]]:
    """HF-named AutoencoderKL params (decoder.* and post_quant_conv.*) -> packed tensors of the step plan (csrc/vae_sdxl.cpp)."""
    bf, f32 = torch.bfloat16, torch.float32
    out: List[Tuple[str, torch.Tensor]] = []
    lc = cfg.latent_channels
    # post_quant_conv as a padded [64, 64] linear with the latent rescaling folded in (exact: all of it is affine):
    #   W (z / s + t) + b  =  (W / s) z + (b + W t 1);   without a post_quant_conv (SD3) W = I, b = 0
    w = torch.zeros(PAD, PAD, dtype=f32); b = torch.zeros(PAD, dtype=f32)
    if cfg.use_post_quant_conv:
        w0 = P["post_quant_conv.weight"].reshape(lc, lc).to(f32); b0 = P["post_quant_conv.bias"].to(f32)
    else:
        w0 = torch.eye(lc, dtype=f32); b0 = torch.zeros(lc, dtype=f32)
    w[:lc, :lc] = w0 / cfg.scaling_factor
    b[:lc] = b0 + w0.sum(dim=1) * cfg.shift_factor
    out.append(("post_quant_conv.weight", w.to(bf))); out.append(("post_quant_conv.bias", b))
    for name, t in P.items():
        if not name.startswith("decoder."):
            continue
        if name == "decoder.conv_in.weight":
            out.append((name, _conv_pack(t, PAD).to(bf)))
        elif name == "decoder.conv_out.weight":
            w = _conv_pack(t); pad = (-w.shape[0]) % 4
            out.append((name, torch.nn.functional.pad(w, (0, 0, 0, pad)).to(bf).contiguous()))
        elif name == "decoder.conv_out.bias":
            out.append((name, torch.nn.functional.pad(t.to(f32), (0, (-t.shape[0]) % 4)).contiguous()))
        elif t.ndim == 4 and t.shape[-1] == 3:
            out.append((name, _conv_pack(t).to(bf)))
        elif t.ndim == 4:                                     # 1x1 conv_shortcut
            out.append((name, t.reshape(t.shape[0], t.shape[1]).to(bf).contiguous()))
        elif t.ndim == 2:
            out.append((name, t.to(bf).contiguous()))
        else:
            out.append((name, t.to(f32).contiguous()))
    return out


class MxVAEDecoder:
    """``decode(latents) -> images`` with latents [B, 4 | 16, H, W] as the denoising loop leaves them (``latents / scaling_factor
    (+ shift_factor)`` of post_inference, SDXL :440 / SD3 :408, is folded into the packed weights) and images [B, 3, 8H, 8W] in [-1, 1]."""

    def __init__(self, cfg: VAEConfig, params: Dict[str, torch.Tensor], device="cuda:0", out_dtype=torch.float32):
        self.cfg = cfg
        self.device = torch.device(device)
        self.out_dtype = out_dtype
        self._lib = _lib.load()
        cc = _lib.VAEConfigC()
        cc.latent_channels, cc.out_channels, cc.n_levels = cfg.latent_channels, cfg.out_channels, len(cfg.block_out_channels)
        for i, v in enumerate(cfg.block_out_channels):
            cc.block_out_channels[i] = v
        cc.layers_per_block, cc.norm_num_groups, cc.norm_eps = cfg.layers_per_block, cfg.norm_num_groups, cfg.norm_eps
        self._handle = self._lib.mx_vae_create(C.byref(cc))
        if not self._handle:
            raise _lib.MxError("mx_vae_create: " + self._lib.mx_last_error().decode())
        self.weights = PackedWeights(pack_vae(cfg, params), self.device)
        _lib.check(self._lib.mx_vae_set_weights(self._handle, self.weights.blob.data_ptr(), self.weights.blob.numel(), self.weights.table,
                                                len(self.weights.names)), "mx_vae_set_weights")
        self._ws = None

    def __del__(self):
        h = getattr(self, "_handle", None)
        if h:
            self._lib.mx_vae_destroy(h)
            self._handle = None

    def validate(self, batch: int, h: int, w: int) -> None:
        _lib.check(self._lib.mx_vae_validate(self._handle, batch, h, w), "mx_vae_validate")

    @torch.inference_mode()
    def decode(self, latents: torch.Tensor) -> torch.Tensor:
        x = latents.contiguous()
        b, _c, h, w = x.shape
        need = self._lib.mx_vae_workspace_bytes(self._handle, b, h, w)
        if need == 0:
            raise _lib.MxError("mx_vae_workspace_bytes: " + self._lib.mx_last_error().decode())
        if self._ws is None or self._ws.numel() < need:
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        f = 2 ** (len(self.cfg.block_out_channels) - 1)
        out = torch.empty((b, self.cfg.out_channels, f * h, f * w), dtype=self.out_dtype, device=self.device)
        _lib.check(self._lib.mx_vae_decode(self._handle, _lib.current_stream(), x.data_ptr(), _lib.torch_dtype_code(x.dtype), out.data_ptr(),
                                           _lib.torch_dtype_code(self.out_dtype), b, h, w, self._ws.data_ptr(), self._ws.numel()), "mx_vae_decode")
        return out


def post_inference(vae: MxVAEDecoder, worker_reqs: Dict[str, list]) -> Dict[str, torch.Tensor]:
    """mirror of post_inference (:406-463) up to the tensor stage: gather the finished requests' latents per resolution, decode,
    postprocess to [0, 1]; returns {resolution: images [n, 3, res, res]}"""
    images = {}
    for res, reqs in worker_reqs.items():
        if not reqs:
            continue
        lat = torch.cat([r.latents for r in reqs], dim=0)
        images[res] = (vae.decode(lat) / 2 + 0.5).clamp(0, 1)
    return images
