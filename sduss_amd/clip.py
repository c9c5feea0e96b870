"""CLIP text encoders behind the C ABI (``mx_clip_encode``): what diffusers' ``encode_prompt`` runs for ``prepare_inference``
(sduss/model_executor/diffusers/pipelines/stable_diffusion_xl/pipeline_stable_diffusion_xl_esymred.py:118-140) -- SDXL's
``text_encoder`` (CLIP ViT-L) and ``text_encoder_2`` (OpenCLIP bigG with projection).  SURVEY.md section 8f rank 2.  The tokenizers stay on
the host (``transformers.CLIPTokenizer``); this module takes token ids."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import torch

from . import lib as _lib
from .weights import PackedWeights


@dataclass(frozen=True)
class CLIPTextConfig:
    vocab_size: int = 49408
    hidden_size: int = 768
    intermediate_size: int = 3072
    num_hidden_layers: int = 12
    num_attention_heads: int = 12
    max_position_embeddings: int = 77
    hidden_act: str = "quick_gelu"
    projection_dim: int = 0            # 0: CLIPTextModel (no text_projection)
    eos_token_id: int = 2
    layer_norm_eps: float = 1e-5
    hidden_layer: int = -2             # encode_prompt with clip_skip None: hidden_states[-2]

    @staticmethod
    def sdxl_text_encoder() -> "CLIPTextConfig":
        return CLIPTextConfig()

    @staticmethod
    def sdxl_text_encoder_2() -> "CLIPTextConfig":
        return CLIPTextConfig(hidden_size=1280, intermediate_size=5120, num_hidden_layers=32, num_attention_heads=20, hidden_act="gelu",
                              projection_dim=1280)

    @staticmethod
    def tiny(projection_dim: int = 0, hidden_act: str = "quick_gelu") -> "CLIPTextConfig":
        return CLIPTextConfig(vocab_size=1000, hidden_size=128, intermediate_size=256, num_hidden_layers=3, num_attention_heads=2,
                              hidden_act=hidden_act, projection_dim=projection_dim)

    @staticmethod
    def from_hf_json(path: str, with_projection: bool) -> "CLIPTextConfig":
        import json
        with open(path) as f:
            c = json.load(f)
        return CLIPTextConfig(vocab_size=c["vocab_size"], hidden_size=c["hidden_size"], intermediate_size=c["intermediate_size"],
                              num_hidden_layers=c["num_hidden_layers"], num_attention_heads=c["num_attention_heads"],
                              max_position_embeddings=c["max_position_embeddings"], hidden_act=c["hidden_act"],
                              projection_dim=c.get("projection_dim", 0) if with_projection else 0, eos_token_id=c.get("eos_token_id", 2),
                              layer_norm_eps=c.get("layer_norm_eps", 1e-5))


def pack_clip(cfg: CLIPTextConfig, P: Dict[str, torch.Tensor]) -> List[Tuple[str, torch.Tensor]]:
    """transformers state dict (``text_model.*``, ``text_projection.weight``) -> packed tensors of csrc/clip_text.cpp: matrices bf16, vectors
    fp32, q / k / v fused."""
    bf, f32 = torch.bfloat16, torch.float32
    if not any(k.startswith("text_model.") for k in P):       # transformers >= 5 names a bare CLIPTextModel's tensors without the prefix
        P = {(k if k.startswith("text_projection.") else "text_model." + k): v for k, v in P.items()}   # that every saved checkpoint carries
    out: List[Tuple[str, torch.Tensor]] = []
    done = set()
    for l in range(cfg.num_hidden_layers):
        a = f"text_model.encoder.layers.{l}.self_attn"
        out.append((f"{a}.qkv_proj.weight", torch.cat([P[f"{a}.{n}.weight"] for n in ("q_proj", "k_proj", "v_proj")], dim=0).to(bf).contiguous()))
        out.append((f"{a}.qkv_proj.bias", torch.cat([P[f"{a}.{n}.bias"] for n in ("q_proj", "k_proj", "v_proj")], dim=0).to(f32).contiguous()))
        done.update(f"{a}.{n}.{t}" for n in ("q_proj", "k_proj", "v_proj") for t in ("weight", "bias"))
    for name, t in P.items():
        if name in done or name.endswith("position_ids"):
            continue
        if name == "text_projection.weight" and cfg.projection_dim <= 0:
            continue
        out.append((name, (t.to(bf) if t.ndim == 2 else t.to(f32)).contiguous()))
    return out


class MxCLIPTextEncoder:
    """``encode(ids) -> (hidden_states[hidden_layer] [B, 77, H] bf16, pooled text_embeds [B, projection_dim] fp32 or None)``"""

    def __init__(self, cfg: CLIPTextConfig, params: Dict[str, torch.Tensor], device="cuda:0"):
        self.cfg = cfg
        self.device = torch.device(device)
        self._lib = _lib.load()
        cc = _lib.CLIPConfigC()
        cc.vocab_size, cc.hidden_size, cc.intermediate_size = cfg.vocab_size, cfg.hidden_size, cfg.intermediate_size
        cc.num_hidden_layers, cc.num_attention_heads, cc.max_position_embeddings = cfg.num_hidden_layers, cfg.num_attention_heads, cfg.max_position_embeddings
        cc.hidden_act = {"quick_gelu": 0, "gelu": 1}[cfg.hidden_act]
        cc.projection_dim, cc.eos_token_id, cc.hidden_layer, cc.layer_norm_eps = cfg.projection_dim, cfg.eos_token_id, cfg.hidden_layer, cfg.layer_norm_eps
        self._handle = self._lib.mx_clip_create(C.byref(cc))
        if not self._handle:
            raise _lib.MxError("mx_clip_create: " + self._lib.mx_last_error().decode())
        self.weights = PackedWeights(pack_clip(cfg, params), self.device)
        _lib.check(self._lib.mx_clip_set_weights(self._handle, self.weights.blob.data_ptr(), self.weights.blob.numel(), self.weights.table,
                                                 len(self.weights.names)), "mx_clip_set_weights")
        self._ws = None

    def __del__(self):
        h = getattr(self, "_handle", None)
        if h:
            self._lib.mx_clip_destroy(h)
            self._handle = None

    def validate(self, batch: int) -> None:
        _lib.check(self._lib.mx_clip_validate(self._handle, batch), "mx_clip_validate")

    @torch.inference_mode()
    def encode(self, ids: torch.Tensor) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
        ids = ids.to(device=self.device, dtype=torch.int32).contiguous()
        b, l = ids.shape
        if l != self.cfg.max_position_embeddings:
            raise ValueError(f"ids must be padded to {self.cfg.max_position_embeddings} tokens")
        need = self._lib.mx_clip_workspace_bytes(self._handle, b)
        if need == 0:
            raise _lib.MxError("mx_clip_workspace_bytes: " + self._lib.mx_last_error().decode())
        if self._ws is None or self._ws.numel() < need:
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        hidden = torch.empty((b, l, self.cfg.hidden_size), dtype=torch.bfloat16, device=self.device)
        pooled = torch.empty((b, self.cfg.projection_dim), dtype=torch.float32, device=self.device) if self.cfg.projection_dim > 0 else None
        _lib.check(self._lib.mx_clip_encode(self._handle, _lib.current_stream(), ids.data_ptr(), hidden.data_ptr(),
                                            pooled.data_ptr() if pooled is not None else None, b, self._ws.data_ptr(), self._ws.numel()), "mx_clip_encode")
        return hidden, pooled


def encode_prompt_sdxl(enc1: MxCLIPTextEncoder, enc2: MxCLIPTextEncoder, ids1: torch.Tensor, ids2: torch.Tensor):
    """the tensor part of diffusers' SDXL encode_prompt: prompt_embeds = cat(hidden_states[-2] of both encoders) [B, 77, 2048],
    pooled_prompt_embeds = text_embeds of the second [B, 1280]"""
    h1, _ = enc1.encode(ids1)
    h2, pooled = enc2.encode(ids2)
    return torch.cat([h1, h2], dim=-1), pooled
