"""T5 v1.1 encoder behind the C ABI (``mx_t5_encode``): SD3's ``text_encoder_3`` as diffusers' ``encode_prompt`` runs it for ``prepare_inference``
of the SD3 pipeline (256 token ids per prompt, no attention mask, last hidden state).  SURVEY.md section 8f rank 2.  The tokenizer stays on the host."""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass
from typing import Dict, List, Sequence, Tuple

import torch

from . import lib as _lib
from .weights import PackedWeights, _geglu_interleave

LOG2E = 1.4426950408889634


@dataclass(frozen=True)
class T5Config:
    vocab_size: int = 32128
    d_model: int = 4096
    d_ff: int = 10240
    num_layers: int = 24
    num_heads: int = 64                # d_kv = 64
    relative_attention_num_buckets: int = 32
    relative_attention_max_distance: int = 128
    layer_norm_epsilon: float = 1e-6

    @staticmethod
    def xxl() -> "T5Config":
        """google/t5-v1_1-xxl encoder: SD3 / SD3.5 text_encoder_3"""
        return T5Config()

    @staticmethod
    def tiny() -> "T5Config":
        return T5Config(vocab_size=500, d_model=128, d_ff=256, num_layers=3, num_heads=2)

    @staticmethod
    def from_hf_json(path: str) -> "T5Config":
        import json
        with open(path) as f:
            c = json.load(f)
        assert c.get("d_kv", 64) == 64 and c.get("feed_forward_proj", "gated-gelu") == "gated-gelu", "T5 v1.1 with 64-wide heads only"
        return T5Config(vocab_size=c["vocab_size"], d_model=c["d_model"], d_ff=c["d_ff"], num_layers=c["num_layers"], num_heads=c["num_heads"],
                        relative_attention_num_buckets=c.get("relative_attention_num_buckets", 32),
                        relative_attention_max_distance=c.get("relative_attention_max_distance", 128), layer_norm_epsilon=c.get("layer_norm_epsilon", 1e-6))


def relative_position_bucket(rel: torch.Tensor, num_buckets: int, max_distance: int) -> torch.Tensor:
    """the published T5 rule, bidirectional form (encoder): half of the buckets per sign; within a sign the first half are exact offsets, the
    rest log-spaced up to max_distance.  ``rel`` = key position - query position."""
    nb = num_buckets // 2
    out = (rel > 0).long() * nb
    a = rel.abs()
    max_exact = nb // 2
    large = max_exact + (torch.log(a.float().clamp(min=1) / max_exact) / math.log(max_distance / max_exact) * (nb - max_exact)).long()
    large = torch.minimum(large, torch.full_like(large, nb - 1))
    return out + torch.where(a < max_exact, a, large)


def position_bias(cfg: T5Config, rel_emb: torch.Tensor, L: int) -> torch.Tensor:
    """[heads, L, ceil64(L)] fp32 = relative_attention_bias[bucket(j - i)][head] * log2(e) (the attention kernel works in the log2 domain);
    the pad columns are zero (the kernel masks keys >= L)."""
    pos = torch.arange(L)
    bucket = relative_position_bucket(pos[None, :] - pos[:, None], cfg.relative_attention_num_buckets, cfg.relative_attention_max_distance)
    b = rel_emb.to(torch.float32)[bucket]                       # [L, L, heads]
    out = torch.zeros(cfg.num_heads, L, (L + 63) // 64 * 64, dtype=torch.float32)
    out[:, :, :L] = b.permute(2, 0, 1) * LOG2E
    return out.contiguous()


def pack_t5(cfg: T5Config, P: Dict[str, torch.Tensor], seq_lens: Sequence[int] = (256,)) -> List[Tuple[str, torch.Tensor]]:
    """transformers T5EncoderModel state dict -> packed tensors of csrc/t5_text.cpp (include/mxdenoise.h lists the renamed ones)."""
    bf, f32 = torch.bfloat16, torch.float32
    out: List[Tuple[str, torch.Tensor]] = []
    emb = P["encoder.embed_tokens.weight"] if "encoder.embed_tokens.weight" in P else P["shared.weight"]
    out.append(("encoder.embed_tokens.weight", emb.to(bf).contiguous()))
    for l in range(cfg.num_layers):
        a = f"encoder.block.{l}.layer.0.SelfAttention"
        out.append((f"{a}.qkv.weight", torch.cat([P[f"{a}.{n}.weight"] for n in ("q", "k", "v")], dim=0).to(bf).contiguous()))
        out.append((f"{a}.o.weight", P[f"{a}.o.weight"].to(bf).contiguous()))
        out.append((f"encoder.block.{l}.layer.0.layer_norm.weight", P[f"encoder.block.{l}.layer.0.layer_norm.weight"].to(f32).contiguous()))
        d = f"encoder.block.{l}.layer.1.DenseReluDense"
        # GEGLU epilogue: out = hidden * gelu(gate) with rows [hidden ; gate] interleaved -- hidden = wi_1 (linear), gate = wi_0 (gelu_new)
        out.append((f"{d}.wi.weight", _geglu_interleave(torch.cat([P[f"{d}.wi_1.weight"], P[f"{d}.wi_0.weight"]], dim=0)).to(bf).contiguous()))
        out.append((f"{d}.wo.weight", P[f"{d}.wo.weight"].to(bf).contiguous()))
        out.append((f"encoder.block.{l}.layer.1.layer_norm.weight", P[f"encoder.block.{l}.layer.1.layer_norm.weight"].to(f32).contiguous()))
    out.append(("encoder.final_layer_norm.weight", P["encoder.final_layer_norm.weight"].to(f32).contiguous()))
    rel = P["encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight"]
    for L in seq_lens:
        out.append((f"encoder.position_bias.{L}", position_bias(cfg, rel, L)))
    return out


class MxT5Encoder:
    """``encode(ids [B, L]) -> last_hidden_state [B, L, d_model] bf16``; L must be one of ``seq_lens`` given at construction (SD3: 256)."""

    def __init__(self, cfg: T5Config, params: Dict[str, torch.Tensor], device="cuda:0", seq_lens: Sequence[int] = (256,)):
        self.cfg = cfg
        self.device = torch.device(device)
        self.seq_lens = tuple(seq_lens)
        self._lib = _lib.load()
        cc = _lib.T5ConfigC()
        cc.vocab_size, cc.d_model, cc.d_ff, cc.num_layers, cc.num_heads, cc.layer_norm_epsilon = (cfg.vocab_size, cfg.d_model, cfg.d_ff, cfg.num_layers,
                                                                                                 cfg.num_heads, cfg.layer_norm_epsilon)
        self._handle = self._lib.mx_t5_create(C.byref(cc))
        if not self._handle:
            raise _lib.MxError("mx_t5_create: " + self._lib.mx_last_error().decode())
        self.weights = PackedWeights(pack_t5(cfg, params, self.seq_lens), self.device)
        _lib.check(self._lib.mx_t5_set_weights(self._handle, self.weights.blob.data_ptr(), self.weights.blob.numel(), self.weights.table,
                                               len(self.weights.names)), "mx_t5_set_weights")
        self._ws = None

    def __del__(self):
        h = getattr(self, "_handle", None)
        if h:
            self._lib.mx_t5_destroy(h)
            self._handle = None

    def validate(self, batch: int, L: int) -> None:
        _lib.check(self._lib.mx_t5_validate(self._handle, batch, L), "mx_t5_validate")

    @torch.inference_mode()
    def encode(self, ids: torch.Tensor) -> torch.Tensor:
        ids = ids.to(device=self.device, dtype=torch.int32).contiguous()
        b, l = ids.shape
        if l not in self.seq_lens:
            raise ValueError(f"sequence length {l} was not prepared (seq_lens = {self.seq_lens})")
        need = self._lib.mx_t5_workspace_bytes(self._handle, b, l)
        if need == 0:
            raise _lib.MxError("mx_t5_workspace_bytes: " + self._lib.mx_last_error().decode())
        if self._ws is None or self._ws.numel() < need:
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        out = torch.empty((b, l, self.cfg.d_model), dtype=torch.bfloat16, device=self.device)
        _lib.check(self._lib.mx_t5_encode(self._handle, _lib.current_stream(), ids.data_ptr(), out.data_ptr(), b, l, self._ws.data_ptr(), self._ws.numel()),
                   "mx_t5_encode")
        return out


def encode_prompt_sd3(enc_l, enc_g, enc_t5: MxT5Encoder, ids_l: torch.Tensor, ids_g: torch.Tensor, ids_t5: torch.Tensor):
    """the tensor part of diffusers' SD3 encode_prompt: the two CLIP encoders' hidden_states[-2] side by side [n, 77, 768 + 1280], zero-padded
    to the T5 width and followed by the T5 states along the token axis -> prompt_embeds [n, 77 + 256, 4096]; pooled_prompt_embeds = the two
    projected CLIP embeddings side by side [n, 2048].  ``enc_l`` / ``enc_g``: sduss_amd.clip.MxCLIPTextEncoder, both with projection."""
    h_l, p_l = enc_l.encode(ids_l)
    h_g, p_g = enc_g.encode(ids_g)
    t5 = enc_t5.encode(ids_t5)
    clip = torch.cat([h_l, h_g], dim=-1)
    clip = torch.nn.functional.pad(clip, (0, t5.shape[-1] - clip.shape[-1]))
    return torch.cat([clip, t5], dim=-2), torch.cat([p_l, p_g], dim=-1)
