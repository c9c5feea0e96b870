"""CPU fp32 restatement of the SD3.5 MMDiT forward that sits in sduss's ``transformer`` slot.

TEST INFRASTRUCTURE (see oracle/__init__.py): checker only.  PARITY UNPINNED (no reference golden vectors; diffusers and
the weights are absent here).

Restated from (paths relative to /root/reference):
* ``PatchSD3Transformer2DModel.forward``  -- sduss/model_executor/modules/SD3Transformer.py:60-262
  (time_text_embed :81, pos_embed per resolution :82-83, context_embedder :115, 24 blocks :116-236, norm_out + proj_out
  :238-239, unpatchify ``nhwpqc->nchpwq`` :250-259).  With the block cache off, the sliced branch only re-chunks the token
  axis (modules/utils.py:86-122) and regroups it before attention (attention.py:300-372), so it computes exactly what the
  unsliced branch computes; there is no halo or statistic approximation on this model.
* ``PatchJointTransformerBlock.forward``  -- modules/transformer.py:299-388 (AdaLN-Zero / SD35AdaLayerNormZeroX on the
  image stream, AdaLN-Zero or AdaLN-continuous on the context stream, gated residuals, GELU-tanh FF).
* ``PatchSD3Attention.forward``           -- modules/attention.py:241-424 (q / fused kv on image tokens, add_{q,k,v}_proj on
  context tokens, RMSNorm(64) on q and k of both streams, concat IMAGE FIRST then text, SDPA, split, to_out / to_add_out;
  ``attn2`` = image-only self-attention).
The third-party modules (diffusers==0.32.1: CombinedTimestepTextProjEmbeddings, PatchEmbed with the persistent
``pos_embed`` buffer, AdaLayerNormZero, SD35AdaLayerNormZeroX, AdaLayerNormContinuous, RMSNorm, FeedForward(gelu-approximate))
are restated from their published definitions with torch functional primitives.  Parameter names are the HF state-dict keys
of ``transformer/diffusion_pytorch_model.safetensors``.
"""
from __future__ import annotations

import json
import math
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

from .sdxl_unet_ref import timestep_embedding


@dataclass
class MMDiTConfig:
    sample_size: int = 128
    patch_size: int = 2
    in_channels: int = 16
    out_channels: int = 16
    num_layers: int = 24
    attention_head_dim: int = 64
    num_attention_heads: int = 24
    joint_attention_dim: int = 4096
    caption_projection_dim: int = 1536
    pooled_projection_dim: int = 2048
    pos_embed_max_size: int = 384
    dual_attention_layers: Tuple[int, ...] = tuple(range(13))
    norm_eps: float = 1e-6

    @property
    def dim(self) -> int:
        return self.attention_head_dim * self.num_attention_heads

    @staticmethod
    def sd35_medium() -> "MMDiTConfig":
        return MMDiTConfig()

    @staticmethod
    def tiny() -> "MMDiTConfig":
        return MMDiTConfig(sample_size=16, num_layers=4, num_attention_heads=2, joint_attention_dim=128,
                           caption_projection_dim=128, pooled_projection_dim=64, pos_embed_max_size=24,
                           dual_attention_layers=(0, 1))

    @staticmethod
    def from_hf_json(path: str) -> "MMDiTConfig":
        with open(path) as f:
            c = json.load(f)
        return MMDiTConfig(sample_size=c["sample_size"], patch_size=c["patch_size"], in_channels=c["in_channels"],
                           out_channels=c.get("out_channels", c["in_channels"]), num_layers=c["num_layers"],
                           attention_head_dim=c["attention_head_dim"], num_attention_heads=c["num_attention_heads"],
                           joint_attention_dim=c["joint_attention_dim"], caption_projection_dim=c["caption_projection_dim"],
                           pooled_projection_dim=c["pooled_projection_dim"], pos_embed_max_size=c["pos_embed_max_size"],
                           dual_attention_layers=tuple(c.get("dual_attention_layers", ())))


def _lin(s, name, n, k, bias=True):
    s[f"{name}.weight"] = (n, k)
    if bias:
        s[f"{name}.bias"] = (n,)


def param_shapes(cfg: MMDiTConfig) -> Dict[str, Tuple[int, ...]]:
    d = cfg.dim
    s: Dict[str, Tuple[int, ...]] = {}
    s["pos_embed.pos_embed"] = (1, cfg.pos_embed_max_size ** 2, d)
    s["pos_embed.proj.weight"] = (d, cfg.in_channels, cfg.patch_size, cfg.patch_size)
    s["pos_embed.proj.bias"] = (d,)
    _lin(s, "time_text_embed.timestep_embedder.linear_1", d, 256)
    _lin(s, "time_text_embed.timestep_embedder.linear_2", d, d)
    _lin(s, "time_text_embed.text_embedder.linear_1", d, cfg.pooled_projection_dim)
    _lin(s, "time_text_embed.text_embedder.linear_2", d, d)
    _lin(s, "context_embedder", d, cfg.joint_attention_dim)
    for i in range(cfg.num_layers):
        b = f"transformer_blocks.{i}"
        last = i == cfg.num_layers - 1
        dual = i in cfg.dual_attention_layers
        _lin(s, f"{b}.norm1.linear", (9 if dual else 6) * d, d)
        _lin(s, f"{b}.norm1_context.linear", (2 if last else 6) * d, d)
        for nm in ("to_q", "to_k", "to_v", "add_q_proj", "add_k_proj", "add_v_proj"):
            _lin(s, f"{b}.attn.{nm}", d, d)
        for nm in ("norm_q", "norm_k", "norm_added_q", "norm_added_k"):
            s[f"{b}.attn.{nm}.weight"] = (cfg.attention_head_dim,)
        _lin(s, f"{b}.attn.to_out.0", d, d)
        if not last:
            _lin(s, f"{b}.attn.to_add_out", d, d)
        if dual:
            for nm in ("to_q", "to_k", "to_v"):
                _lin(s, f"{b}.attn2.{nm}", d, d)
            for nm in ("norm_q", "norm_k"):
                s[f"{b}.attn2.{nm}.weight"] = (cfg.attention_head_dim,)
            _lin(s, f"{b}.attn2.to_out.0", d, d)
        _lin(s, f"{b}.ff.net.0.proj", 4 * d, d)
        _lin(s, f"{b}.ff.net.2", d, 4 * d)
        if not last:
            _lin(s, f"{b}.ff_context.net.0.proj", 4 * d, d)
            _lin(s, f"{b}.ff_context.net.2", d, 4 * d)
    _lin(s, "norm_out.linear", 2 * d, d)
    _lin(s, "proj_out", cfg.patch_size ** 2 * cfg.out_channels, d)
    return s


def init_params(cfg: MMDiTConfig, seed: int = 10086) -> Dict[str, torch.Tensor]:
    """Seeded synthetic weights (bf16-representable fp32), same convention as sdxl_unet_ref.init_params.  The AdaLN
    projections get a smaller gain so the modulated activations stay O(1) through the stack."""
    g = torch.Generator().manual_seed(seed)
    pool = torch.randn(1 << 22, generator=g)
    off = 0
    out: Dict[str, torch.Tensor] = {}
    for name, shape in sorted(param_shapes(cfg).items()):
        n = 1
        for x in shape:
            n *= x
        if n <= pool.numel():
            off = off if off + n <= pool.numel() else 0
            t = pool[off:off + n]
            off += n
        else:
            t = pool.repeat((n + pool.numel() - 1) // pool.numel())[:n]
        t = t.reshape(shape).clone()
        if name == "pos_embed.pos_embed":
            t = 0.5 * t
        elif name.endswith("norm_q.weight") or name.endswith("norm_k.weight") or name.endswith("norm_added_q.weight") \
                or name.endswith("norm_added_k.weight"):
            t = 1.0 + 0.1 * t
        elif name.endswith(".weight"):
            gain = 0.3 if (".norm1" in name or "norm_out" in name) else 1.0
            t = t * gain * (n // shape[0]) ** -0.5
        else:
            t = 0.05 * t
        out[name] = t.to(torch.bfloat16).to(torch.float32)
    return out


# --------------------------------------------------------------------------------------
def _ln(x, eps):
    return F.layer_norm(x, (x.shape[-1],), None, None, eps)


def _rms(x, w, eps):
    """diffusers RMSNorm(64, eps, elementwise_affine=True) on the head dim."""
    return x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps) * w


def _heads(x, h):
    b, l, c = x.shape
    return x.reshape(b, l, h, c // h).transpose(1, 2)


_LOWP = {"sdpa": False}


def _sdpa(q, k, v):
    if _LOWP["sdpa"]:      # the low-precision comparator only (mmdit_forward compute_dtype != fp32): the fused kernel the reference itself calls (attention.py:350)
        return F.scaled_dot_product_attention(q, k, v)
    s = torch.matmul(q, k.transpose(-1, -2)) * (q.shape[-1] ** -0.5)
    return torch.matmul(torch.softmax(s, dim=-1), v)


def joint_attention(P, p, x, ctx, cfg: MMDiTConfig, context_pre_only: bool):
    """attention.py:241-424: returns (image_out, context_out or None)."""
    h, eps = cfg.num_attention_heads, cfg.norm_eps
    q = _rms(_heads(F.linear(x, P[f"{p}.to_q.weight"], P[f"{p}.to_q.bias"]), h), P[f"{p}.norm_q.weight"], eps)
    k = _rms(_heads(F.linear(x, P[f"{p}.to_k.weight"], P[f"{p}.to_k.bias"]), h), P[f"{p}.norm_k.weight"], eps)
    v = _heads(F.linear(x, P[f"{p}.to_v.weight"], P[f"{p}.to_v.bias"]), h)
    li = x.shape[1]
    if ctx is not None:
        cq = _rms(_heads(F.linear(ctx, P[f"{p}.add_q_proj.weight"], P[f"{p}.add_q_proj.bias"]), h), P[f"{p}.norm_added_q.weight"], eps)
        ck = _rms(_heads(F.linear(ctx, P[f"{p}.add_k_proj.weight"], P[f"{p}.add_k_proj.bias"]), h), P[f"{p}.norm_added_k.weight"], eps)
        cv = _heads(F.linear(ctx, P[f"{p}.add_v_proj.weight"], P[f"{p}.add_v_proj.bias"]), h)
        q = torch.cat([q, cq], dim=2); k = torch.cat([k, ck], dim=2); v = torch.cat([v, cv], dim=2)   # image first (:347-348)
    o = _sdpa(q, k, v).transpose(1, 2).reshape(x.shape[0], -1, cfg.dim)
    xo = F.linear(o[:, :li], P[f"{p}.to_out.0.weight"], P[f"{p}.to_out.0.bias"])
    co = None
    if ctx is not None and not context_pre_only:
        co = F.linear(o[:, li:], P[f"{p}.to_add_out.weight"], P[f"{p}.to_add_out.bias"])
    return xo, co


def _ff(P, p, x):
    hcat = F.gelu(F.linear(x, P[f"{p}.net.0.proj.weight"], P[f"{p}.net.0.proj.bias"]), approximate="tanh")
    return F.linear(hcat, P[f"{p}.net.2.weight"], P[f"{p}.net.2.bias"])


def joint_block(P, b, x, ctx, temb, cfg: MMDiTConfig, last: bool, dual: bool):
    """transformer.py:299-388."""
    eps = cfg.norm_eps
    semb = F.silu(temb)
    mod = F.linear(semb, P[f"{b}.norm1.linear.weight"], P[f"{b}.norm1.linear.bias"])
    nx = _ln(x, eps)
    if dual:
        sh, sc, gate, sh_mlp, sc_mlp, g_mlp, sh2, sc2, gate2 = mod.chunk(9, dim=1)
        x2in = nx * (1 + sc2[:, None]) + sh2[:, None]
    else:
        sh, sc, gate, sh_mlp, sc_mlp, g_mlp = mod.chunk(6, dim=1)
    xin = nx * (1 + sc[:, None]) + sh[:, None]
    cmod = F.linear(semb, P[f"{b}.norm1_context.linear.weight"], P[f"{b}.norm1_context.linear.bias"])
    if last:
        c_sc, c_sh = cmod.chunk(2, dim=1)                        # AdaLayerNormContinuous: scale first
        cin = _ln(ctx, eps) * (1 + c_sc[:, None]) + c_sh[:, None]
    else:
        c_sh, c_sc, c_gate, c_sh_mlp, c_sc_mlp, c_g_mlp = cmod.chunk(6, dim=1)
        cin = _ln(ctx, eps) * (1 + c_sc[:, None]) + c_sh[:, None]
    ao, co = joint_attention(P, f"{b}.attn", xin, cin, cfg, last)
    x = x + gate[:, None] * ao
    if dual:
        ao2, _ = joint_attention(P, f"{b}.attn2", x2in, None, cfg, True)
        x = x + gate2[:, None] * ao2
    nh = _ln(x, eps) * (1 + sc_mlp[:, None]) + sh_mlp[:, None]
    x = x + g_mlp[:, None] * _ff(P, f"{b}.ff", nh)
    if last:
        return x, None
    ctx = ctx + c_gate[:, None] * co
    nc = _ln(ctx, eps) * (1 + c_sc_mlp[:, None]) + c_sh_mlp[:, None]
    ctx = ctx + c_g_mlp[:, None] * _ff(P, f"{b}.ff_context", nc)
    return x, ctx


def mmdit_forward(P: Dict[str, torch.Tensor], cfg: MMDiTConfig, latents: torch.Tensor, timestep: torch.Tensor,
                  encoder_hidden_states: torch.Tensor, pooled: torch.Tensor, trace: Optional[dict] = None,
                  compute_dtype: torch.dtype = torch.float32, device=None, sdpa: bool = False) -> torch.Tensor:
    """latents [B, C, H, W]; timestep [B]; encoder_hidden_states [B, Lt, joint_dim]; pooled [B, pooled_dim] -> [B, C, H, W]."""
    # compute_dtype / device / sdpa: the SAME graph evaluated by stock torch ops in bf16 or fp16 on the GPU box -- the yardstick the parity
    # tolerances are calibrated against (tests/test_parity_calibration_gpu.py); the oracle proper is the fp32 default.
    dev = torch.device(device) if device is not None else latents.device
    dt = compute_dtype
    P = {k: v.to(device=dev, dtype=dt) for k, v in P.items()}
    x = latents.to(device=dev, dtype=dt)
    timestep = timestep.to(dev)
    _LOWP["sdpa"] = bool(sdpa)
    b, c, hh, ww = x.shape
    ps = cfg.patch_size
    h, w = hh // ps, ww // ps
    d = cfg.dim
    # time_text_embed (SD3Transformer.py:81)
    t = F.linear(timestep_embedding(timestep, 256).to(dt), P["time_text_embed.timestep_embedder.linear_1.weight"], P["time_text_embed.timestep_embedder.linear_1.bias"])
    t = F.linear(F.silu(t), P["time_text_embed.timestep_embedder.linear_2.weight"], P["time_text_embed.timestep_embedder.linear_2.bias"])
    pp = F.linear(pooled.to(device=dev, dtype=dt), P["time_text_embed.text_embedder.linear_1.weight"], P["time_text_embed.text_embedder.linear_1.bias"])
    pp = F.linear(F.silu(pp), P["time_text_embed.text_embedder.linear_2.weight"], P["time_text_embed.text_embedder.linear_2.bias"])
    temb = t + pp
    # pos_embed (PatchEmbed): conv p x p stride p, flatten, + centre-cropped table (:82-83)
    x = F.conv2d(x, P["pos_embed.proj.weight"], P["pos_embed.proj.bias"], stride=ps).flatten(2).transpose(1, 2)
    m = cfg.pos_embed_max_size
    top, left = (m - h) // 2, (m - w) // 2
    pe = P["pos_embed.pos_embed"].reshape(1, m, m, d)[:, top:top + h, left:left + w].reshape(1, h * w, d)
    x = x + pe
    ctx = F.linear(encoder_hidden_states.to(device=dev, dtype=dt), P["context_embedder.weight"], P["context_embedder.bias"])
    if trace is not None:
        trace["embed"] = x; trace["context_embed"] = ctx; trace["temb"] = temb
    for i in range(cfg.num_layers):
        x, ctx = joint_block(P, f"transformer_blocks.{i}", x, ctx, temb, cfg, i == cfg.num_layers - 1, i in cfg.dual_attention_layers)
        if trace is not None:
            trace[f"transformer_blocks.{i}"] = x
            if ctx is not None:
                trace[f"transformer_blocks.{i}.context"] = ctx
    sc, sh = F.linear(F.silu(temb), P["norm_out.linear.weight"], P["norm_out.linear.bias"]).chunk(2, dim=1)
    x = _ln(x, cfg.norm_eps) * (1 + sc[:, None]) + sh[:, None]
    x = F.linear(x, P["proj_out.weight"], P["proj_out.bias"])
    x = x.reshape(b, h, w, ps, ps, cfg.out_channels)
    x = torch.einsum("nhwpqc->nchpwq", x).reshape(b, cfg.out_channels, h * ps, w * ps)    # :250-259
    _LOWP["sdpa"] = False
    return x


def make_inputs(cfg: MMDiTConfig, batch: int, latent_hw: int, seed: int = 10086, ctx_len: int = 333):
    g = torch.Generator().manual_seed(seed + 3)
    lat = torch.randn(batch, cfg.in_channels, latent_hw, latent_hw, generator=g)
    ehs = torch.randn(batch, ctx_len, cfg.joint_attention_dim, generator=g)
    pooled = torch.randn(batch, cfg.pooled_projection_dim, generator=g)
    t = torch.full((batch,), 701.0)
    return lat, t, ehs, pooled


# --------------------------------------------------------------------------------------
# token re-chunk either side of the sliced MMDiT forward  (modules/utils.py:86-136)
# Pinned against the reference itself: tests/golden/ref_utils_sd3.npz (tests/golden/make_ref_fixtures.py).
# --------------------------------------------------------------------------------------
def split_sample_sd3(samples: Dict[str, torch.Tensor], patch_size: int, input_indices: Dict[str, list]):
    """samples: {resolution: tokens [n, L, D]} -> (indices, encoder_indices, latent_offset, resolution_offset, chunks).
    Every latent is cut into (resolution // patch_size)^2 equal TOKEN RANGES (not 2-D patches: utils.py:110) and all chunks
    are stacked -- so every chunk of the call must have the same length (a torch.stack of unequal chunks raises)."""
    latent_offset, resolution_offset = [0], [0]
    chunks, indices, encoder_indices = [], [], []
    for resolution, res_sample in samples.items():
        if res_sample is None or res_sample.shape[0] == 0:
            continue
        pn = int(resolution) // patch_size
        for i, sample in enumerate(res_sample):
            latent_offset.append(latent_offset[-1] + pn ** 2)
            rid = input_indices[str(int(resolution))][i]
            encoder_indices.append(rid)
            indices.extend(f"{rid}-{h}" for h in range(pn * pn))
            chunks.extend(sample.chunk(pn * pn, dim=0))
        resolution_offset.append(len(latent_offset) - 1)
    return indices, encoder_indices, latent_offset, resolution_offset, torch.stack(chunks)


def concat_sample_tokens(patch_size: int, new_sample: torch.Tensor, latent_offset) -> Dict[str, torch.Tensor]:
    """inverse of split_sample_sd3 on the token axis (utils.py:124-136): chunks of one latent are viewed back as one
    [1, L, D] sequence and grouped under str(sqrt(chunks) * patch_size)."""
    import math
    samples: Dict[str, list] = {}
    for i in range(len(latent_offset) - 1):
        n = latent_offset[i + 1] - latent_offset[i]
        size = int(math.sqrt(n)) * patch_size
        samples.setdefault(str(size), []).append(new_sample[latent_offset[i]:latent_offset[i + 1]].reshape(1, -1, new_sample.shape[-1]))
    return {k: torch.cat(v, dim=0) for k, v in samples.items()}
