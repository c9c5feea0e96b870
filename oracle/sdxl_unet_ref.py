"""CPU fp32 restatement of the SDXL UNet forward that sits in sduss's model slot.

TEST INFRASTRUCTURE (see oracle/__init__.py): checker only, never the product path.
PARITY UNPINNED (no reference golden vectors exist; diffusers + weights absent here).

What is restated, and from where (paths relative to /root/reference):

* ``PatchUNet.forward`` unsliced branch -- sduss/model_executor/modules/unet.py:205-530
  (calls into diffusers==0.32.1 ``UNet2DConditionModel`` sub-modules: ``get_time_embed``,
  ``time_embedding``, ``get_aug_embed`` at unet.py:314-341; ``conv_in`` :344; down/mid/up
  loops :371-503 with the skip-tuple handling of :458-462; ``conv_norm_out``/``conv_out``
  :508-517).
* ``PatchResnetBlock2D.forward`` -- modules/resnet.py:390-460.
* ``PatchTransformer2DModel.forward`` (use_linear_projection=True) -- modules/transformer.py:32-128.
* ``PatchBasicTransformerBlock.forward`` (norm_type layer_norm) -- modules/transformer.py:167-290.
* ``PatchSelfAttention`` / ``PatchCrossAttention`` (fused to_kv, softmax(QK^T/sqrt(d))V,
  to_out[0]) -- modules/attention.py:23-50, 59-110, 121-232.
* ``PatchUpsample2D`` / ``PatchDownsample2D`` -- modules/resnet.py:280-378.

The third-party arithmetic (diffusers 0.32.1 / torch 2.2.2) is restated from its published
definition with the torch functional primitives it bottoms out in: F.conv2d, F.group_norm,
F.layer_norm, F.linear, F.silu, F.gelu (exact/erf, GEGLU), F.interpolate(nearest), softmax.

Parameter names are the HF diffusers state-dict keys, so a real
``unet/diffusion_pytorch_model.safetensors`` loads without renaming.
"""
from __future__ import annotations

import json
import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------------------
# configuration
# --------------------------------------------------------------------------------------
@dataclass
class UNetConfig:
    """Subset of the HF ``unet/config.json`` keys the forward depends on."""

    in_channels: int = 4
    out_channels: int = 4
    block_out_channels: Tuple[int, ...] = (320, 640, 1280)
    layers_per_block: int = 2
    # True where the down block is CrossAttnDownBlock2D (mirrored for the up blocks)
    down_has_attn: Tuple[bool, ...] = (False, True, True)
    transformer_layers_per_block: Tuple[int, ...] = (1, 2, 10)
    # HF calls this "attention_head_dim" but for SDXL it is the number of heads per level
    num_heads: Tuple[int, ...] = (5, 10, 20)
    cross_attention_dim: int = 2048
    addition_time_embed_dim: int = 256
    projection_class_embeddings_input_dim: int = 2816
    norm_num_groups: int = 32
    norm_eps: float = 1e-5
    transformer_norm_eps: float = 1e-6  # diffusers Transformer2DModel GroupNorm eps
    layer_norm_eps: float = 1e-5

    @property
    def time_embed_dim(self) -> int:
        return self.block_out_channels[0] * 4

    @property
    def text_embed_dim(self) -> int:
        return self.projection_class_embeddings_input_dim - 6 * self.addition_time_embed_dim

    @staticmethod
    def sdxl_base() -> "UNetConfig":
        return UNetConfig()

    @staticmethod
    def tiny() -> "UNetConfig":
        """Reduced-width config with the SDXL topology (tests; every GEMM K stays a multiple of 64)."""
        return UNetConfig(
            block_out_channels=(64, 128, 256),
            transformer_layers_per_block=(1, 1, 2),
            num_heads=(1, 2, 4),
            cross_attention_dim=128,
            addition_time_embed_dim=32,
            projection_class_embeddings_input_dim=64 + 6 * 32,
        )

    @staticmethod
    def from_hf_json(path: str) -> "UNetConfig":
        with open(path) as f:
            c = json.load(f)
        heads = c["attention_head_dim"]
        if isinstance(heads, int):
            heads = [heads] * len(c["block_out_channels"])
        tl = c.get("transformer_layers_per_block", 1)
        if isinstance(tl, int):
            tl = [tl] * len(c["block_out_channels"])
        return UNetConfig(
            in_channels=c["in_channels"],
            out_channels=c["out_channels"],
            block_out_channels=tuple(c["block_out_channels"]),
            layers_per_block=c["layers_per_block"],
            down_has_attn=tuple(t.startswith("CrossAttn") for t in c["down_block_types"]),
            transformer_layers_per_block=tuple(tl),
            num_heads=tuple(heads),
            cross_attention_dim=c["cross_attention_dim"],
            addition_time_embed_dim=c["addition_time_embed_dim"],
            projection_class_embeddings_input_dim=c["projection_class_embeddings_input_dim"],
            norm_num_groups=c["norm_num_groups"],
            norm_eps=c["norm_eps"],
        )


# --------------------------------------------------------------------------------------
# parameter inventory (HF names) and synthetic init
# --------------------------------------------------------------------------------------
def _resnet_shapes(p: str, cin: int, cout: int, temb: int) -> Dict[str, Tuple[int, ...]]:
    s = {
        f"{p}.norm1.weight": (cin,), f"{p}.norm1.bias": (cin,),
        f"{p}.conv1.weight": (cout, cin, 3, 3), f"{p}.conv1.bias": (cout,),
        f"{p}.time_emb_proj.weight": (cout, temb), f"{p}.time_emb_proj.bias": (cout,),
        f"{p}.norm2.weight": (cout,), f"{p}.norm2.bias": (cout,),
        f"{p}.conv2.weight": (cout, cout, 3, 3), f"{p}.conv2.bias": (cout,),
    }
    if cin != cout:
        s[f"{p}.conv_shortcut.weight"] = (cout, cin, 1, 1)
        s[f"{p}.conv_shortcut.bias"] = (cout,)
    return s


def _transformer_shapes(p: str, dim: int, layers: int, ctx: int) -> Dict[str, Tuple[int, ...]]:
    s = {
        f"{p}.norm.weight": (dim,), f"{p}.norm.bias": (dim,),
        f"{p}.proj_in.weight": (dim, dim), f"{p}.proj_in.bias": (dim,),
        f"{p}.proj_out.weight": (dim, dim), f"{p}.proj_out.bias": (dim,),
    }
    for k in range(layers):
        b = f"{p}.transformer_blocks.{k}"
        for n in ("norm1", "norm2", "norm3"):
            s[f"{b}.{n}.weight"] = (dim,)
            s[f"{b}.{n}.bias"] = (dim,)
        s[f"{b}.attn1.to_q.weight"] = (dim, dim)
        s[f"{b}.attn1.to_k.weight"] = (dim, dim)
        s[f"{b}.attn1.to_v.weight"] = (dim, dim)
        s[f"{b}.attn1.to_out.0.weight"] = (dim, dim)
        s[f"{b}.attn1.to_out.0.bias"] = (dim,)
        s[f"{b}.attn2.to_q.weight"] = (dim, dim)
        s[f"{b}.attn2.to_k.weight"] = (dim, ctx)
        s[f"{b}.attn2.to_v.weight"] = (dim, ctx)
        s[f"{b}.attn2.to_out.0.weight"] = (dim, dim)
        s[f"{b}.attn2.to_out.0.bias"] = (dim,)
        s[f"{b}.ff.net.0.proj.weight"] = (8 * dim, dim)
        s[f"{b}.ff.net.0.proj.bias"] = (8 * dim,)
        s[f"{b}.ff.net.2.weight"] = (dim, 4 * dim)
        s[f"{b}.ff.net.2.bias"] = (dim,)
    return s


def param_shapes(cfg: UNetConfig) -> Dict[str, Tuple[int, ...]]:
    """Ordered {HF state-dict key: shape} for the UNet described by ``cfg``."""
    ch = cfg.block_out_channels
    temb = cfg.time_embed_dim
    s: Dict[str, Tuple[int, ...]] = {}
    s["conv_in.weight"] = (ch[0], cfg.in_channels, 3, 3)
    s["conv_in.bias"] = (ch[0],)
    s["time_embedding.linear_1.weight"] = (temb, ch[0]); s["time_embedding.linear_1.bias"] = (temb,)
    s["time_embedding.linear_2.weight"] = (temb, temb); s["time_embedding.linear_2.bias"] = (temb,)
    s["add_embedding.linear_1.weight"] = (temb, cfg.projection_class_embeddings_input_dim)
    s["add_embedding.linear_1.bias"] = (temb,)
    s["add_embedding.linear_2.weight"] = (temb, temb); s["add_embedding.linear_2.bias"] = (temb,)
    # down
    out = ch[0]
    n = len(ch)
    for i in range(n):
        cin, out = out, ch[i]
        for j in range(cfg.layers_per_block):
            s.update(_resnet_shapes(f"down_blocks.{i}.resnets.{j}", cin if j == 0 else out, out, temb))
            if cfg.down_has_attn[i]:
                s.update(_transformer_shapes(f"down_blocks.{i}.attentions.{j}", out,
                                             cfg.transformer_layers_per_block[i], cfg.cross_attention_dim))
        if i != n - 1:
            s[f"down_blocks.{i}.downsamplers.0.conv.weight"] = (out, out, 3, 3)
            s[f"down_blocks.{i}.downsamplers.0.conv.bias"] = (out,)
    # mid
    s.update(_resnet_shapes("mid_block.resnets.0", ch[-1], ch[-1], temb))
    s.update(_transformer_shapes("mid_block.attentions.0", ch[-1],
                                 cfg.transformer_layers_per_block[-1], cfg.cross_attention_dim))
    s.update(_resnet_shapes("mid_block.resnets.1", ch[-1], ch[-1], temb))
    # up
    rev = list(reversed(ch))
    rev_attn = list(reversed(cfg.down_has_attn))
    rev_layers = list(reversed(cfg.transformer_layers_per_block))
    prev = rev[0]
    for i in range(n):
        out_c = rev[i]
        in_c = rev[min(i + 1, n - 1)]
        for j in range(cfg.layers_per_block + 1):
            skip = in_c if j == cfg.layers_per_block else out_c
            rin = prev if j == 0 else out_c
            s.update(_resnet_shapes(f"up_blocks.{i}.resnets.{j}", rin + skip, out_c, temb))
            if rev_attn[i]:
                s.update(_transformer_shapes(f"up_blocks.{i}.attentions.{j}", out_c, rev_layers[i],
                                             cfg.cross_attention_dim))
        if i != n - 1:
            s[f"up_blocks.{i}.upsamplers.0.conv.weight"] = (out_c, out_c, 3, 3)
            s[f"up_blocks.{i}.upsamplers.0.conv.bias"] = (out_c,)
        prev = out_c
    s["conv_norm_out.weight"] = (ch[0],); s["conv_norm_out.bias"] = (ch[0],)
    s["conv_out.weight"] = (cfg.out_channels, ch[0], 3, 3); s["conv_out.bias"] = (cfg.out_channels,)
    return s


def init_params(cfg: UNetConfig, seed: int = 10086, bf16_round: bool = True) -> Dict[str, torch.Tensor]:
    """Seeded synthetic weights: N(0,1)*fan_in^-1/2 for matrices/kernels, norm gains ~1, biases small.

    Seed 10086 is the reference's default seed (sduss/engine/arg_utils.py:20).  Values are rounded
    to bf16-representable fp32 so the HIP path (bf16 storage) and the oracle (fp32 math) share
    *identical* weights and the comparison isolates activation/accumulation precision.
    """
    g = torch.Generator().manual_seed(seed)
    out: Dict[str, torch.Tensor] = {}
    for name, shape in sorted(param_shapes(cfg).items()):
        if name.endswith(".weight") and len(shape) >= 2:
            fan_in = 1
            for d in shape[1:]:
                fan_in *= d
            t = torch.randn(shape, generator=g) * fan_in ** -0.5
        elif name.endswith(".weight"):  # norm gain
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        else:  # bias
            t = 0.05 * torch.randn(shape, generator=g)
        if bf16_round:
            t = t.to(torch.bfloat16).to(torch.float32)
        out[name] = t
    return out


def fast_params(cfg: UNetConfig, seed: int = 10086) -> Dict[str, torch.Tensor]:
    """Same distribution as ``init_params`` but tiled from one 16M-element random pool, so the 2.57 B-parameter
    SDXL-base inventory is ready in seconds (full-width parity test, bench cpu_baseline)."""
    g = torch.Generator().manual_seed(seed)
    pool = torch.randn(1 << 24, generator=g)
    out: Dict[str, torch.Tensor] = {}
    off = 0
    for name, shape in param_shapes(cfg).items():
        n = 1
        for d in shape:
            n *= d
        if n <= pool.numel():
            off = off if off + n <= pool.numel() else 0
            t = pool[off:off + n]
            off += n
        else:
            t = pool.repeat((n + pool.numel() - 1) // pool.numel())[:n]
        t = t.reshape(shape)
        if name.endswith(".weight") and len(shape) >= 2:
            t = t * (n // shape[0]) ** -0.5
        elif name.endswith(".weight"):
            t = 1.0 + 0.1 * t
        else:
            t = 0.05 * t
        out[name] = t.to(torch.bfloat16).to(torch.float32)
    return out


# --------------------------------------------------------------------------------------
# primitives
# --------------------------------------------------------------------------------------
def timestep_embedding(t: torch.Tensor, dim: int) -> torch.Tensor:
    """diffusers ``Timesteps(dim, flip_sin_to_cos=True, downscale_freq_shift=0)`` -> [cos | sin]."""
    half = dim // 2
    exponent = -math.log(10000.0) * torch.arange(half, dtype=torch.float32, device=t.device) / half
    ang = t.reshape(-1, 1).to(torch.float32) * torch.exp(exponent)[None, :]
    return torch.cat([torch.cos(ang), torch.sin(ang)], dim=-1)


def group_norm_patchavg(x: torch.Tensor, groups: int, w: torch.Tensor, b: torch.Tensor, eps: float,
                        patch: int) -> torch.Tensor:
    """GroupNorm whose statistics are the *average over p x p spatial patches* of per-patch
    mean and biased variance -- the cross-patch merge of ``GetFullMeanAndRstd``
    (kernels/norm_silu_concat.cu:361-386: mean <- avg(patch means); rstd <- rsqrt(avg(patch vars)+eps)).
    With one patch per image this is exactly F.group_norm.
    """
    n, c, h, wd = x.shape
    ph, pw = h // patch, wd // patch
    xg = x.reshape(n, groups, c // groups, ph, patch, pw, patch)
    mean_p = xg.mean(dim=(2, 4, 6))  # [n, g, ph, pw]
    var_p = xg.var(dim=(2, 4, 6), unbiased=False)
    mean = mean_p.mean(dim=(2, 3))
    rstd = torch.rsqrt(var_p.mean(dim=(2, 3)) + eps)
    y = (x.reshape(n, groups, -1) - mean[:, :, None]) * rstd[:, :, None]
    return y.reshape(n, c, h, wd) * w[None, :, None, None] + b[None, :, None, None]


def _gn(x, groups, w, b, eps, gn_patch):
    if gn_patch is None or gn_patch >= x.shape[-1]:
        return F.group_norm(x, groups, w, b, eps)
    return group_norm_patchavg(x, groups, w, b, eps, gn_patch)


_LOWP = {"sdpa": False}


def attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, heads: int) -> torch.Tensor:
    """softmax(Q K^T / sqrt(d)) V, no mask -- what xformers.memory_efficient_attention(q,k,v)
    computes at attention.py:86,214 after ``head_to_batch_dim``."""
    b, lq, c = q.shape
    d = c // heads
    qh = q.reshape(b, lq, heads, d).transpose(1, 2)
    kh = k.reshape(b, -1, heads, d).transpose(1, 2)
    vh = v.reshape(b, -1, heads, d).transpose(1, 2)
    if _LOWP["sdpa"]:      # the low-precision comparator only (unet_forward compute_dtype != fp32): the fused kernel a stock pipeline would call
        o = F.scaled_dot_product_attention(qh, kh, vh)
    else:
        s = torch.matmul(qh, kh.transpose(-1, -2)) * (d ** -0.5)
        o = torch.matmul(torch.softmax(s, dim=-1), vh)
    return o.transpose(1, 2).reshape(b, lq, c)


def conv3x3(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, stride: int = 1,
            corner_patch: Optional[int] = None) -> torch.Tensor:
    """3x3 conv, padding 1.  With ``corner_patch`` = p it reproduces, on the WHOLE image, what the
    reference's sliced path computes from halo'd p x p patches: a diagonal tap that leaves the centre
    pixel's patch through a patch CORNER does not see the diagonal neighbour but the pixel the
    left/right neighbour replicated into that halo corner -- global pixel (y, x+dx) -- or zero when
    x+dx is outside the image (norm_silu_concat.cu:210-221, 228-239; top/bottom writes cover columns
    1..W only, :186-201).  Everything else equals the zero-padded whole-image conv.
    """
    if corner_patch is None or corner_patch >= x.shape[-1]:
        return F.conv2d(x, w, b, stride=stride, padding=1)
    n, c, h, wd = x.shape
    p = corner_patch
    xp = F.pad(x, (1, 1, 1, 1))
    ys = torch.arange(h)[:, None].expand(h, wd)
    xs = torch.arange(wd)[None, :].expand(h, wd)
    out = None
    for dy in (-1, 0, 1):
        for dx in (-1, 0, 1):
            sh = xp[:, :, 1 + dy:1 + dy + h, 1 + dx:1 + dx + wd]
            if dy != 0 and dx != 0:
                cross_r = torch.div(ys + dy + p, p, rounding_mode="floor") != torch.div(ys + p, p, rounding_mode="floor")
                cross_c = torch.div(xs + dx + p, p, rounding_mode="floor") != torch.div(xs + p, p, rounding_mode="floor")
                repl = xp[:, :, 1:1 + h, 1 + dx:1 + dx + wd]          # pixel (y, x+dx); zero outside the image
                sh = torch.where((cross_r & cross_c)[None, None], repl, sh)
            t = torch.einsum("nchw,oc->nohw", sh, w[:, :, dy + 1, dx + 1])
            out = t if out is None else out + t
    out = out + b[None, :, None, None]
    return out[:, :, ::stride, ::stride] if stride != 1 else out


# --------------------------------------------------------------------------------------
# blocks (whole-image semantics; gn_level_patch = patch edge at this level or None)
# --------------------------------------------------------------------------------------
def resnet_block(P, p, x, emb, cfg: UNetConfig, gn_patch, trace=None, corner=False):
    """modules/resnet.py:390-460 (time_embedding_norm == 'default', output_scale_factor 1)."""
    g = cfg.norm_num_groups
    cp = gn_patch if corner else None
    h = F.silu(_gn(x, g, P[f"{p}.norm1.weight"], P[f"{p}.norm1.bias"], cfg.norm_eps, gn_patch))
    h = conv3x3(h, P[f"{p}.conv1.weight"], P[f"{p}.conv1.bias"], 1, cp)
    t = F.linear(F.silu(emb), P[f"{p}.time_emb_proj.weight"], P[f"{p}.time_emb_proj.bias"])
    h = h + t[:, :, None, None]
    h = F.silu(_gn(h, g, P[f"{p}.norm2.weight"], P[f"{p}.norm2.bias"], cfg.norm_eps, gn_patch))
    h = conv3x3(h, P[f"{p}.conv2.weight"], P[f"{p}.conv2.bias"], 1, cp)
    if f"{p}.conv_shortcut.weight" in P:
        x = F.conv2d(x, P[f"{p}.conv_shortcut.weight"], P[f"{p}.conv_shortcut.bias"])
    out = x + h
    if trace is not None:
        trace[p] = out
    return out


def basic_transformer_block(P, b, x, ctx, heads, cfg: UNetConfig):
    """modules/transformer.py:167-290 with norm_type == 'layer_norm'."""
    dim = x.shape[-1]
    eps = cfg.layer_norm_eps
    n = F.layer_norm(x, (dim,), P[f"{b}.norm1.weight"], P[f"{b}.norm1.bias"], eps)
    q = F.linear(n, P[f"{b}.attn1.to_q.weight"])
    k = F.linear(n, P[f"{b}.attn1.to_k.weight"])
    v = F.linear(n, P[f"{b}.attn1.to_v.weight"])
    a = attention(q, k, v, heads)
    x = F.linear(a, P[f"{b}.attn1.to_out.0.weight"], P[f"{b}.attn1.to_out.0.bias"]) + x
    n = F.layer_norm(x, (dim,), P[f"{b}.norm2.weight"], P[f"{b}.norm2.bias"], eps)
    q = F.linear(n, P[f"{b}.attn2.to_q.weight"])
    k = F.linear(ctx, P[f"{b}.attn2.to_k.weight"])
    v = F.linear(ctx, P[f"{b}.attn2.to_v.weight"])
    a = attention(q, k, v, heads)
    x = F.linear(a, P[f"{b}.attn2.to_out.0.weight"], P[f"{b}.attn2.to_out.0.bias"]) + x
    n = F.layer_norm(x, (dim,), P[f"{b}.norm3.weight"], P[f"{b}.norm3.bias"], eps)
    hg = F.linear(n, P[f"{b}.ff.net.0.proj.weight"], P[f"{b}.ff.net.0.proj.bias"])
    hid, gate = hg.chunk(2, dim=-1)  # diffusers GEGLU: hidden * gelu(gate)
    f = hid * F.gelu(gate)
    x = F.linear(f, P[f"{b}.ff.net.2.weight"], P[f"{b}.ff.net.2.bias"]) + x
    return x


def transformer_2d(P, p, x, ctx, heads, layers, cfg: UNetConfig, gn_patch, trace=None):
    """modules/transformer.py:32-128 (is_input_continuous, use_linear_projection=True)."""
    n, c, h, w = x.shape
    res = x
    y = _gn(x, cfg.norm_num_groups, P[f"{p}.norm.weight"], P[f"{p}.norm.bias"], cfg.transformer_norm_eps, gn_patch)
    y = y.permute(0, 2, 3, 1).reshape(n, h * w, c)
    y = F.linear(y, P[f"{p}.proj_in.weight"], P[f"{p}.proj_in.bias"])
    for k in range(layers):
        y = basic_transformer_block(P, f"{p}.transformer_blocks.{k}", y, ctx, heads, cfg)
    y = F.linear(y, P[f"{p}.proj_out.weight"], P[f"{p}.proj_out.bias"])
    y = y.reshape(n, h, w, c).permute(0, 3, 1, 2)
    out = y + res
    if trace is not None:
        trace[p] = out
    return out


def time_and_aug_embedding(P, cfg: UNetConfig, timestep, text_embeds, time_ids):
    """unet.py:314-334 -> diffusers get_time_embed / time_embedding / get_aug_embed ('text_time')."""
    dt = P["time_embedding.linear_1.weight"].dtype       # (diffusers: the sinusoids are fp32, then cast to the model dtype)
    t_emb = timestep_embedding(timestep, cfg.block_out_channels[0]).to(dt)
    emb = F.linear(t_emb, P["time_embedding.linear_1.weight"], P["time_embedding.linear_1.bias"])
    emb = F.linear(F.silu(emb), P["time_embedding.linear_2.weight"], P["time_embedding.linear_2.bias"])
    b = text_embeds.shape[0]
    tid = timestep_embedding(time_ids.reshape(-1), cfg.addition_time_embed_dim).reshape(b, -1)
    add = torch.cat([text_embeds.to(torch.float32), tid], dim=-1).to(dt)
    aug = F.linear(add, P["add_embedding.linear_1.weight"], P["add_embedding.linear_1.bias"])
    aug = F.linear(F.silu(aug), P["add_embedding.linear_2.weight"], P["add_embedding.linear_2.bias"])
    return emb + aug


def unet_forward(P: Dict[str, torch.Tensor], cfg: UNetConfig, sample: torch.Tensor, timestep: torch.Tensor,
                 encoder_hidden_states: torch.Tensor, text_embeds: torch.Tensor, time_ids: torch.Tensor,
                 gn_patch: Optional[int] = None, trace: Optional[dict] = None,
                 sliced_corners: bool = False, compute_dtype: torch.dtype = torch.float32, device=None,
                 sdpa: bool = False) -> torch.Tensor:
    """One UNet forward on whole latents.

    sample [B, C_in, H, W] fp32; timestep [B]; encoder_hidden_states [B, 77, ctx];
    text_embeds [B, text_dim]; time_ids [B, 6].  ``gn_patch`` (latent pixels at the top level)
    switches every GroupNorm to the patch-averaged statistics of the reference's sliced path
    (patch edge halves at each downsample); ``None`` = exact GroupNorm = the ``is_sliced=False``
    branch (unet.py:261-272).  ``sliced_corners`` additionally applies the halo-corner rule of
    ``conv3x3`` to every 3x3 conv after conv_in; ``gn_patch=p, sliced_corners=True`` is the whole-image
    equivalent of ``patch_ref.unet_forward_sliced`` (tests/test_oracle.py checks the two agree).
    """
    # compute_dtype / device / sdpa: the SAME graph evaluated by stock torch ops in bf16 or fp16 on the GPU box -- the yardstick the parity
    # tolerances are calibrated against (tests/test_parity_calibration_gpu.py); the oracle proper is the fp32 default.
    dev = torch.device(device) if device is not None else sample.device
    P = {k: v.to(device=dev, dtype=compute_dtype) for k, v in P.items()}
    x = sample.to(device=dev, dtype=compute_dtype)
    ctx = encoder_hidden_states.to(device=dev, dtype=compute_dtype)
    _LOWP["sdpa"] = bool(sdpa)
    emb = time_and_aug_embedding(P, cfg, timestep.to(dev), text_embeds.to(dev), time_ids.to(dev))
    ch = cfg.block_out_channels
    nlev = len(ch)

    def lp(level):  # patch edge at a resolution level
        return None if gn_patch is None else max(gn_patch >> level, 1)

    x = F.conv2d(x, P["conv_in.weight"], P["conv_in.bias"], padding=1)
    if trace is not None:
        trace["conv_in"] = x
    skips = [x]
    for i in range(nlev):
        for j in range(cfg.layers_per_block):
            x = resnet_block(P, f"down_blocks.{i}.resnets.{j}", x, emb, cfg, lp(i), trace, sliced_corners)
            if cfg.down_has_attn[i]:
                x = transformer_2d(P, f"down_blocks.{i}.attentions.{j}", x, ctx, cfg.num_heads[i],
                                   cfg.transformer_layers_per_block[i], cfg, lp(i), trace)
            skips.append(x)
        if i != nlev - 1:
            x = conv3x3(x, P[f"down_blocks.{i}.downsamplers.0.conv.weight"],
                        P[f"down_blocks.{i}.downsamplers.0.conv.bias"], 2, lp(i) if sliced_corners else None)
            if trace is not None:
                trace[f"down_blocks.{i}.downsamplers.0"] = x
            skips.append(x)
    top = nlev - 1
    x = resnet_block(P, "mid_block.resnets.0", x, emb, cfg, lp(top), trace, sliced_corners)
    x = transformer_2d(P, "mid_block.attentions.0", x, ctx, cfg.num_heads[-1],
                       cfg.transformer_layers_per_block[-1], cfg, lp(top), trace)
    x = resnet_block(P, "mid_block.resnets.1", x, emb, cfg, lp(top), trace, sliced_corners)
    rev_attn = list(reversed(cfg.down_has_attn))
    rev_layers = list(reversed(cfg.transformer_layers_per_block))
    rev_heads = list(reversed(cfg.num_heads))
    for i in range(nlev):
        level = nlev - 1 - i
        for j in range(cfg.layers_per_block + 1):
            x = torch.cat([x, skips.pop()], dim=1)
            x = resnet_block(P, f"up_blocks.{i}.resnets.{j}", x, emb, cfg, lp(level), trace, sliced_corners)
            if rev_attn[i]:
                x = transformer_2d(P, f"up_blocks.{i}.attentions.{j}", x, ctx, rev_heads[i], rev_layers[i],
                                   cfg, lp(level), trace)
        if i != nlev - 1:
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
            x = conv3x3(x, P[f"up_blocks.{i}.upsamplers.0.conv.weight"],
                        P[f"up_blocks.{i}.upsamplers.0.conv.bias"], 1, lp(level - 1) if sliced_corners else None)
            if trace is not None:
                trace[f"up_blocks.{i}.upsamplers.0"] = x
    x = F.silu(_gn(x, cfg.norm_num_groups, P["conv_norm_out.weight"], P["conv_norm_out.bias"], cfg.norm_eps, lp(0)))
    x = conv3x3(x, P["conv_out.weight"], P["conv_out.bias"], 1, lp(0) if sliced_corners else None)
    _LOWP["sdpa"] = False
    return x


def make_inputs(cfg: UNetConfig, batch: int, latent_hw: int, seed: int = 10086, ctx_len: int = 77):
    """Seeded synthetic step inputs of the shapes in SURVEY.md §8d."""
    g = torch.Generator().manual_seed(seed + 1)
    sample = torch.randn(batch, cfg.in_channels, latent_hw, latent_hw, generator=g)
    ehs = torch.randn(batch, ctx_len, cfg.cross_attention_dim, generator=g)
    text = torch.randn(batch, cfg.text_embed_dim, generator=g)
    px = float(latent_hw * 8)
    time_ids = torch.tensor([[px, px, 0.0, 0.0, px, px]]).repeat(batch, 1)
    timestep = torch.full((batch,), 801.0)
    return sample, timestep, ehs, text, time_ids
