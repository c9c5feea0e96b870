"""Literal CPU restatement of the reference's *sliced* (patch) path and of its native op.

TEST INFRASTRUCTURE (see oracle/__init__.py).  PARITY UNPINNED.

Restated from (paths relative to /root/reference):
* ``PatchUNet.split_sample`` / ``concat_sample``  -- sduss/model_executor/modules/unet.py:104-202
* ``esymred_mp.groupnorm`` / ``mock_groupnorm``   -- modules/kernels/norm_silu_concat.cpp:66-101 and the four
  kernels of modules/kernels/norm_silu_concat.cu:41-386 (moments, cross-patch merge, affine + halo scatter,
  halo-only).  fp32 throughout: the reference stores the statistics in the tensor dtype (cpp:84-85), the
  oracle keeps them in fp32 and the tests state the tolerance.
* ``PatchGroupNorm.forward`` / ``get_adjacency``   -- modules/groupnorm.py:31-61
* sliced branches of ``PatchResnetBlock2D`` (resnet.py:390-460), ``PatchUpsample2D``/``PatchDownsample2D``
  (:280-378), ``PatchTransformer2DModel`` (transformer.py:32-128), ``PatchSelfAttention`` dense regroup branch
  (attention.py:176-201), ``PatchCrossAttention`` (:59-110); cache OFF (mask all True, cache_manager.py:59-60).

Noted behaviour of the reference that this file reproduces on purpose:
* halo CORNER cells are filled by the left/right neighbour with *its own* corner pixel
  (norm_silu_concat.cu:210-221, 228-239), not with the diagonal neighbour's pixel; rows written by the
  top/bottom neighbours cover columns 1..W only (:186-201).  So sliced 3x3 convs differ from whole-image convs
  at patch-junction corners (and at junctions on the image border, where the true value is the zero pad).
* cross-patch statistics are averaged (mean of means, mean of biased variances) -- cu:361-386.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

from .sdxl_unet_ref import (UNetConfig, attention, time_and_aug_embedding)


# --------------------------------------------------------------------------------------
# split / concat   (unet.py:104-202)
# --------------------------------------------------------------------------------------
def split_sample(samples: Dict[str, torch.Tensor], patch_size: int, input_indices: Optional[Dict[str, List[str]]] = None):
    """Returns (padding_idx[int32, 4N] (top,left,bottom,right), latent_offset, resolution_offset,
    patches [N, C, p+2, p+2], patch_map (1-based latent index per patch)); with ``input_indices`` (request ids per resolution) also the
    per-patch cache keys "<id>-<h>-<w>" (unet.py:163; modules/utils.py:37,60).  Pinned bit for bit against the reference function itself:
    tests/golden/ref_split_sample.npz (tests/test_ref_fixtures.py)."""
    latent_offset = [0]
    patch_map: List[int] = []
    resolution_offset = [0]
    padding_idx: List[List[int]] = []
    new_sample: List[torch.Tensor] = []
    keys: List[str] = []
    for resolution, res_sample in samples.items():
        res_key = resolution
        resolution = int(resolution)
        pn = resolution // patch_size
        lp = patch_size // 8
        if res_sample is None or res_sample.shape[0] == 0:
            continue
        for si, sample in enumerate(res_sample):
            latent_offset.append(latent_offset[-1] + pn ** 2)
            s = F.pad(sample, (1, 1, 1, 1), "constant", 0).unsqueeze(0)
            for h in range(pn):
                for w in range(pn):
                    pad = [0, 0, 0, 0]
                    cur = len(new_sample)
                    if pn == 1:
                        pad = [-1, -1, -1, -1]
                    else:
                        pad[1] = -1 if w == 0 else cur - 1
                        pad[3] = -1 if w == pn - 1 else cur + 1
                        pad[0] = -1 if h == 0 else cur - pn
                        pad[2] = -1 if h == pn - 1 else cur + pn
                    new_sample.append(s[:, :, h * lp:(h + 1) * lp + 2, w * lp:(w + 1) * lp + 2])
                    patch_map.append(len(latent_offset) - 1)
                    padding_idx.append(pad)
                    if input_indices is not None:
                        keys.append(input_indices[str(res_key)][si] + f"-{h}-{w}")
        resolution_offset.append(len(latent_offset) - 1)
    out = (torch.tensor(padding_idx, dtype=torch.int32).reshape(-1), latent_offset, resolution_offset,
           torch.cat(new_sample, dim=0), torch.tensor(patch_map, dtype=torch.int32))
    return out + (keys,) if input_indices is not None else out


def concat_sample(patch_size: int, new_sample: torch.Tensor, latent_offset: List[int]) -> Dict[str, torch.Tensor]:
    samples: Dict[str, List[torch.Tensor]] = {}
    for index in range(len(latent_offset) - 1):
        n = latent_offset[index + 1] - latent_offset[index]
        pn = int(math.sqrt(n))
        rows = []
        for h in range(pn):
            rows.append(torch.cat([new_sample[x].unsqueeze(0) for x in
                                   range(latent_offset[index] + h * pn, latent_offset[index] + (h + 1) * pn)], dim=-1))
        samples.setdefault(str(pn * patch_size), []).append(torch.cat(rows, dim=-2))
    return {k: torch.cat(v, dim=0) for k, v in samples.items()}


# --------------------------------------------------------------------------------------
# the native op   (norm_silu_concat.cu / .cpp)
# --------------------------------------------------------------------------------------
def halo_scatter(y: torch.Tensor, padding_idx: torch.Tensor) -> torch.Tensor:
    """Interior copy + neighbour halo scatter of NormSiluConcat/MockNormSiluConcat (cu:164-241, 285-357).
    y [N,C,H,W] -> [N,C,H+2,W+2], zero where nobody writes (torch::zeros, cpp:71,92)."""
    n, c, h, w = y.shape
    out = torch.zeros(n, c, h + 2, w + 2, dtype=y.dtype)
    out[:, :, 1:h + 1, 1:w + 1] = y
    pidx = padding_idx.reshape(n, 4).tolist()
    for b in range(n):
        top, left, bottom, right = pidx[b]
        if top != -1:
            out[top, :, h + 1, 1:w + 1] = y[b, :, 0, :]
        if bottom != -1:
            out[bottom, :, 0, 1:w + 1] = y[b, :, h - 1, :]
        if left != -1:
            out[left, :, 1:h + 1, w + 1] = y[b, :, :, 0]
            out[left, :, 0, w + 1] = y[b, :, 0, 0]
            out[left, :, h + 1, w + 1] = y[b, :, h - 1, 0]
        if right != -1:
            out[right, :, 1:h + 1, 0] = y[b, :, :, w - 1]
            out[right, :, 0, 0] = y[b, :, 0, w - 1]
            out[right, :, h + 1, 0] = y[b, :, h - 1, w - 1]
    return out


def patch_moments(x: torch.Tensor, cpg: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """RowwiseMomentsCUDAKernel (cu:41-81): per (patch, group) mean and BIASED variance."""
    n, c, h, w = x.shape
    xg = x.reshape(n, c // cpg, -1).to(torch.float32)
    return xg.mean(dim=2), xg.var(dim=2, unbiased=False)


def merge_moments(mean: torch.Tensor, var: torch.Tensor, eps: float, latent_offset: List[int],
                  patch_map: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """GetFullMeanAndRstd (cu:361-386), out of place (the in-place original races)."""
    m2 = torch.empty_like(mean)
    r2 = torch.empty_like(var)
    for b in range(mean.shape[0]):
        li = int(patch_map[b])
        lo, hi = latent_offset[li - 1], latent_offset[li]
        m2[b] = mean[lo:hi].mean(dim=0)
        r2[b] = torch.rsqrt(var[lo:hi].mean(dim=0) + eps)
    return m2, r2


def groupnorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, cpg: int, eps: float, padding: bool,
              latent_offset: List[int], patch_map: torch.Tensor, padding_idx: torch.Tensor) -> torch.Tensor:
    """``esymred_mp.groupnorm`` (cpp:75-101).  ``cpg`` is the host parameter the reference calls ``group``."""
    n, c, h, w = x.shape
    mean, var = patch_moments(x, cpg)
    mean, rstd = merge_moments(mean, var, eps, latent_offset, patch_map)
    g = c // cpg
    scale = rstd[:, :, None] * gamma.reshape(1, g, cpg)                     # cu:157
    shift = beta.reshape(1, g, cpg) - scale * mean[:, :, None]              # cu:158
    y = x.to(torch.float32) * scale.reshape(n, c, 1, 1) + shift.reshape(n, c, 1, 1)  # cu:163
    return halo_scatter(y, padding_idx) if padding else y


def mock_groupnorm(x: torch.Tensor, padding_idx: torch.Tensor) -> torch.Tensor:
    """``esymred_mp.mock_groupnorm`` / ``get_adjacency`` (cpp:66-74, groupnorm.py:31-33)."""
    return halo_scatter(x, padding_idx)


# --------------------------------------------------------------------------------------
# sliced UNet forward, literally per patch
# --------------------------------------------------------------------------------------
class _Ctx:
    def __init__(self, cfg, latent_offset, resolution_offset, patch_map, padding_idx):
        self.cfg = cfg
        self.latent_offset = latent_offset
        self.resolution_offset = resolution_offset
        self.patch_map = patch_map
        self.padding_idx = padding_idx


def _gn_fused(P, name, x, eps, c: _Ctx, padding=True):
    cfg = c.cfg
    return groupnorm(x, P[f"{name}.weight"], P[f"{name}.bias"], x.shape[1] // cfg.norm_num_groups, eps, padding,
                     c.latent_offset, c.patch_map, c.padding_idx)


def _resnet_sliced(P, p, x, emb, c: _Ctx):
    cfg = c.cfg
    h = F.silu(_gn_fused(P, f"{p}.norm1", x, cfg.norm_eps, c))              # resnet.py:401-402
    h = F.conv2d(h, P[f"{p}.conv1.weight"], P[f"{p}.conv1.bias"])           # pad 0 on halo'd input (:418, :113-133)
    t = F.linear(F.silu(emb), P[f"{p}.time_emb_proj.weight"], P[f"{p}.time_emb_proj.bias"])
    h = h + t[:, :, None, None]
    h = F.silu(_gn_fused(P, f"{p}.norm2", h, cfg.norm_eps, c))              # :429, :446
    h = F.conv2d(h, P[f"{p}.conv2.weight"], P[f"{p}.conv2.bias"])
    if f"{p}.conv_shortcut.weight" in P:
        x = F.conv2d(x, P[f"{p}.conv_shortcut.weight"], P[f"{p}.conv_shortcut.bias"])
    return x + h


def _self_attention_regrouped(q, k, v, heads, c: _Ctx):
    """attention.py:176-201: patches of one resolution are viewed as [latents, patches*L_p, C]."""
    outs = []
    lo_, ro = c.latent_offset, c.resolution_offset
    for r in range(len(ro) - 1):
        a, b = lo_[ro[r]], lo_[ro[r + 1]]
        nl = ro[r + 1] - ro[r]
        dim = q.shape[-1]
        o = attention(q[a:b].reshape(nl, -1, dim), k[a:b].reshape(nl, -1, dim), v[a:b].reshape(nl, -1, dim), heads)
        outs.append(o.reshape(b - a, -1, dim))
    return torch.cat(outs, dim=0)


def _transformer_sliced(P, p, x, ctx, heads, layers, c: _Ctx):
    from .sdxl_unet_ref import F as _F  # noqa
    cfg = c.cfg
    n, ch, h, w = x.shape
    res = x
    y = _gn_fused(P, f"{p}.norm", x, cfg.transformer_norm_eps, c, padding=False)   # transformer.py:53 (is_fused=False)
    y = y.permute(0, 2, 3, 1).reshape(n, h * w, ch)
    y = F.linear(y, P[f"{p}.proj_in.weight"], P[f"{p}.proj_in.bias"])
    eps = cfg.layer_norm_eps
    for k in range(layers):
        b = f"{p}.transformer_blocks.{k}"
        nn_ = F.layer_norm(y, (ch,), P[f"{b}.norm1.weight"], P[f"{b}.norm1.bias"], eps)
        q = F.linear(nn_, P[f"{b}.attn1.to_q.weight"])
        kk = F.linear(nn_, P[f"{b}.attn1.to_k.weight"])
        v = F.linear(nn_, P[f"{b}.attn1.to_v.weight"])
        a = _self_attention_regrouped(q, kk, v, heads, c)
        y = F.linear(a, P[f"{b}.attn1.to_out.0.weight"], P[f"{b}.attn1.to_out.0.bias"]) + y
        nn_ = F.layer_norm(y, (ch,), P[f"{b}.norm2.weight"], P[f"{b}.norm2.bias"], eps)
        q = F.linear(nn_, P[f"{b}.attn2.to_q.weight"])
        kk = F.linear(ctx, P[f"{b}.attn2.to_k.weight"])
        v = F.linear(ctx, P[f"{b}.attn2.to_v.weight"])
        a = attention(q, kk, v, heads)                                             # per patch, attention.py:59-110
        y = F.linear(a, P[f"{b}.attn2.to_out.0.weight"], P[f"{b}.attn2.to_out.0.bias"]) + y
        nn_ = F.layer_norm(y, (ch,), P[f"{b}.norm3.weight"], P[f"{b}.norm3.bias"], eps)
        hid, gate = F.linear(nn_, P[f"{b}.ff.net.0.proj.weight"], P[f"{b}.ff.net.0.proj.bias"]).chunk(2, dim=-1)
        y = F.linear(hid * F.gelu(gate), P[f"{b}.ff.net.2.weight"], P[f"{b}.ff.net.2.bias"]) + y
    y = F.linear(y, P[f"{p}.proj_out.weight"], P[f"{p}.proj_out.bias"])
    return y.reshape(n, h, w, ch).permute(0, 3, 1, 2) + res


def unet_forward_sliced(P, cfg: UNetConfig, samples: Dict[str, torch.Tensor], timestep, encoder_hidden_states,
                        text_embeds, time_ids, patch_size: int) -> Dict[str, torch.Tensor]:
    """``PatchUNet.forward(..., is_sliced=True, patch_size=patch_size)`` with the block cache off.

    ``samples``: {str(res): [n_res, C, res/8, res/8]} in ascending-res order; the per-latent conditioning rows
    follow the same order (unet.py:249-260 replicates them per patch).
    """
    P = {k: v.to(torch.float32) for k, v in P.items()}
    padding_idx, latent_offset, resolution_offset, x, patch_map = split_sample(
        {k: v.to(torch.float32) for k, v in samples.items()}, patch_size)
    reps = [latent_offset[i + 1] - latent_offset[i] for i in range(len(latent_offset) - 1)]
    rep = torch.tensor(reps)
    timestep = torch.repeat_interleave(timestep, rep, dim=0)
    ctx = torch.repeat_interleave(encoder_hidden_states.to(torch.float32), rep, dim=0)
    text_embeds = torch.repeat_interleave(text_embeds, rep, dim=0)
    time_ids = torch.repeat_interleave(time_ids, rep, dim=0)
    c = _Ctx(cfg, latent_offset, resolution_offset, patch_map, padding_idx)
    emb = time_and_aug_embedding(P, cfg, timestep, text_embeds, time_ids)
    ch = cfg.block_out_channels
    nlev = len(ch)
    x = F.conv2d(x, P["conv_in.weight"], P["conv_in.bias"])                        # padding 0 (unet.py:40, :344)
    skips = [x]
    for i in range(nlev):
        for j in range(cfg.layers_per_block):
            x = _resnet_sliced(P, f"down_blocks.{i}.resnets.{j}", x, emb, c)
            if cfg.down_has_attn[i]:
                x = _transformer_sliced(P, f"down_blocks.{i}.attentions.{j}", x, ctx, cfg.num_heads[i],
                                        cfg.transformer_layers_per_block[i], c)
            skips.append(x)
        if i != nlev - 1:
            x = F.conv2d(mock_groupnorm(x, padding_idx), P[f"down_blocks.{i}.downsamplers.0.conv.weight"],
                         P[f"down_blocks.{i}.downsamplers.0.conv.bias"], stride=2)  # resnet.py:364-368
            skips.append(x)
    x = _resnet_sliced(P, "mid_block.resnets.0", x, emb, c)
    x = _transformer_sliced(P, "mid_block.attentions.0", x, ctx, cfg.num_heads[-1],
                            cfg.transformer_layers_per_block[-1], c)
    x = _resnet_sliced(P, "mid_block.resnets.1", x, emb, c)
    rev_attn = list(reversed(cfg.down_has_attn))
    rev_layers = list(reversed(cfg.transformer_layers_per_block))
    rev_heads = list(reversed(cfg.num_heads))
    for i in range(nlev):
        for j in range(cfg.layers_per_block + 1):
            x = torch.cat([x, skips.pop()], dim=1)
            x = _resnet_sliced(P, f"up_blocks.{i}.resnets.{j}", x, emb, c)
            if rev_attn[i]:
                x = _transformer_sliced(P, f"up_blocks.{i}.attentions.{j}", x, ctx, rev_heads[i], rev_layers[i], c)
        if i != nlev - 1:
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")                  # resnet.py:316
            x = F.conv2d(mock_groupnorm(x, padding_idx), P[f"up_blocks.{i}.upsamplers.0.conv.weight"],
                         P[f"up_blocks.{i}.upsamplers.0.conv.bias"])                # :327-331
    x = F.silu(_gn_fused(P, "conv_norm_out", x, cfg.norm_eps, c))                   # unet.py:513-514
    x = F.conv2d(x, P["conv_out.weight"], P["conv_out.bias"])                       # :515-517 is_padding=False
    return concat_sample(patch_size, x, latent_offset)
