"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED (the reference holds no vector for this path; its predictors are cuML
forests that cannot be unpickled here, so the decisions come from a predictor the test supplies).

CPU restatement of the reference's block-skip cache AT ITS OWN UNIT: ``ESYMRED_USE_CACHE=TRUE`` with ``is_sliced=True`` -- the only mode in which
the reference's cache works (the unsliced branches of ``update_and_return`` receive all rows with a partial mask and cannot broadcast) and the one
its mixed policies force (policy/FCFS_Mixed.py:69-70, policy/ESyMReD.py:446-447).  The unit is the 256-px PATCH: every cache is a dictionary keyed
``"<request id>-<h>-<w>"`` (modules/utils.py:37,60; unet.py:163).  Restated, literally per patch, on top of oracle/patch_ref.py:

* block wrappers (unet_2d_blocks.py:9-62 mid, :66-170 down, :180-382 up): ``mask = input.get_mask(keys, hidden_states, block, timestep, is_up
  [, res_tuple])`` -- per patch: mean squared difference to the cached input (and to each cached skip tensor), predictor, forced run after four
  reuses (cache_manager.py:101-161; bookkeeping = oracle/cache_ref.CacheManagerRef) -- the body runs when ANY patch of the batch asks, else every
  output comes from the block's output cache (``save_and_get_block_states`` / ``_tupple``, cache_manager.py:58-82);
* inside a running block (what a partial mask means, op by op):
    - ``PatchResnetBlock2D`` (resnet.py:390-460): norm1 / norm2 (GroupNorm over ALL patches: statistics, halos and SiLU on the fresh tensors),
      the 1x1 shortcut and the time-embedding add run for all patches; ``conv1`` and ``conv2`` run for the asking patches only and the others
      take that op's OWN cached output (``update_and_return``, cache_manager.py:84-99);
    - ``PatchDownsample2D`` / ``PatchUpsample2D`` (resnet.py:280-378): halo exchange on all, the conv for the asking patches, op cache for the rest;
    - ``PatchTransformer2DModel`` / ``PatchBasicTransformerBlock`` (transformer.py:32-128, 167-290): GroupNorm, proj_in, the three LayerNorms,
      the q / k / v projections of attn1 (``SplitLinear.forward`` returns before its mask branch, resnet.py:157-163), the GEGLU feed-forward
      (diffusers' FeedForward ignores ``mask``) and proj_out run for all patches;
    - ``PatchSelfAttention`` (attention.py:121-232): the asking patches' queries against ALL keys of their latent (fresh, whether or not the
      key's patch asked), ``to_out`` on those rows, op cache for the others, then the residual;
    - ``PatchCrossAttention`` (attention.py:59-110): ``to_q``, attention and ``to_out`` for the asking patches only, op cache for the others.
  So a patch that did not ask still sees fresh normalisation, fresh feed-forward and fresh residual paths; what it reuses is the output of the
  convolutions and of the two attention sub-blocks at its last run.
"""
from __future__ import annotations

from typing import Dict, List

import numpy as np
import torch
import torch.nn.functional as F

from . import patch_ref as pr
from .cache_ref import CacheManagerRef
from .sdxl_unet_ref import attention, time_and_aug_embedding


class OpCache:
    """CacheManager.update_and_return (cache_manager.py:84-99): rows of the asking patches are the new ones, the others the cached ones; the
    cache then holds the merged rows of exactly the keys of this call."""

    def __init__(self):
        self.cache: Dict[str, torch.Tensor] = {}

    def update_and_return(self, keys: List[str], new_output: torch.Tensor, mask: np.ndarray) -> torch.Tensor:
        rows, j = [], 0
        for i, k in enumerate(keys):
            if mask[i]:
                rows.append(new_output[j]); j += 1
            else:
                rows.append(self.cache[k])          # a patch that neither asked nor is cached: uninitialised memory in the reference (:90)
        out = torch.stack(rows)
        if mask.sum() != 0:
            self.cache = {k: out[i] for i, k in enumerate(keys)}
        return out


class BlockIO:
    """one block's `input` manager (decision + cached inputs) and its output caches"""

    def __init__(self, predictor, forced_after):
        self.mgr = CacheManagerRef(predictor, forced_after)
        self.cin: Dict[str, List[torch.Tensor]] = {}
        self.cout: Dict[str, List[torch.Tensor]] = {}


class CachedSlicedUNetRef:
    def __init__(self, P, cfg, down_predictor, up_predictor=None, forced_after: int = 4):
        self.P = {k: v.to(torch.float32) for k, v in P.items()}
        self.cfg = cfg
        nlev = len(cfg.block_out_channels)
        self.n_blocks = 2 * nlev + 1
        up = up_predictor if up_predictor is not None else down_predictor
        self.io = [BlockIO(up if b > nlev else down_predictor, forced_after) for b in range(self.n_blocks)]
        self.ops: Dict[str, OpCache] = {}
        self.masks: List[List[np.ndarray]] = []          # per forward: the per-patch run mask of every block
        self.features: List[List[np.ndarray]] = []
        self.blocks_run: List[int] = []

    def op(self, name: str) -> OpCache:
        return self.ops.setdefault(name, OpCache())

    # ---- decision (get_mask) with real tensors ----
    def _gate(self, b: int, keys, tpp, ins):
        io = self.io[b]
        n = len(keys)
        mse = [[0.0] * len(ins) for _ in range(n)]
        for i, k in enumerate(keys):
            if k in io.cin:
                for j, t in enumerate(ins):
                    mse[i][j] = float(((t[i] - io.cin[k][j]) ** 2).mean())          # MSELoss(reduction='none').mean(dim=(-1,-2,-3))
        is_up = len(ins) > 1
        mask, feat = io.mgr.get_mask(keys, [m[0] for m in mse], b, tpp, [m[1:] for m in mse] if is_up else None)
        io.cin = {k: [t[i].clone() for t in ins] for i, k in enumerate(keys)}
        self.masks[-1].append(mask.copy())
        self.features[-1].append(feat)
        return mask

    def _finish(self, b: int, keys, mask, outs):
        """save_and_get_block_states / _tupple: a block that ran caches and returns what it produced (for every patch)"""
        io = self.io[b]
        if mask.sum() == 0:
            return [torch.stack([io.cout[k][j] for k in keys]) for j in range(len(io.cout[keys[0]]))]
        io.cout = {k: [t[i].clone() for t in outs] for i, k in enumerate(keys)}
        return outs

    # ---- ops under a partial mask ----
    def _resnet(self, p, x, emb, c, mask, keys):
        P, cfg = self.P, self.cfg
        m = torch.from_numpy(mask)
        h = F.silu(pr._gn_fused(P, f"{p}.norm1", x, cfg.norm_eps, c))
        o1 = F.conv2d(h[m], P[f"{p}.conv1.weight"], P[f"{p}.conv1.bias"])
        h = self.op(p + ".conv1").update_and_return(keys, o1, mask)
        t = F.linear(F.silu(emb), P[f"{p}.time_emb_proj.weight"], P[f"{p}.time_emb_proj.bias"])
        h = h + t[:, :, None, None]
        h = F.silu(pr._gn_fused(P, f"{p}.norm2", h, cfg.norm_eps, c))
        o2 = F.conv2d(h[m], P[f"{p}.conv2.weight"], P[f"{p}.conv2.bias"])
        h = self.op(p + ".conv2").update_and_return(keys, o2, mask)
        if f"{p}.conv_shortcut.weight" in P:
            x = F.conv2d(x, P[f"{p}.conv_shortcut.weight"], P[f"{p}.conv_shortcut.bias"])
        return x + h

    def _transformer(self, p, x, ctx, heads, layers, c, mask, keys):
        P, cfg = self.P, self.cfg
        m = torch.from_numpy(mask)
        n, ch, h, w = x.shape
        res = x
        y = pr._gn_fused(P, f"{p}.norm", x, cfg.transformer_norm_eps, c, padding=False)
        y = y.permute(0, 2, 3, 1).reshape(n, h * w, ch)
        y = F.linear(y, P[f"{p}.proj_in.weight"], P[f"{p}.proj_in.bias"])
        eps = cfg.layer_norm_eps
        for k in range(layers):
            b = f"{p}.transformer_blocks.{k}"
            nn_ = F.layer_norm(y, (ch,), P[f"{b}.norm1.weight"], P[f"{b}.norm1.bias"], eps)
            q = F.linear(nn_, P[f"{b}.attn1.to_q.weight"])
            kk = F.linear(nn_, P[f"{b}.attn1.to_k.weight"])
            v = F.linear(nn_, P[f"{b}.attn1.to_v.weight"])
            a = pr._self_attention_regrouped(q, kk, v, heads, c)[m]          # asking queries, all keys of the latent (attention.py:156-203)
            o = F.linear(a, P[f"{b}.attn1.to_out.0.weight"], P[f"{b}.attn1.to_out.0.bias"])
            y = self.op(b + ".attn1").update_and_return(keys, o, mask) + y
            nn_ = F.layer_norm(y, (ch,), P[f"{b}.norm2.weight"], P[f"{b}.norm2.bias"], eps)
            q = F.linear(nn_[m], P[f"{b}.attn2.to_q.weight"])
            kk = F.linear(ctx[m], P[f"{b}.attn2.to_k.weight"])
            v = F.linear(ctx[m], P[f"{b}.attn2.to_v.weight"])
            a = attention(q, kk, v, heads)
            o = F.linear(a, P[f"{b}.attn2.to_out.0.weight"], P[f"{b}.attn2.to_out.0.bias"])
            y = self.op(b + ".attn2").update_and_return(keys, o, mask) + y
            nn_ = F.layer_norm(y, (ch,), P[f"{b}.norm3.weight"], P[f"{b}.norm3.bias"], eps)
            hid, gate = F.linear(nn_, P[f"{b}.ff.net.0.proj.weight"], P[f"{b}.ff.net.0.proj.bias"]).chunk(2, dim=-1)
            y = F.linear(hid * F.gelu(gate), P[f"{b}.ff.net.2.weight"], P[f"{b}.ff.net.2.bias"]) + y
        y = F.linear(y, P[f"{p}.proj_out.weight"], P[f"{p}.proj_out.bias"])
        return y.reshape(n, h, w, ch).permute(0, 3, 1, 2) + res

    def _sampler(self, name, x, c, mask, keys, stride, up):
        P = self.P
        m = torch.from_numpy(mask)
        if up:
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
        o = F.conv2d(pr.mock_groupnorm(x, c.padding_idx)[m], P[f"{name}.conv.weight"], P[f"{name}.conv.bias"], stride=stride)
        return self.op(name).update_and_return(keys, o, mask)

    # ---- the forward ----
    def forward(self, ids: Dict[str, List[str]], samples: Dict[str, torch.Tensor], timestep, encoder_hidden_states, text_embeds, time_ids,
                patch_size: int) -> Dict[str, torch.Tensor]:
        """ids / samples: per resolution (ascending), the request ids and latents [n, C, res/8, res/8]; conditioning rows in the same order."""
        P, cfg = self.P, self.cfg
        padding_idx, latent_offset, resolution_offset, x, patch_map, keys = pr.split_sample(
            {k: v.to(torch.float32) for k, v in samples.items()}, patch_size, ids)
        reps = torch.tensor([latent_offset[i + 1] - latent_offset[i] for i in range(len(latent_offset) - 1)])
        tpp = torch.repeat_interleave(timestep, reps, dim=0)
        ctx = torch.repeat_interleave(encoder_hidden_states.to(torch.float32), reps, dim=0)
        emb = time_and_aug_embedding(P, cfg, tpp, torch.repeat_interleave(text_embeds, reps, dim=0), torch.repeat_interleave(time_ids, reps, dim=0))
        c = pr._Ctx(cfg, latent_offset, resolution_offset, patch_map, padding_idx)
        tl = [float(t) for t in tpp]
        nlev = len(cfg.block_out_channels)
        self.masks.append([]); self.features.append([])
        ran = 0
        x = F.conv2d(x, P["conv_in.weight"], P["conv_in.bias"])
        skips = [x]
        for i in range(nlev):                                                # down blocks
            b = i
            mask = self._gate(b, keys, tl, [x])
            outs = []
            if mask.sum() != 0:
                for j in range(cfg.layers_per_block):
                    x = self._resnet(f"down_blocks.{i}.resnets.{j}", x, emb, c, mask, keys)
                    if cfg.down_has_attn[i]:
                        x = self._transformer(f"down_blocks.{i}.attentions.{j}", x, ctx, cfg.num_heads[i], cfg.transformer_layers_per_block[i], c, mask, keys)
                    outs.append(x)
                if i != nlev - 1:
                    x = self._sampler(f"down_blocks.{i}.downsamplers.0", x, c, mask, keys, 2, False)
                    outs.append(x)
                ran |= 1 << b
            outs = self._finish(b, keys, mask, outs)
            skips.extend(outs)
            x = outs[-1]
        b = nlev                                                             # mid block
        mask = self._gate(b, keys, tl, [x])
        if mask.sum() != 0:
            x = self._resnet("mid_block.resnets.0", x, emb, c, mask, keys)
            x = self._transformer("mid_block.attentions.0", x, ctx, cfg.num_heads[-1], cfg.transformer_layers_per_block[-1], c, mask, keys)
            x = self._resnet("mid_block.resnets.1", x, emb, c, mask, keys)
            ran |= 1 << b
        x = self._finish(b, keys, mask, [x])[0]
        rev_attn, rev_layers, rev_heads = list(reversed(cfg.down_has_attn)), list(reversed(cfg.transformer_layers_per_block)), list(reversed(cfg.num_heads))
        for i in range(nlev):                                                # up blocks
            b = nlev + 1 + i
            n_res = cfg.layers_per_block + 1
            res_tuple = skips[-n_res:]
            skips = skips[:-n_res]
            mask = self._gate(b, keys, tl, [x] + list(res_tuple))
            if mask.sum() != 0:
                res = list(res_tuple)
                for j in range(n_res):
                    x = torch.cat([x, res.pop()], dim=1)
                    x = self._resnet(f"up_blocks.{i}.resnets.{j}", x, emb, c, mask, keys)
                    if rev_attn[i]:
                        x = self._transformer(f"up_blocks.{i}.attentions.{j}", x, ctx, rev_heads[i], rev_layers[i], c, mask, keys)
                if i != nlev - 1:
                    x = self._sampler(f"up_blocks.{i}.upsamplers.0", x, c, mask, keys, 1, True)
                ran |= 1 << b
            x = self._finish(b, keys, mask, [x])[0]
        x = F.silu(pr._gn_fused(P, "conv_norm_out", x, cfg.norm_eps, c))
        x = F.conv2d(x, P["conv_out.weight"], P["conv_out.bias"])
        self.blocks_run.append(ran)
        return pr.concat_sample(patch_size, x, latent_offset)
