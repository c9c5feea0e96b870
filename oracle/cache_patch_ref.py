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


# =========================================================================================================================================
# SD3 / SD3.5 MMDiT: the same cache around the joint blocks (SD3Transformer.py:140-228; transformer.py:299-388; attention.py:241-424)
# =========================================================================================================================================
class CachedSlicedMMDiTRef:
    """``PatchSD3Transformer2DModel.forward(..., is_sliced=True)`` with ESYMRED_USE_CACHE=TRUE, restated on whole token sequences (the sliced
    branch only re-chunks the token axis: every latent is cut into (res // patch_size)^2 equal TOKEN RANGES keyed "<request id>-<k>",
    modules/utils.py:86-122; oracle/sd3_mmdit_ref.split_sample_sd3 is pinned against it):

    * per joint block ``state_mask = state_input[i].get_sd3_mask(chunk keys, hidden chunks, i, timestep per chunk)`` -- per chunk the mean squared
      difference to the cached input chunk, predictor, forced run after TWO reuses (cache_manager.py:163-191); the block runs when any chunk of
      the batch asks, else the image stream comes from ``state_output[i]`` and the text stream from ``encoder_output[i]`` (SD3Transformer.py:151-228);
    * inside a running block everything runs on all tokens EXCEPT the attention (attention.py:296-372): a RESOLUTION none of whose chunks asks
      skips its joint attention -- its image tokens take ``attn.output``'s cached ``to_out`` result and its text tokens ``attn.encoder_output``'s
      cached ``to_add_out`` result -- while a resolution with any asking chunk is computed whole (``mask[start:end] = True``); the image-only
      ``attn2`` of the dual blocks does the same, except that a resolution whose asking ratio is <= 1/16 computes the asking chunks' queries only
      (against all keys of their latent) and keeps the cache for its other chunks (:303-325)."""

    def __init__(self, P, cfg, predictor, forced_after: int = 2):
        from . import sd3_mmdit_ref as m
        self.m = m
        self.P = {k: v.to(torch.float32) for k, v in P.items()}
        self.cfg = cfg
        self.io = [BlockIO(predictor, forced_after) for _ in range(cfg.num_layers)]
        self.enc_out = [None] * cfg.num_layers
        self.ops: Dict[str, OpCache] = {}
        self.masks, self.features, self.blocks_run = [], [], []

    def op(self, name):
        return self.ops.setdefault(name, OpCache())

    def _attention(self, name, xin, cin, groups, keys, enc_keys, mask, last):
        """xin: per group [n, L, d]; cin [sum n, Lt, d] or None (attn2).  Returns (image attention output per group, context output or None)."""
        m, P, cfg = self.m, self.P, self.cfg
        mask = mask.copy()
        enc_mask = np.zeros(len(enc_keys), dtype=bool)
        skip = False
        new_rows, new_enc = [], []
        ci = 0                                            # chunk cursor
        si = 0                                            # sample cursor
        for g, (n, nc) in enumerate(groups):
            start, end = ci, ci + n * nc
            x = xin[g]
            L = x.shape[1]
            if mask[start:end].sum() == 0:
                skip = True
            else:
                ratio = mask[start:end].sum() / (end - start)
                c = cin[si:si + n] if cin is not None else None
                xo, co = m.joint_attention(P, name, x, c, cfg, last if cin is not None else True)
                chunks = xo.reshape(n * nc, L // nc, -1)
                if cin is None and ratio <= 1 / 16:       # attention.py:303-325: the asking chunks' queries only, the cache for the rest
                    skip = True
                    new_rows.append(chunks[torch.from_numpy(mask[start:end])])
                else:
                    mask[start:end] = True
                    new_rows.append(chunks)
                    if cin is not None:
                        enc_mask[si:si + n] = True
                        if co is not None:
                            new_enc.append(co)
            ci, si = end, si + n
        new = torch.cat(new_rows) if new_rows else torch.zeros(0)
        if not skip:                                      # update_and_return(..., skip=False): the cache is simply replaced
            self.op(name + ".out").cache = {k: new[i] for i, k in enumerate(keys)}
            out = new
        else:
            out = self.op(name + ".out").update_and_return(keys, new, mask)
        co_all = None
        if cin is not None and not last:
            newe = torch.cat(new_enc) if new_enc else torch.zeros(0)
            if not skip:
                self.op(name + ".enc").cache = {k: newe[i] for i, k in enumerate(enc_keys)}
                co_all = newe
            else:
                co_all = self.op(name + ".enc").update_and_return(enc_keys, newe, enc_mask)
        # back to per-group [n, L, d]
        outs, ci = [], 0
        for g, (n, nc) in enumerate(groups):
            outs.append(out[ci:ci + n * nc].reshape(n, -1, out.shape[-1]))
            ci += n * nc
        return outs, co_all

    def _block(self, i, b, xs, ctx, temb_g, temb, groups, keys, enc_keys, mask, last, dual):
        m, P, cfg = self.m, self.P, self.cfg
        eps = cfg.norm_eps
        F_ = F
        mods = [F_.linear(F_.silu(t), P[f"{b}.norm1.linear.weight"], P[f"{b}.norm1.linear.bias"]) for t in temb_g]
        xin, x2in, gates = [], [], []
        for x, mod in zip(xs, mods):
            nx = m._ln(x, eps)
            if dual:
                sh, sc, gate, sh_mlp, sc_mlp, g_mlp, sh2, sc2, gate2 = mod.chunk(9, dim=1)
                x2in.append(nx * (1 + sc2[:, None]) + sh2[:, None])
            else:
                sh, sc, gate, sh_mlp, sc_mlp, g_mlp = mod.chunk(6, dim=1)
                gate2 = None
            xin.append(nx * (1 + sc[:, None]) + sh[:, None])
            gates.append((gate, sh_mlp, sc_mlp, g_mlp, gate2))
        cmod = F_.linear(F_.silu(temb), P[f"{b}.norm1_context.linear.weight"], P[f"{b}.norm1_context.linear.bias"])
        if last:
            c_sc, c_sh = cmod.chunk(2, dim=1)
        else:
            c_sh, c_sc, c_gate, c_sh_mlp, c_sc_mlp, c_g_mlp = cmod.chunk(6, dim=1)
        cin = m._ln(ctx, eps) * (1 + c_sc[:, None]) + c_sh[:, None]
        ao, co = self._attention(f"{b}.attn", xin, cin, groups, keys, enc_keys, mask, last)
        xs = [x + g_[0][:, None] * a for x, g_, a in zip(xs, gates, ao)]
        if dual:
            ao2, _ = self._attention(f"{b}.attn2", x2in, None, groups, keys, enc_keys, mask, True)
            xs = [x + g_[4][:, None] * a for x, g_, a in zip(xs, gates, ao2)]
        out = []
        for x, g_ in zip(xs, gates):
            nh = m._ln(x, eps) * (1 + g_[2][:, None]) + g_[1][:, None]
            out.append(x + g_[3][:, None] * m._ff(P, f"{b}.ff", nh))
        if last:
            return out, None
        ctx = ctx + c_gate[:, None] * co
        nc_ = m._ln(ctx, eps) * (1 + c_sc_mlp[:, None]) + c_sh_mlp[:, None]
        ctx = ctx + c_g_mlp[:, None] * m._ff(P, f"{b}.ff_context", nc_)
        return out, ctx

    def forward(self, ids: Dict[str, List[str]], latents: Dict[str, torch.Tensor], timestep, encoder_hidden_states, pooled, patch_size: int):
        m, P, cfg = self.m, self.P, self.cfg
        ps, d = cfg.patch_size, cfg.dim
        t = F.linear(m.timestep_embedding(timestep, 256), P["time_text_embed.timestep_embedder.linear_1.weight"], P["time_text_embed.timestep_embedder.linear_1.bias"])
        t = F.linear(F.silu(t), P["time_text_embed.timestep_embedder.linear_2.weight"], P["time_text_embed.timestep_embedder.linear_2.bias"])
        pp = F.linear(pooled.to(torch.float32), P["time_text_embed.text_embedder.linear_1.weight"], P["time_text_embed.text_embedder.linear_1.bias"])
        pp = F.linear(F.silu(pp), P["time_text_embed.text_embedder.linear_2.weight"], P["time_text_embed.text_embedder.linear_2.bias"])
        temb = t + pp
        xs, groups, keys, enc_keys, temb_g, tpp, shapes = [], [], [], [], [], [], []
        row = 0
        for res, lat in latents.items():
            n, _c, hh, ww = lat.shape
            h, w = hh // ps, ww // ps
            x = F.conv2d(lat.to(torch.float32), P["pos_embed.proj.weight"], P["pos_embed.proj.bias"], stride=ps).flatten(2).transpose(1, 2)
            mm = cfg.pos_embed_max_size
            top, left = (mm - h) // 2, (mm - w) // 2
            x = x + P["pos_embed.pos_embed"].reshape(1, mm, mm, d)[:, top:top + h, left:left + w].reshape(1, h * w, d)
            nc = (int(res) // patch_size) ** 2
            xs.append(x); groups.append((n, nc)); shapes.append((n, hh, ww))
            temb_g.append(temb[row:row + n])
            for i in range(n):
                enc_keys.append(ids[res][i])
                keys += [f"{ids[res][i]}-{k}" for k in range(nc)]
                tpp += [float(timestep[row + i])] * nc
            row += n
        ctx = F.linear(encoder_hidden_states.to(torch.float32), P["context_embedder.weight"], P["context_embedder.bias"])
        self.masks.append([]); self.features.append([])
        ran = 0
        for i in range(cfg.num_layers):
            b = f"transformer_blocks.{i}"
            last, dual = i == cfg.num_layers - 1, i in cfg.dual_attention_layers
            io = self.io[i]
            per_chunk = [c for x, (n, nc) in zip(xs, groups) for c in x.reshape(n * nc, x.shape[1] // nc, d)]
            mse = [float(((c - io.cin[k][0]) ** 2).mean()) if k in io.cin else 0.0 for c, k in zip(per_chunk, keys)]
            mask, feat = io.mgr.get_mask(keys, mse, i, tpp, None)
            io.cin = {k: [c.clone()] for c, k in zip(per_chunk, keys)}
            self.masks[-1].append(mask.copy()); self.features[-1].append(feat)
            if mask.sum() > 0:
                xs_new, ctx_new = self._block(i, b, xs, ctx, temb_g, temb, groups, keys, enc_keys, mask, last, dual)
                ran |= 1 << i
                per_new = [c for x, (n, nc) in zip(xs_new, groups) for c in x.reshape(n * nc, x.shape[1] // nc, d)]
                io.cout = {k: [c.clone()] for c, k in zip(per_new, keys)}
                xs = xs_new
                if not last:
                    ctx = ctx_new
                    self.enc_out[i] = {k: ctx_new[j].clone() for j, k in enumerate(enc_keys)}
            else:
                flat = [io.cout[k][0] for k in keys]
                xs, ci = [], 0
                for (n, nc) in groups:
                    xs.append(torch.stack(flat[ci:ci + n * nc]).reshape(n, -1, d)); ci += n * nc
                if not last:
                    # SD3Transformer.py:219-226 keeps ONE tensor (the text stream of the block's last run, whatever batch that was: it cannot follow a
                    # batch whose composition changed); restated per request, as every other cache of the path is keyed
                    ctx = torch.stack([self.enc_out[i][k] for k in enc_keys])
        self.blocks_run.append(ran)
        outs, row = {}, 0
        for (res, lat), x, (n, hh, ww) in zip(latents.items(), xs, shapes):
            tg = temb[row:row + n]
            sc, sh = F.linear(F.silu(tg), P["norm_out.linear.weight"], P["norm_out.linear.bias"]).chunk(2, dim=1)
            y = m._ln(x, cfg.norm_eps) * (1 + sc[:, None]) + sh[:, None]
            y = F.linear(y, P["proj_out.weight"], P["proj_out.bias"])
            h, w = hh // ps, ww // ps
            y = y.reshape(n, h, w, ps, ps, cfg.out_channels)
            outs[res] = torch.einsum("nhwpqc->nchpwq", y).reshape(n, cfg.out_channels, h * ps, w * ps)
            row += n
        return outs
