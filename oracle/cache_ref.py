"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): CPU restatement of the decision bookkeeping of the reference's CacheManager
(sduss/model_executor/modules/cache_manager.py:101-161, get_mask), with the tensors replaced by the numbers the bookkeeping
needs.  Checked against sduss_amd/block_cache.py in tests/test_cpu.py; parity of this file itself is unpinned (the reference
holds no fixture for it and its predictors cannot be unpickled here)."""
import sys

import numpy as np

MAX = float(sys.maxsize)                                  # cache_manager.py:19


class CacheManagerRef:
    """One manager = one block's `input` cache: which request ids it holds and each id's count of consecutive reuses."""

    def __init__(self, predictor, forced_after=4):        # 4: cache_manager.py:134,154; the SD3 variant uses 2 (:184)
        self.predictor = predictor
        self.forced_after = forced_after
        self.cache = set()                                # ids with a cached input (the reference keeps the tensors; :133,153)
        self.previous_mask = {}                           # :25

    def get_mask(self, new_indices, input_mse, total_blocks, timestep, res_mse=None):
        """new_indices: request ids in row order; input_mse[i] / res_mse[i][j]: what mse_loss(...).mean(...) would give for a cached id
        (ignored for an id that is not cached: the reference leaves MAX there, :110,139)."""
        n = len(new_indices)
        nres = 0 if res_mse is None else len(res_mse[0])
        C = np.full((n, 1 + nres), MAX)                   # :110 / :139
        for i, k in enumerate(new_indices):
            if k in self.cache:                           # :111-123 / :140-144
                C[i, 0] = input_mse[i]
                for j in range(nres):
                    C[i, 1 + j] = res_mse[i][j]
        feature = np.concatenate([np.full((n, 1), float(int(total_blocks))), np.asarray(timestep, dtype=np.float64).reshape(-1, 1), C], axis=1)  # :124-126 / :145-147
        self.previous_mask = {k: (0 if k not in self.cache else self.previous_mask[k]) for k in new_indices}   # :128-129 / :150-151
        mask = np.array(self.predictor.predict(feature)).astype(np.int64)      # :133 / :152
        self.cache = set(new_indices)                     # :131 / :153
        forced = np.array([self.previous_mask[k] == self.forced_after for k in new_indices])
        mask[forced] = 1                                  # :134 / :156
        self.previous_mask = {k: (0 if mask[i] == 1 or self.previous_mask[k] == self.forced_after else self.previous_mask[k] + 1)
                              for i, k in enumerate(new_indices)}             # :135 / :158
        return mask > 0.5, feature                        # :159


# ---------------------------------------------------------------------------------------------------------------------------------------
# The cached FORWARD at the SAMPLE unit: the seven block wrappers of modules/unet_2d_blocks.py around the oracle's blocks, with real tensors.
# (Round 4: this is the oracle of mx_unet_forward_cached's own unsliced unit -- "a sample that did not ask keeps its block outputs" -- not of the
# reference's behaviour inside a running block, which exists only sliced and per patch: oracle/cache_patch_ref.py restates that.)
# Restates, for the case of one patch per latent (is_sliced False: unet.py:261-272 keys every cache by request id, so a "patch" is a sample):
#   * PatchCrossAttnDownBlock2D / PatchDownBlock2D (unet_2d_blocks.py:66-170): mask = input.get_mask(...); the body runs when mask.sum() != 0;
#     output = save_and_get_block_states / _tupple (hidden state and the skip tensors the block emits);
#   * PatchUNetMidBlock2DCrossAttn (:9-62);
#   * PatchCrossAttnUpBlock2D / PatchUpBlock2D (:180-382): get_mask(..., is_upsample=True, res_tuple = the skips the block consumes);
#   * inside a running block this oracle (like mx_unet_forward_cached) RESTORES THE BLOCK'S OUTPUTS of a sample that did not ask.  Round 3
#     argued that this is what the reference's per-op caches give for non-interacting samples; it is not: in the reference only the
#     convolutions and the attention sub-blocks return cached outputs, while the normalisations, the feed-forward, the shortcut and the
#     residual adds of such a sample run on its fresh input (resnet.py:401-458, transformer.py:191-288), and its unsliced attention branches
#     cannot take a partial mask at all (attention.py:204-224).  So this is the library's own sample-unit semantics, stated as such.
# The mse features are the reference's: nn.MSELoss(reduction='none')(cached, new).mean(dim=(-1, -2, -3)) per sample.
# ---------------------------------------------------------------------------------------------------------------------------------------
class CachedUNetRef:
    """State across steps: per block the cached inputs / outputs per request id, and the decision managers."""

    def __init__(self, P, cfg, down_predictor, up_predictor=None, forced_after=4):
        import torch
        from . import sdxl_unet_ref as ref
        self.torch, self.ref = torch, ref
        self.P = {k: v.to(torch.float32) for k, v in P.items()}
        self.cfg = cfg
        nlev = len(cfg.block_out_channels)
        self.n_blocks = 2 * nlev + 1
        up = up_predictor if up_predictor is not None else down_predictor
        self.mgr = [CacheManagerRef(up if b > nlev else down_predictor, forced_after) for b in range(self.n_blocks)]
        self.cin = [dict() for _ in range(self.n_blocks)]         # id -> cached input tensors (list: hidden state [, skips])
        self.cout = [dict() for _ in range(self.n_blocks)]        # id -> cached output tensors (list: [skips..., ] hidden state)
        self.blocks_run = []
        self.features = []

    def _gate(self, b, ids, timestep, ins):
        """get_mask of block b on its input tensors `ins` ([B, ...] each); returns the run mask"""
        torch = self.torch
        n = len(ids)
        mse = [[0.0] * len(ins) for _ in range(n)]
        for i, k in enumerate(ids):
            if k in self.cin[b]:
                for j, t in enumerate(ins):
                    mse[i][j] = float(((t[i] - self.cin[b][k][j]) ** 2).mean())
        is_up = len(ins) > 1
        mask, feat = self.mgr[b].get_mask(ids, [m[0] for m in mse], b, timestep, [m[1:] for m in mse] if is_up else None)
        self.cin[b] = {k: [t[i].clone() for t in ins] for i, k in enumerate(ids)}        # the cached input is always the latest one (:133,153)
        self.features.append(feat)
        return mask

    def _merge(self, b, ids, mask, outs):
        """outputs of block b after a run with `mask`: rows that did not ask come from the cache; the cache then holds the merged rows
        (update_and_return per op == per block for non-interacting samples; save_and_get_block_states :60-67)"""
        torch = self.torch
        merged = []
        for j, t in enumerate(outs):
            t = t.clone()
            for i, k in enumerate(ids):
                if not mask[i] and k in self.cout[b]:
                    t[i] = self.cout[b][k][j]
            merged.append(t)
        self.cout[b] = {k: [t[i].clone() for t in merged] for i, k in enumerate(ids)}
        return merged

    def forward(self, ids, sample, timestep, encoder_hidden_states, text_embeds, time_ids):
        torch, ref, P, cfg = self.torch, self.ref, self.P, self.cfg
        import torch.nn.functional as F
        x = sample.to(torch.float32)
        ctx = encoder_hidden_states.to(torch.float32)
        emb = ref.time_and_aug_embedding(P, cfg, timestep, text_embeds, time_ids)
        ch = cfg.block_out_channels
        nlev = len(ch)
        x = F.conv2d(x, P["conv_in.weight"], P["conv_in.bias"], padding=1)
        skips = [x]
        ran = 0
        tl = [float(t) for t in timestep]

        def cached_outputs(b, n_out):
            return [torch.stack([self.cout[b][k][j] for k in ids]) for j in range(n_out)]

        for i in range(nlev):                                                # down blocks: unet_2d_blocks.py:66-170
            b = i
            mask = self._gate(b, ids, tl, [x])
            n_out = cfg.layers_per_block + (1 if i != nlev - 1 else 0)
            if mask.sum() != 0:
                outs = []
                for j in range(cfg.layers_per_block):
                    x = ref.resnet_block(P, f"down_blocks.{i}.resnets.{j}", x, emb, cfg, None)
                    if cfg.down_has_attn[i]:
                        x = ref.transformer_2d(P, f"down_blocks.{i}.attentions.{j}", x, ctx, cfg.num_heads[i], cfg.transformer_layers_per_block[i], cfg, None)
                    outs.append(x)
                if i != nlev - 1:
                    x = F.conv2d(x, P[f"down_blocks.{i}.downsamplers.0.conv.weight"], P[f"down_blocks.{i}.downsamplers.0.conv.bias"], stride=2, padding=1)
                    outs.append(x)
                outs = self._merge(b, ids, mask, outs)
                ran |= 1 << b
            else:
                outs = cached_outputs(b, n_out)
            skips.extend(outs)
            x = outs[-1]
        b = nlev                                                             # mid block: :9-62
        mask = self._gate(b, ids, tl, [x])
        if mask.sum() != 0:
            top = nlev - 1
            x = ref.resnet_block(P, "mid_block.resnets.0", x, emb, cfg, None)
            x = ref.transformer_2d(P, "mid_block.attentions.0", x, ctx, cfg.num_heads[-1], cfg.transformer_layers_per_block[-1], cfg, None)
            x = ref.resnet_block(P, "mid_block.resnets.1", x, emb, cfg, None)
            x = self._merge(b, ids, mask, [x])[0]
            ran |= 1 << b
        else:
            x = cached_outputs(b, 1)[0]
        rev_attn, rev_layers, rev_heads = list(reversed(cfg.down_has_attn)), list(reversed(cfg.transformer_layers_per_block)), list(reversed(cfg.num_heads))
        for i in range(nlev):                                                # up blocks: :180-382
            b = nlev + 1 + i
            n_res = cfg.layers_per_block + 1
            res_tuple = skips[-n_res:]                                       # oldest first; the block consumes res_tuple[-1] first (:250-257)
            skips = skips[:-n_res]
            mask = self._gate(b, ids, tl, [x] + list(res_tuple))
            if mask.sum() != 0:
                res = list(res_tuple)
                for j in range(n_res):
                    x = torch.cat([x, res.pop()], dim=1)
                    x = ref.resnet_block(P, f"up_blocks.{i}.resnets.{j}", x, emb, cfg, None)
                    if rev_attn[i]:
                        x = ref.transformer_2d(P, f"up_blocks.{i}.attentions.{j}", x, ctx, rev_heads[i], rev_layers[i], cfg, None)
                if i != nlev - 1:
                    x = F.interpolate(x, scale_factor=2.0, mode="nearest")
                    x = F.conv2d(x, P[f"up_blocks.{i}.upsamplers.0.conv.weight"], P[f"up_blocks.{i}.upsamplers.0.conv.bias"], padding=1)
                x = self._merge(b, ids, mask, [x])[0]
                ran |= 1 << b
            else:
                x = cached_outputs(b, 1)[0]
        x = F.silu(F.group_norm(x, cfg.norm_num_groups, P["conv_norm_out.weight"], P["conv_norm_out.bias"], cfg.norm_eps))
        x = F.conv2d(x, P["conv_out.weight"], P["conv_out.bias"], padding=1)
        self.blocks_run.append(ran)
        return x
