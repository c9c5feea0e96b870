"""The model-slot adapter: presents ``PatchUNet.forward``'s signature
(sduss/model_executor/modules/unet.py:205-225, 521-530) over the MI355X step plan in libmxdenoise.so.

Installed where ``instantiate_pipeline`` puts ``PatchUNet(unet)``
(pipelines/stable_diffusion_xl/pipeline_stable_diffusion_xl_esymred.py:30-41) -- see INTEGRATION.md.
PyTorch is used for device memory and the current stream only; every FLOP runs in the HIP library.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional

import torch

from . import lib as _lib
from .config import UNetConfig
from .weights import PackedWeights, pack


class _Config(dict):
    """``.config`` as the pipeline reads it (attribute and item access)."""
    __getattr__ = dict.__getitem__


def _row_ids(ids, n_rows):
    """one id per row of a resolution's batch: the request ids repeat once per classifier-free-guidance half ([uncond..., cond...])"""
    return [f"{ids[i % len(ids)]}#{i // len(ids)}" for i in range(n_rows)]


class MxUNet:
    """Drop-in for ``PatchUNet``: ``forward(sample_dict, timestep, encoder_hidden_states, ..., added_cond_kwargs,
    return_dict=False, is_sliced, patch_size, input_indices) -> (dict,)``.

    Row order contract (same as the reference): resolutions in the dict's (ascending) order, the conditioning rows
    of all resolutions concatenated in that order (pipeline_..._esymred.py:275-276, 327-339).
    """

    def __init__(self, cfg: UNetConfig, params: Dict[str, torch.Tensor], device="cuda:0"):
        self.cfg = cfg
        self.device = torch.device(device)
        self.dtype = torch.bfloat16
        self._lib = _lib.load()
        cc = _lib.UNetConfigC()
        cc.in_channels, cc.out_channels = cfg.in_channels, cfg.out_channels
        cc.n_levels = len(cfg.block_out_channels)
        for i, v in enumerate(cfg.block_out_channels):
            cc.block_out_channels[i] = v
            cc.down_has_attn[i] = int(cfg.down_has_attn[i])
            cc.transformer_layers[i] = cfg.transformer_layers_per_block[i]
            cc.num_heads[i] = cfg.num_heads[i]
        cc.layers_per_block = cfg.layers_per_block
        cc.cross_attention_dim = cfg.cross_attention_dim
        cc.addition_time_embed_dim = cfg.addition_time_embed_dim
        cc.projection_class_embeddings_input_dim = cfg.projection_class_embeddings_input_dim
        cc.norm_num_groups = cfg.norm_num_groups
        cc.norm_eps, cc.transformer_norm_eps, cc.layer_norm_eps = cfg.norm_eps, cfg.transformer_norm_eps, cfg.layer_norm_eps
        self._handle = self._lib.mx_unet_create(C.byref(cc))
        if not self._handle:
            raise _lib.MxError("mx_unet_create: " + self._lib.mx_last_error().decode())
        self.weights = PackedWeights(pack(cfg, params), self.device)
        _lib.check(self._lib.mx_unet_set_weights(self._handle, self.weights.blob.data_ptr(), self.weights.blob.numel(),
                                                 self.weights.table, len(self.weights.names)), "mx_unet_set_weights")
        self.mixed_one_sequence = True     # False: one launch sequence per resolution (the round-2 form; A/B and tests)
        self.max_mixed_groups = _lib.MAX_SEGS
        self._ws_by_stream: Dict[int, Optional[torch.Tensor]] = {}
        self._ws_need = {}
        self._next_ctx_key = 0             # set_context_key: names the batch composition of the NEXT forward only
        self.config = _Config(in_channels=cfg.in_channels, time_cond_proj_dim=None,
                              addition_time_embed_dim=cfg.addition_time_embed_dim,
                              projection_class_embeddings_input_dim=cfg.projection_class_embeddings_input_dim,
                              sample_size=128, center_input_sample=False)

    def __del__(self):
        h = getattr(self, "_handle", None)
        if h:
            self._lib.mx_unet_destroy(h)
            self._handle = None

    def to(self, *args, **kwargs):
        return self

    def set_context_key(self, key: int) -> None:
        """Name the batch composition of the NEXT forward (mx_unet_set_context_key): forwards announced with the same non-zero key receive
        encoder_hidden_states of identical content, so the cross-attention K / V^T of all 70 layers are projected once per composition instead
        of once per step (the reference re-concatenates and re-projects them every step: pipeline_..._esymred.py:287-339, attention.py:59-110).
        One-shot: a forward that is not announced projects as before."""
        self._next_ctx_key = int(key)

    def _apply_ctx_key(self) -> None:
        key, self._next_ctx_key = self._next_ctx_key, 0
        _lib.check(self._lib.mx_unet_set_context_key(self._handle, key), "mx_unet_set_context_key")

    def context_stats(self):
        """(hits, misses) of the per-composition K / V^T store since the handle was created"""
        h, m = C.c_long(0), C.c_long(0)
        _lib.check(self._lib.mx_unet_context_stats(self._handle, C.byref(h), C.byref(m)), "mx_unet_context_stats")
        return h.value, m.value

    # -------------------------------------------------------------------------------------------------
    def _workspace(self, batch: int, h: int, w: int, ctx_len: int, stream: int) -> torch.Tensor:
        """grow-only arena, one per stream: launch sequences issued on different streams (pipeline.py runs the resolutions
        of a mixed batch concurrently) must not share scratch"""
        key = (batch, h, w, ctx_len)
        need = self._ws_need.get(key)
        if need is None:                      # a dry run of the whole plan: once per shape
            need = self._ws_need[key] = self._lib.mx_unet_workspace_bytes(self._handle, batch, h, w, ctx_len)
        if need == 0:
            raise _lib.MxError("mx_unet_workspace_bytes: " + self._lib.mx_last_error().decode())
        ws = self._ws_by_stream.get(stream)
        if ws is None or ws.numel() < need:
            self._ws_by_stream[stream] = None
            ws = self._ws_by_stream[stream] = torch.empty(need, dtype=torch.uint8, device=self.device)
        return ws

    def forward_one(self, sample: torch.Tensor, timestep: torch.Tensor, encoder_hidden_states: torch.Tensor,
                    text_embeds: torch.Tensor, time_ids: torch.Tensor, gn_patch: int = 0,
                    stage: Optional[str] = None, stage_shape=None) -> torch.Tensor:
        """One launch sequence over a batch of same-resolution latents [B, C, H, W] (any of fp32/fp16/bf16)."""
        assert sample.is_cuda and sample.ndim == 4
        sample = sample.contiguous()
        b, _c, h, w = sample.shape
        ctx_len = encoder_hidden_states.shape[1]
        ts = timestep.to(device=self.device, dtype=torch.float32).reshape(-1)
        if ts.numel() == 1:
            ts = ts.expand(b)
        ts = ts.contiguous()
        ehs = encoder_hidden_states.to(device=self.device, dtype=torch.bfloat16).contiguous()
        te = text_embeds.to(device=self.device, dtype=torch.bfloat16).contiguous()
        ti = time_ids.to(device=self.device, dtype=torch.float32).contiguous()
        assert ts.shape[0] == b and ehs.shape[0] == b and te.shape[0] == b and ti.shape == (b, 6)
        assert ehs.shape[2] == self.cfg.cross_attention_dim and te.shape[1] == self.cfg.text_embed_dim
        out = torch.empty((b, self.cfg.out_channels, h, w), dtype=sample.dtype, device=self.device)
        code = _lib.torch_dtype_code(sample.dtype)
        stream = _lib.current_stream()
        ws = self._workspace(b, h, w, ctx_len, int(stream or 0))
        if stage is None:
            self._apply_ctx_key()
            _lib.check(self._lib.mx_unet_forward(self._handle, stream, sample.data_ptr(), code, ts.data_ptr(), ehs.data_ptr(),
                                                 te.data_ptr(), ti.data_ptr(), out.data_ptr(), b, h, w, ctx_len, gn_patch,
                                                 ws.data_ptr(), ws.numel()), "mx_unet_forward")
            return out
        st = torch.empty(stage_shape, dtype=torch.bfloat16, device=self.device)
        _lib.check(self._lib.mx_unet_forward_trace(self._handle, stream, sample.data_ptr(), code, ts.data_ptr(), ehs.data_ptr(),
                                                   te.data_ptr(), ti.data_ptr(), out.data_ptr(), b, h, w, ctx_len, gn_patch,
                                                   ws.data_ptr(), ws.numel(), stage.encode(), st.data_ptr(),
                                                   st.numel() * 2), "mx_unet_forward_trace")
        return st

    def forward_mixed(self, samples: List[torch.Tensor], timestep: torch.Tensor, encoder_hidden_states: torch.Tensor,
                      text_embeds: torch.Tensor, time_ids: torch.Tensor, gn_patch: int = 0, stage: Optional[str] = None) -> List[torch.Tensor]:
        """ONE launch sequence over the latents of several resolutions (mx_unet_forward_mixed): ``samples[g]`` is [B_g, C, H_g, W_g]; the
        conditioning rows are those of all groups concatenated in list order.  What the reference's sliced branch does by cutting every latent
        into one patch batch (unet.py:104-185, 242-260).  With ``stage`` returns that stage's NHWC activation of all groups instead
        ([sum of pixels, C], tests)."""
        assert 1 <= len(samples) <= _lib.MAX_SEGS, f"a mixed batch holds up to {_lib.MAX_SEGS} resolutions"
        if not torch.is_tensor(timestep):
            timestep = torch.tensor([float(timestep)], device=self.device)
        samples = [x.contiguous() for x in samples]
        dt = samples[0].dtype
        assert all(x.is_cuda and x.ndim == 4 and x.dtype == dt for x in samples)
        btot = sum(x.shape[0] for x in samples)
        ctx_len = encoder_hidden_states.shape[1]
        ts = timestep.to(device=self.device, dtype=torch.float32).reshape(-1)
        if ts.numel() == 1:
            ts = ts.expand(btot)
        ts = ts.contiguous()
        ehs = encoder_hidden_states.to(device=self.device, dtype=torch.bfloat16).contiguous()
        te = text_embeds.to(device=self.device, dtype=torch.bfloat16).contiguous()
        ti = time_ids.to(device=self.device, dtype=torch.float32).contiguous()
        assert ts.shape[0] == btot and ehs.shape[0] == btot and te.shape[0] == btot and ti.shape == (btot, 6)
        outs = [torch.empty((x.shape[0], self.cfg.out_channels, x.shape[2], x.shape[3]), dtype=dt, device=self.device) for x in samples]
        groups = (_lib.UNetGroup * len(samples))()
        for g, (x, o) in enumerate(zip(samples, outs)):
            groups[g].latents, groups[g].out = x.data_ptr(), o.data_ptr()
            groups[g].batch, groups[g].H, groups[g].W = x.shape[0], x.shape[2], x.shape[3]
        key = ("mixed", tuple((x.shape[0], x.shape[2], x.shape[3]) for x in samples), ctx_len)
        need = self._ws_need.get(key)
        if need is None:
            need = self._ws_need[key] = self._lib.mx_unet_workspace_bytes_mixed(self._handle, groups, len(samples), ctx_len)
        if need == 0:
            raise _lib.MxError("mx_unet_workspace_bytes_mixed: " + self._lib.mx_last_error().decode())
        stream = _lib.current_stream()
        sk = int(stream or 0)
        ws = self._ws_by_stream.get(sk)
        if ws is None or ws.numel() < need:
            self._ws_by_stream[sk] = None
            ws = self._ws_by_stream[sk] = torch.empty(need, dtype=torch.uint8, device=self.device)
        code = _lib.torch_dtype_code(dt)
        if stage is None:
            self._apply_ctx_key()
            _lib.check(self._lib.mx_unet_forward_mixed(self._handle, stream, groups, len(samples), code, ts.data_ptr(), ehs.data_ptr(), te.data_ptr(),
                                                       ti.data_ptr(), ctx_len, gn_patch, ws.data_ptr(), ws.numel()), "mx_unet_forward_mixed")
            return outs
        st = torch.empty(64 << 20, dtype=torch.bfloat16, device=self.device)        # large enough for any stage of the test shapes
        _lib.check(self._lib.mx_unet_forward_mixed_trace(self._handle, stream, groups, len(samples), code, ts.data_ptr(), ehs.data_ptr(), te.data_ptr(),
                                                         ti.data_ptr(), ctx_len, gn_patch, ws.data_ptr(), ws.numel(), stage.encode(), st.data_ptr(),
                                                         st.numel() * 2), "mx_unet_forward_mixed_trace")
        return [st]

    def forward_one_cached(self, cache, sample: torch.Tensor, timestep: torch.Tensor, encoder_hidden_states: torch.Tensor,
                           text_embeds: torch.Tensor, time_ids: torch.Tensor, batch_key: int = 0, gn_patch: int = 0, row_ids=None) -> torch.Tensor:
        """forward_one through the block-skip cache (sduss_amd/block_cache.py BlockSkipCache; the reference's ESYMRED_USE_CACHE=TRUE
        path, cache_manager.py:101-161).  Approximate by design; forward_one never consults it."""
        assert sample.is_cuda and sample.ndim == 4
        sample = sample.contiguous()
        b, _c, h, w = sample.shape
        ctx_len = encoder_hidden_states.shape[1]
        ts = timestep.to(device=self.device, dtype=torch.float32).reshape(-1)
        if ts.numel() == 1:
            ts = ts.expand(b)
        ts = ts.contiguous()
        ehs = encoder_hidden_states.to(device=self.device, dtype=torch.bfloat16).contiguous()
        te = text_embeds.to(device=self.device, dtype=torch.bfloat16).contiguous()
        ti = time_ids.to(device=self.device, dtype=torch.float32).contiguous()
        assert ts.shape[0] == b and ehs.shape[0] == b and te.shape[0] == b and ti.shape == (b, 6)
        out = torch.empty((b, self.cfg.out_channels, h, w), dtype=sample.dtype, device=self.device)
        stream = _lib.current_stream()
        ws = self._workspace(b, h, w, ctx_len, int(stream or 0))
        desc = cache.bind(self, b, h, w, batch_key, row_ids=row_ids)
        rc = self._lib.mx_unet_forward_cached(self._handle, stream, sample.data_ptr(), _lib.torch_dtype_code(sample.dtype),
                                              ts.data_ptr(), ehs.data_ptr(), te.data_ptr(), ti.data_ptr(), out.data_ptr(), b, h, w,
                                              ctx_len, gn_patch, ws.data_ptr(), ws.numel(), desc)
        if rc:
            err = cache.error
            cache.invalidate()                # a forward that stopped part-way stored some blocks' rows and not others: nothing cached survives it
            if err is not None:
                raise err                     # the predictor's own exception, not the library's "predictor failed"
        _lib.check(rc, "mx_unet_forward_cached")
        cache.after_forward()
        return out

    def forward_mixed_cached(self, cache, samples: List[torch.Tensor], row_ids, timestep: torch.Tensor, encoder_hidden_states: torch.Tensor,
                             text_embeds: torch.Tensor, time_ids: torch.Tensor, gn_patch: int) -> List[torch.Tensor]:
        """forward_mixed through the block-skip cache at the reference's unit, the patch (block_cache.PatchSkipCache; mx_unet_forward_cached_mixed):
        ONE launch sequence over the latents of every resolution, one host decision per block for all their patches.  ``row_ids``: one id per
        sample in row order (request id + CFG half)."""
        assert 1 <= len(samples) <= _lib.MAX_SEGS and gn_patch > 0
        samples = [x.contiguous() for x in samples]
        dt = samples[0].dtype
        assert all(x.is_cuda and x.ndim == 4 and x.dtype == dt for x in samples)
        btot = sum(x.shape[0] for x in samples)
        ctx_len = encoder_hidden_states.shape[1]
        ts = timestep.to(device=self.device, dtype=torch.float32).reshape(-1)
        ts = (ts.expand(btot) if ts.numel() == 1 else ts).contiguous()
        ehs = encoder_hidden_states.to(device=self.device, dtype=torch.bfloat16).contiguous()
        te = text_embeds.to(device=self.device, dtype=torch.bfloat16).contiguous()
        ti = time_ids.to(device=self.device, dtype=torch.float32).contiguous()
        assert ts.shape[0] == btot and ehs.shape[0] == btot and te.shape[0] == btot and ti.shape == (btot, 6)
        outs = [torch.empty((x.shape[0], self.cfg.out_channels, x.shape[2], x.shape[3]), dtype=dt, device=self.device) for x in samples]
        groups = (_lib.UNetGroup * len(samples))()
        for g, (x, o) in enumerate(zip(samples, outs)):
            groups[g].latents, groups[g].out = x.data_ptr(), o.data_ptr()
            groups[g].batch, groups[g].H, groups[g].W = x.shape[0], x.shape[2], x.shape[3]
        shapes = tuple((x.shape[0], x.shape[2], x.shape[3]) for x in samples)
        key = ("mixed_cached", shapes, ctx_len, gn_patch)
        need = self._ws_need.get(key)
        if need is None:
            need = self._ws_need[key] = self._lib.mx_unet_workspace_bytes_cached_mixed(self._handle, groups, len(samples), ctx_len, gn_patch)
        if need == 0:
            raise _lib.MxError("mx_unet_workspace_bytes_cached_mixed: " + self._lib.mx_last_error().decode())
        stream = _lib.current_stream()
        sk = int(stream or 0)
        ws = self._ws_by_stream.get(sk)
        if ws is None or ws.numel() < need:
            self._ws_by_stream[sk] = None
            ws = self._ws_by_stream[sk] = torch.empty(need, dtype=torch.uint8, device=self.device)
        desc = cache.bind(self, shapes, row_ids, gn_patch)
        rc = self._lib.mx_unet_forward_cached_mixed(self._handle, stream, groups, len(samples), _lib.torch_dtype_code(dt), ts.data_ptr(), ehs.data_ptr(),
                                                    te.data_ptr(), ti.data_ptr(), ctx_len, gn_patch, ws.data_ptr(), ws.numel(), desc)
        if rc:
            err = cache.error
            cache.invalidate()
            if err is not None:
                raise err
        _lib.check(rc, "mx_unet_forward_cached_mixed")
        cache.after_forward()
        return outs

    def forward(self, sample: Dict[str, torch.Tensor], timestep, encoder_hidden_states: torch.Tensor,
                class_labels=None, timestep_cond=None, attention_mask=None, cross_attention_kwargs=None,
                added_cond_kwargs: Optional[dict] = None, down_block_additional_residuals=None,
                mid_block_additional_residual=None, down_intrablock_additional_residuals=None,
                encoder_attention_mask=None, return_dict: bool = True, record: bool = False, patch_size: int = None,
                is_sliced: bool = False, save_index: int = 0, input_indices: dict = None):
        # same argument contract as unet.py:229-238
        assert (class_labels is None and timestep_cond is None and attention_mask is None
                and cross_attention_kwargs is None and down_block_additional_residuals is None
                and mid_block_additional_residual is None and down_intrablock_additional_residuals is None
                and encoder_attention_mask is None)
        assert added_cond_kwargs is not None, "SDXL needs added_cond_kwargs (text_embeds, time_ids)"
        text_embeds, time_ids = added_cond_kwargs["text_embeds"], added_cond_kwargs["time_ids"]
        out: Dict[str, torch.Tensor] = {}
        row = 0
        ctx_key, self._next_ctx_key = self._next_ctx_key, 0      # set_context_key names THIS call; it reaches the library only where the call is one forward
        keys = [k for k in sample if sample[k] is not None and sample[k].shape[0] > 0]
        if not is_sliced:
            keys = keys[:1]  # the reference's unsliced branch runs the first resolution only (unet.py:268-272)
        if is_sliced and getattr(self, "_block_caches", None) is not None and len(keys) <= _lib.MAX_SEGS and patch_size is not None \
                and all(int(k) % patch_size == 0 and int(k) > patch_size for k in keys):
            # ESYMRED_USE_CACHE=TRUE with is_sliced=True: the cache at its reference unit, the patch; the resolutions in ONE launch sequence
            ids = input_indices or {}
            assert all(k in ids and len(ids[k]) > 0 and sample[k].shape[0] % len(ids[k]) == 0 for k in keys), \
                "the block-skip cache keys its state by input_indices[resolution] (cache_manager.py:105)"
            row_ids = [r for k in keys for r in _row_ids(ids[k], sample[k].shape[0])]
            if self._patch_cache is None:
                self._patch_cache = self._new_patch_cache()
            res = self.forward_mixed_cached(self._patch_cache, [sample[k] for k in keys], row_ids, timestep, encoder_hidden_states, text_embeds, time_ids,
                                            gn_patch=patch_size // 8)
            return (dict(zip(keys, res)),)
        if is_sliced and len(keys) > 1 and len(keys) <= _lib.MAX_SEGS and getattr(self, "_block_caches", None) is None and self.mixed_one_sequence:
            # the resolutions of a mixed batch as ONE launch sequence (the reference: one patch batch, unet.py:242-260)
            assert patch_size is not None and all(int(k) % patch_size == 0 for k in keys)
            self._next_ctx_key = ctx_key
            res = self.forward_mixed([sample[k] for k in keys], timestep, encoder_hidden_states, text_embeds, time_ids, gn_patch=patch_size // 8)
            return (dict(zip(keys, res)),)
        for key in keys:
            x = sample[key]
            n = x.shape[0]
            gn_patch = 0
            if is_sliced:
                assert patch_size is not None and int(key) % patch_size == 0
                gn_patch = patch_size // 8
            sl = slice(row, row + n)
            ts = timestep if (not torch.is_tensor(timestep) or timestep.ndim == 0) else timestep[sl]
            if not torch.is_tensor(ts):
                ts = torch.tensor([float(ts)], device=self.device)
            caches = getattr(self, "_block_caches", None)
            if caches is not None:                     # ESYMRED_USE_CACHE=TRUE (enable_block_cache)
                ids = (input_indices or {}).get(key)
                assert ids is not None and len(ids) > 0 and n % len(ids) == 0, "the block-skip cache keys its state by input_indices[resolution] (cache_manager.py:105)"
                bc = caches.get(key)
                if bc is None:
                    bc = caches[key] = self._new_block_cache()
                out[key] = self.forward_one_cached(bc, x, ts, encoder_hidden_states[sl], text_embeds[sl], time_ids[sl], gn_patch=gn_patch,
                                                   row_ids=_row_ids(ids, n))
            else:
                if len(keys) == 1:
                    self._next_ctx_key = ctx_key
                out[key] = self.forward_one(x, ts, encoder_hidden_states[sl], text_embeds[sl], time_ids[sl], gn_patch)
            row += n
        return (out,)

    def enable_block_cache(self, down, up=None, forced_after: Optional[int] = None, observe: bool = False) -> None:
        """Route forward() through the block-skip cache, one state per resolution key: what ESYMRED_USE_CACHE=TRUE does to the
        reference's model (cache_manager.py:46-50).  `down` / `up`: objects with .predict(features) (block_cache.py)."""
        from .block_cache import BlockSkipCache, FORCED_RUN_AFTER, PatchSkipCache
        fa = FORCED_RUN_AFTER if forced_after is None else forced_after
        self._new_block_cache = lambda: BlockSkipCache(down, up, forced_after=fa, observe=observe)
        self._new_patch_cache = lambda: PatchSkipCache(down, up, forced_after=fa)      # is_sliced=True: the patch unit, all resolutions in one sequence
        self._block_caches = {}
        self._patch_cache = None

    def disable_block_cache(self) -> None:
        self._block_caches = None
        self._patch_cache = None

    __call__ = forward

    @property
    def add_embedding(self):  # unet.py:533-535; the pipeline only reads .linear_1.in_features
        class _L:  # noqa
            in_features = self.cfg.projection_class_embeddings_input_dim
        class _A:  # noqa
            linear_1 = _L
        return _A
