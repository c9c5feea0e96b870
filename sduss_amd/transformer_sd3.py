"""The model-slot adapter for SD3.5: presents ``PatchSD3Transformer2DModel.forward``'s signature
(sduss/model_executor/modules/SD3Transformer.py:60-74, 262) over the MI355X MMDiT step plan in libmxdenoise.so.
Installed where ``instantiate_pipeline`` wraps the transformer
(pipelines/stable_diffusion_3/pipeline_stable_diffusion_3_esymred.py:24-36)."""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import torch

from . import lib as _lib
from .config import MMDiTConfig
from .unet import _Config, _row_ids
from .weights import PackedWeights, pack_mmdit


def mmdit_config_c(cfg: MMDiTConfig) -> "_lib.MMDiTConfigC":
    cc = _lib.MMDiTConfigC()
    cc.patch_size, cc.in_channels, cc.out_channels = cfg.patch_size, cfg.in_channels, cfg.out_channels
    cc.num_layers, cc.num_attention_heads = cfg.num_layers, cfg.num_attention_heads
    cc.joint_attention_dim, cc.pooled_projection_dim = cfg.joint_attention_dim, cfg.pooled_projection_dim
    cc.pos_embed_max_size, cc.norm_eps = cfg.pos_embed_max_size, cfg.norm_eps
    assert cfg.attention_head_dim == 64, "the attention kernel is built for head_dim 64"
    for i in range(cfg.num_layers):
        cc.dual_attention[i] = int(i in cfg.dual_attention_layers)
    return cc


class MxSD3Transformer:
    """``forward(hidden_states: {str(res): [n,16,h,w]}, encoder_hidden_states [N,333,4096], pooled_projections [N,2048],
    timestep [N], ..., return_dict=False, is_sliced, patch_size, input_indices) -> (dict,)``.  Unlike the reference it does
    NOT mutate its input dict (the reference overwrites it with the patch embeddings, SD3Transformer.py:82-83)."""

    def __init__(self, cfg: MMDiTConfig, params: Dict[str, torch.Tensor], device="cuda:0"):
        self.cfg = cfg
        self.device = torch.device(device)
        self.dtype = torch.bfloat16
        self._lib = _lib.load()
        cc = mmdit_config_c(cfg)
        self._handle = self._lib.mx_mmdit_create(C.byref(cc))
        if not self._handle:
            raise _lib.MxError("mx_mmdit_create: " + self._lib.mx_last_error().decode())
        self.weights = PackedWeights(pack_mmdit(cfg, params), self.device)
        _lib.check(self._lib.mx_mmdit_set_weights(self._handle, self.weights.blob.data_ptr(), self.weights.blob.numel(),
                                                  self.weights.table, len(self.weights.names)), "mx_mmdit_set_weights")
        self._ws_by_stream = {}
        self._ws_need = {}
        self.mixed_one_sequence = True     # False: one launch sequence per resolution (the round-2 form; A/B and tests)
        self.max_mixed_groups = _lib.MAX_SEGS
        self.config = _Config(in_channels=cfg.in_channels, patch_size=cfg.patch_size, sample_size=cfg.sample_size,
                              joint_attention_dim=cfg.joint_attention_dim, pooled_projection_dim=cfg.pooled_projection_dim)

    def __del__(self):
        h = getattr(self, "_handle", None)
        if h:
            self._lib.mx_mmdit_destroy(h)
            self._handle = None

    def to(self, *args, **kwargs):
        return self

    def forward_one(self, latents: torch.Tensor, timestep: torch.Tensor, encoder_hidden_states: torch.Tensor,
                    pooled: torch.Tensor, stage: Optional[str] = None, stage_shape=None, cache=None, batch_key: int = 0, row_ids=None) -> torch.Tensor:
        """`cache` (sduss_amd/block_cache.py BlockSkipCache, forced_after=2) routes the step through mx_mmdit_forward_cached: the
        reference's ESYMRED_USE_CACHE=TRUE path (SD3Transformer.py:151-228).  Approximate by design; off by default."""
        assert latents.is_cuda and latents.ndim == 4
        latents = latents.contiguous()
        b, _c, h, w = latents.shape
        lt = encoder_hidden_states.shape[1]
        ts = timestep.to(device=self.device, dtype=torch.float32).reshape(-1)
        if ts.numel() == 1:
            ts = ts.expand(b)
        ts = ts.contiguous()
        ehs = encoder_hidden_states.to(device=self.device, dtype=torch.bfloat16).contiguous()
        pp = pooled.to(device=self.device, dtype=torch.bfloat16).contiguous()
        assert ts.shape[0] == b and ehs.shape[0] == b and pp.shape == (b, self.cfg.pooled_projection_dim)
        assert ehs.shape[2] == self.cfg.joint_attention_dim
        key = (b, h, w, lt)
        need = self._ws_need.get(key)
        if need is None:                      # a dry run of the whole plan: once per shape
            need = self._ws_need[key] = self._lib.mx_mmdit_workspace_bytes(self._handle, b, h, w, lt)
        if need == 0:
            raise _lib.MxError("mx_mmdit_workspace_bytes: " + self._lib.mx_last_error().decode())
        code = _lib.torch_dtype_code(latents.dtype)
        stream = _lib.current_stream()
        skey = int(stream or 0)               # one grow-only arena per stream (pipeline_sd3.py overlaps resolutions)
        ws = self._ws_by_stream.get(skey)
        if ws is None or ws.numel() < need:
            self._ws_by_stream[skey] = None
            ws = self._ws_by_stream[skey] = torch.empty(need, dtype=torch.uint8, device=self.device)
        out = torch.empty((b, self.cfg.out_channels, h, w), dtype=latents.dtype, device=self.device)
        if cache is not None:
            assert stage is None
            desc = cache.bind(self, b, h, w, batch_key, ctx_len=lt, row_ids=row_ids)
            rc = self._lib.mx_mmdit_forward_cached(self._handle, stream, latents.data_ptr(), code, ts.data_ptr(), ehs.data_ptr(),
                                                   pp.data_ptr(), out.data_ptr(), b, h, w, lt, ws.data_ptr(), ws.numel(), desc)
            if rc:
                err = cache.error
                cache.invalidate()            # see MxUNet.forward_one_cached: a forward that stopped part-way leaves nothing cached behind
                if err is not None:
                    raise err
            _lib.check(rc, "mx_mmdit_forward_cached")
            cache.after_forward()
            return out
        if stage is None:
            _lib.check(self._lib.mx_mmdit_forward(self._handle, stream, latents.data_ptr(), code, ts.data_ptr(), ehs.data_ptr(),
                                                  pp.data_ptr(), out.data_ptr(), b, h, w, lt, ws.data_ptr(),
                                                  ws.numel()), "mx_mmdit_forward")
            return out
        st = torch.empty(stage_shape, dtype=torch.bfloat16, device=self.device)
        _lib.check(self._lib.mx_mmdit_forward_trace(self._handle, stream, latents.data_ptr(), code, ts.data_ptr(), ehs.data_ptr(),
                                                    pp.data_ptr(), out.data_ptr(), b, h, w, lt, ws.data_ptr(),
                                                    ws.numel(), stage.encode(), st.data_ptr(), st.numel() * 2),
                   "mx_mmdit_forward_trace")
        return st

    def forward_mixed(self, latents, timestep: torch.Tensor, encoder_hidden_states: torch.Tensor, pooled: torch.Tensor):
        """ONE launch sequence over the latents of several resolutions (mx_mmdit_forward_mixed): ``latents[g]`` is [B_g, C, H_g, W_g]; the
        conditioning rows are those of all groups concatenated in list order (SD3Transformer.py:86 re-chunks all resolutions into one batch)."""
        assert 1 <= len(latents) <= _lib.MAX_SEGS
        latents = [x.contiguous() for x in latents]
        dt = latents[0].dtype
        assert all(x.is_cuda and x.ndim == 4 and x.dtype == dt for x in latents)
        btot = sum(x.shape[0] for x in latents)
        lt = encoder_hidden_states.shape[1]
        ts = timestep.to(device=self.device, dtype=torch.float32).reshape(-1)
        if ts.numel() == 1:
            ts = ts.expand(btot)
        ts = ts.contiguous()
        ehs = encoder_hidden_states.to(device=self.device, dtype=torch.bfloat16).contiguous()
        pp = pooled.to(device=self.device, dtype=torch.bfloat16).contiguous()
        assert ts.shape[0] == btot and ehs.shape[0] == btot and pp.shape == (btot, self.cfg.pooled_projection_dim)
        outs = [torch.empty((x.shape[0], self.cfg.out_channels, x.shape[2], x.shape[3]), dtype=dt, device=self.device) for x in latents]
        groups = (_lib.UNetGroup * len(latents))()
        for g, (x, o) in enumerate(zip(latents, outs)):
            groups[g].latents, groups[g].out = x.data_ptr(), o.data_ptr()
            groups[g].batch, groups[g].H, groups[g].W = x.shape[0], x.shape[2], x.shape[3]
        key = ("mixed", tuple((x.shape[0], x.shape[2], x.shape[3]) for x in latents), lt)
        need = self._ws_need.get(key)
        if need is None:
            need = self._ws_need[key] = self._lib.mx_mmdit_workspace_bytes_mixed(self._handle, groups, len(latents), lt)
        if need == 0:
            raise _lib.MxError("mx_mmdit_workspace_bytes_mixed: " + self._lib.mx_last_error().decode())
        stream = _lib.current_stream()
        skey = int(stream or 0)
        ws = self._ws_by_stream.get(skey)
        if ws is None or ws.numel() < need:
            self._ws_by_stream[skey] = None
            ws = self._ws_by_stream[skey] = torch.empty(need, dtype=torch.uint8, device=self.device)
        _lib.check(self._lib.mx_mmdit_forward_mixed(self._handle, stream, groups, len(latents), _lib.torch_dtype_code(dt), ts.data_ptr(), ehs.data_ptr(),
                                                    pp.data_ptr(), lt, ws.data_ptr(), ws.numel()), "mx_mmdit_forward_mixed")
        return outs

    def forward_mixed_cached(self, cache, latents, row_ids, timestep: torch.Tensor, encoder_hidden_states: torch.Tensor, pooled: torch.Tensor, patch: int):
        """forward_mixed through the block-skip cache at the reference's unit, the token chunk (block_cache.PatchSkipCache with mmdit_ctx_len;
        mx_mmdit_forward_cached_mixed): ONE launch sequence, one host decision per joint block for the chunks of every resolution.  ``patch``: the
        patch edge in latent pixels (patch_size / 8 of the reference's call)."""
        assert 1 <= len(latents) <= _lib.MAX_SEGS and patch > 0
        latents = [x.contiguous() for x in latents]
        dt = latents[0].dtype
        btot = sum(x.shape[0] for x in latents)
        lt = encoder_hidden_states.shape[1]
        ts = timestep.to(device=self.device, dtype=torch.float32).reshape(-1)
        ts = (ts.expand(btot) if ts.numel() == 1 else ts).contiguous()
        ehs = encoder_hidden_states.to(device=self.device, dtype=torch.bfloat16).contiguous()
        pp = pooled.to(device=self.device, dtype=torch.bfloat16).contiguous()
        assert ts.shape[0] == btot and ehs.shape[0] == btot and pp.shape == (btot, self.cfg.pooled_projection_dim)
        outs = [torch.empty((x.shape[0], self.cfg.out_channels, x.shape[2], x.shape[3]), dtype=dt, device=self.device) for x in latents]
        groups = (_lib.UNetGroup * len(latents))()
        for g, (x, o) in enumerate(zip(latents, outs)):
            groups[g].latents, groups[g].out = x.data_ptr(), o.data_ptr()
            groups[g].batch, groups[g].H, groups[g].W = x.shape[0], x.shape[2], x.shape[3]
        shapes = tuple((x.shape[0], x.shape[2], x.shape[3]) for x in latents)
        key = ("mixed_cached", shapes, lt, patch)
        need = self._ws_need.get(key)
        if need is None:
            need = self._ws_need[key] = self._lib.mx_mmdit_workspace_bytes_cached_mixed(self._handle, groups, len(latents), lt, patch)
        if need == 0:
            raise _lib.MxError("mx_mmdit_workspace_bytes_cached_mixed: " + self._lib.mx_last_error().decode())
        stream = _lib.current_stream()
        skey = int(stream or 0)
        ws = self._ws_by_stream.get(skey)
        if ws is None or ws.numel() < need:
            self._ws_by_stream[skey] = None
            ws = self._ws_by_stream[skey] = torch.empty(need, dtype=torch.uint8, device=self.device)
        assert cache.mmdit_ctx_len == lt
        desc = cache.bind(self, shapes, row_ids, patch)
        rc = self._lib.mx_mmdit_forward_cached_mixed(self._handle, stream, groups, len(latents), _lib.torch_dtype_code(dt), ts.data_ptr(), ehs.data_ptr(),
                                                     pp.data_ptr(), lt, patch, ws.data_ptr(), ws.numel(), desc)
        if rc:
            err = cache.error
            cache.invalidate()
            if err is not None:
                raise err
        _lib.check(rc, "mx_mmdit_forward_cached_mixed")
        cache.after_forward()
        return outs

    def forward(self, hidden_states: Dict[str, torch.Tensor], encoder_hidden_states: torch.Tensor = None,
                pooled_projections: torch.Tensor = None, timestep: torch.Tensor = None, block_controlnet_hidden_states=None,
                joint_attention_kwargs=None, return_dict: bool = True, skip_layers=None, patch_size: int = None,
                is_sliced: bool = False, save_index: int = 0, input_indices: dict = None):
        assert block_controlnet_hidden_states is None and skip_layers is None and not joint_attention_kwargs
        out: Dict[str, torch.Tensor] = {}
        row = 0
        keys = [k for k in hidden_states if hidden_states[k] is not None and hidden_states[k].shape[0] > 0]
        if not is_sliced:
            keys = keys[:1]   # the reference's unsliced branch runs the first resolution only (SD3Transformer.py:105-109)
        if is_sliced and getattr(self, "_block_caches", None) is not None and len(keys) <= _lib.MAX_SEGS and patch_size is not None \
                and all(int(k) % patch_size == 0 and int(k) > patch_size for k in keys):
            # ESYMRED_USE_CACHE=TRUE with is_sliced=True: the cache at its reference unit, the token chunk; every resolution in ONE launch sequence
            ids = input_indices or {}
            assert all(k in ids and len(ids[k]) > 0 and hidden_states[k].shape[0] % len(ids[k]) == 0 for k in keys), \
                "the block-skip cache keys its state by input_indices[resolution] (cache_manager.py:166)"
            row_ids = [r for k in keys for r in _row_ids(ids[k], hidden_states[k].shape[0])]
            lt = encoder_hidden_states.shape[1]
            if self._patch_cache is None or self._patch_cache.mmdit_ctx_len != lt:
                self._patch_cache = self._new_patch_cache(lt)
            res = self.forward_mixed_cached(self._patch_cache, [hidden_states[k] for k in keys], row_ids, timestep, encoder_hidden_states, pooled_projections,
                                            patch_size // 8)
            return (dict(zip(keys, res)),)
        if is_sliced and 1 < len(keys) <= _lib.MAX_SEGS and getattr(self, "_block_caches", None) is None and self.mixed_one_sequence:
            res = self.forward_mixed([hidden_states[k] for k in keys], timestep, encoder_hidden_states, pooled_projections)
            return (dict(zip(keys, res)),)
        for key in keys:
            x = hidden_states[key]
            n = x.shape[0]
            sl = slice(row, row + n)
            ts = timestep if timestep.ndim == 0 else timestep[sl]
            caches = getattr(self, "_block_caches", None)
            if caches is not None:                     # ESYMRED_USE_CACHE=TRUE (enable_block_cache)
                ids = (input_indices or {}).get(key)
                assert ids is not None and len(ids) > 0 and n % len(ids) == 0, "the block-skip cache keys its state by input_indices[resolution] (cache_manager.py:166)"
                bc = caches.get(key)
                if bc is None:
                    bc = caches[key] = self._new_block_cache()
                out[key] = self.forward_one(x, ts, encoder_hidden_states[sl], pooled_projections[sl], cache=bc, row_ids=_row_ids(ids, n))
            else:
                out[key] = self.forward_one(x, ts, encoder_hidden_states[sl], pooled_projections[sl])
            row += n
        return (out,)

    def enable_block_cache(self, predictor, forced_after: Optional[int] = None, observe: bool = False) -> None:
        """Route forward() through the block-skip cache, one state per resolution key (SD3Transformer.py:151-228 with
        ESYMRED_USE_CACHE=TRUE).  `predictor`: an object with .predict(features) (block_cache.py)."""
        from .block_cache import BlockSkipCache, FORCED_RUN_AFTER_SD3, PatchSkipCache
        fa = FORCED_RUN_AFTER_SD3 if forced_after is None else forced_after
        self._new_block_cache = lambda: BlockSkipCache(predictor, forced_after=fa, observe=observe)
        ml = min(128, self.cfg.pos_embed_max_size * self.cfg.patch_size)                                    # state rows: up to 1024 px (or the positional table)
        self._new_patch_cache = lambda lt: PatchSkipCache(predictor, forced_after=fa, mmdit_ctx_len=lt, max_latent=ml)   # is_sliced=True: the chunk unit
        self._block_caches = {}
        self._patch_cache = None

    def disable_block_cache(self) -> None:
        self._block_caches = None
        self._patch_cache = None

    __call__ = forward
