"""Block-skip cache: the host half of ``mx_unet_forward_cached`` (include/mxdenoise.h).

The reference wraps each of the SDXL UNet's seven blocks with a ``CacheManager`` (modules/cache_manager.py:21-161): the block's
input is compared with the input it saw at its last run, a predictor sees ``[block index, timestep, mse (, mse of each skip)]``
per sample and answers run / reuse, and a sample that reused a block four times in a row is forced to run it
(cache_manager.py:134,154).  The comparison, the copies and the reuse live in the library; this module owns the decision:

  * ``BlockSkipCache`` keeps the device state, the per-sample reuse counters and the ctypes callback;
  * the predictors the reference loads are cuML random forests pickled with joblib (ESYMRED_UPSAMPLE_PATH /
    ESYMRED_DOWNSAMPLE_PATH) and cannot be loaded without cuML -- any object with ``.predict(features) -> 0/1 per row`` is
    accepted; tools/fit_skip_predictor.py re-fits scikit-learn forests of the same kind from traces of this denoiser,
    ``CompiledForest`` evaluates them natively (mx_forest_predict), and ``ThresholdPredictor`` is the rule shipped without any fitting.

Granularity is the step batch (see the header): the library reuses a block only when no sample asks to run it.
"""
import ctypes as C
from typing import Optional, Sequence

import numpy as np
import torch

from . import lib as _lib

MSE_UNCACHED = float(np.float32(_lib.MSE_UNCACHED))
FORCED_RUN_AFTER = 4          # cache_manager.py:134,154 (SDXL)
FORCED_RUN_AFTER_SD3 = 2      # cache_manager.py:184-186


class ThresholdPredictor:
    """run (1) when any input difference of the row exceeds ``threshold``; rows are the reference's feature rows
    ``[block, timestep, mse, (mse of each skip ...)]``."""

    def __init__(self, threshold: float):
        self.threshold = float(threshold)

    def predict(self, features: np.ndarray) -> np.ndarray:
        features = np.asarray(features, dtype=np.float64)
        return (features[:, 2:].max(axis=1) > self.threshold).astype(np.int64)


class QuantilePredictor:
    """run the rows whose largest input difference lies above the ``q`` quantile of the call's cached rows (uncached rows run): a fixed asking
    fraction of about 1 - q whatever the scale of the differences -- for timing the mechanism on random-init weights, where no fitted threshold
    means anything (bench.py's cached mixed-stream leg, tools/block_cache_bench.py)."""

    def __init__(self, q: float):
        self.q = float(q)

    def predict(self, features: np.ndarray) -> np.ndarray:
        f = np.asarray(features, dtype=np.float64)
        m = f[:, 2:].max(axis=1)
        unc = m >= MSE_UNCACHED * 0.5
        out = np.ones(len(f), dtype=np.int64)
        if (~unc).any():
            thr = np.quantile(m[~unc], self.q)
            out[~unc] = (m[~unc] > thr).astype(np.int64)
        return out


class CompiledForest:
    """A fitted scikit-learn RandomForestClassifier (binary) flattened for mx_forest_predict: same answers as ``forest.predict`` without
    its per-call overhead (0.4 ms per call through joblib / validation -- 24 calls per SD3 step cost more than the blocks they save)."""

    def __init__(self, forest):
        classes = list(getattr(forest, "classes_", [0, 1]))
        left, right, feat, thr, p1, roots = [], [], [], [], [], []
        base = 0
        for est in forest.estimators_:
            t = est.tree_
            n = t.node_count
            roots.append(base)
            cl, cr = t.children_left.astype(np.int64), t.children_right.astype(np.int64)
            left.append(np.where(cl >= 0, cl + base, -1)); right.append(np.where(cr >= 0, cr + base, -1))
            feat.append(t.feature.astype(np.int64)); thr.append(t.threshold.astype(np.float64))
            v = t.value[:, 0, :].astype(np.float64)                 # [nodes, classes]: counts or fractions, normalised per node below
            tot = v.sum(axis=1)
            one = v[:, classes.index(1)] if 1 in classes else np.zeros(n)
            p1.append(np.where(tot > 0, one / np.where(tot > 0, tot, 1.0), 0.0))
            base += n
        as32 = lambda a: np.ascontiguousarray(np.concatenate(a), dtype=np.int32)
        self.left, self.right, self.feature = as32(left), as32(right), as32(feat)
        self.threshold = np.ascontiguousarray(np.concatenate(thr), dtype=np.float64)
        self.p1 = np.ascontiguousarray(np.concatenate(p1), dtype=np.float64)
        self.roots = np.ascontiguousarray(roots, dtype=np.int32)
        self.n_features = int(forest.n_features_in_)
        self._lib = _lib.load()

    def predict(self, features: np.ndarray) -> np.ndarray:
        X = np.ascontiguousarray(features, dtype=np.float32)
        assert X.ndim == 2 and X.shape[1] == self.n_features
        out = np.zeros(X.shape[0], dtype=np.uint8)
        _lib.check(self._lib.mx_forest_predict(self.left.ctypes.data, self.right.ctypes.data, self.feature.ctypes.data, self.threshold.ctypes.data,
                                               self.p1.ctypes.data, self.roots.ctypes.data, len(self.roots), X.ctypes.data, X.shape[0], X.shape[1],
                                               out.ctypes.data), "mx_forest_predict")
        return out.astype(np.int64)


def decide(mask: np.ndarray, previous: np.ndarray, forced_after: int = FORCED_RUN_AFTER):
    """The reference's post-processing of a predictor's answer (cache_manager.py:134-136 = 154-156): a sample whose counter reached
    ``forced_after`` runs; the counter resets on a run and counts reuses otherwise.  Returns (run mask, new counters)."""
    mask = np.asarray(mask).astype(np.int64).copy()
    previous = np.asarray(previous, dtype=np.int64)
    forced = previous == forced_after
    mask[forced] = 1
    new_prev = np.where((mask == 1) | forced, 0, previous + 1)
    return mask > 0, new_prev


class BlockSkipCache:
    """State for one stream of steps over one batch composition.  ``down`` decides the down and mid blocks, ``up`` the up blocks
    (downsample_predictor / upsample_predictor of the reference)."""

    def __init__(self, down, up=None, forced_after: int = FORCED_RUN_AFTER, observe: bool = False):
        """observe=True records, for every block that ran, how far its output moved since its last run (``self.observed``: (block,
        mse per sample)) next to the feature rows the predictor saw (``self.features``) -- what tools/fit_skip_predictor.py fits on."""
        # fitted scikit-learn forests are flattened once and evaluated natively (CompiledForest: same answers, ~100x less per call)
        wrap = lambda p: CompiledForest(p) if hasattr(p, "estimators_") and hasattr(p, "n_features_in_") else p
        self.down, self.up = wrap(down), wrap(up if up is not None else down)
        self.forced_after = forced_after
        self.state: Optional[torch.Tensor] = None
        self.desc = _lib.BlockCacheC()
        self._cb = _lib.SKIP_PREDICT_FN(self._predict)       # kept alive with the object
        self.desc.predict = self._cb
        self.observed, self.features = [], []
        self._row_ids, self._geom, self._slot_of, self._cap = None, None, {}, 0
        self._pending = None                                 # request -> row table of a forward in flight (committed by after_forward)
        self._record = observe
        if observe:
            self._cb_obs = _lib.SKIP_OBSERVE_FN(self._observe)
            self.desc.observe = self._cb_obs
        self.previous = {}                                   # block -> int64[batch]
        self.decisions = []                                  # (block, run mask) of the last forward
        self.history = []                                    # blocks_run bit mask per forward
        self.error: Optional[BaseException] = None

    # called from the library, once per block, on the thread that called mx_unet_forward_cached
    def _predict(self, _ctx, block, is_up, n, nf, timesteps, mse, run_out):
        try:
            ts = np.ctypeslib.as_array(timesteps, shape=(n,)).astype(np.float64)
            m = np.ctypeslib.as_array(mse, shape=(n, nf)).astype(np.float64)
            feats = np.concatenate([np.full((n, 1), float(block)), ts[:, None], m], axis=1)
            if self._record:
                self.features.append(feats.copy())
            ids = getattr(self, "_row_ids", None)
            uncached = m[:, 0] >= MSE_UNCACHED * 0.5
            if ids is not None:                              # per request: "0 if not in the cache else previous" (cache_manager.py:128,150)
                counts = self.previous.get(block, {})
                prev = np.array([0 if uncached[i] else counts.get(ids[i], 0) for i in range(n)], dtype=np.int64)
            else:
                prev = self.previous.get(block)
                if prev is None or isinstance(prev, dict) or prev.shape[0] != n or bool(uncached.any()):
                    prev = np.zeros(n, dtype=np.int64)
            raw = np.asarray((self.up if is_up else self.down).predict(feats))
            run, new_prev = decide(raw, prev, self.forced_after)
            self.previous[block] = {ids[i]: int(new_prev[i]) for i in range(n)} if ids is not None else new_prev
            self.decisions.append((block, run.copy()))
            for i in range(n):
                run_out[i] = 1 if run[i] else 0
            return 0
        except BaseException as e:                           # never unwind through the C frame
            self.error = e
            return 1

    def _observe(self, _ctx, block, n, out_mse):
        self.observed.append((int(block), np.ctypeslib.as_array(out_mse, shape=(n,)).astype(np.float64).copy()))

    def bind(self, model, batch: int, h: int, w: int, batch_key: int, ctx_len: Optional[int] = None, row_ids: Optional[Sequence] = None):
        """size / (re)allocate the device state for `model` (MxUNet, or MxMMDiT when ctx_len is given) and return the descriptor.
        row_ids (one hashable id per sample, e.g. "<request id>#<cfg half>"): the state is kept per REQUEST, as the reference's dictionaries
        keyed by request id do (cache_manager.py:105-133) -- a request that stays while the batch around it changes keeps its cached tensors and
        its reuse counters, one that left is forgotten, a new one makes the blocks run once.  Without row_ids the state belongs to the batch
        composition named by batch_key."""
        unet = model
        if row_ids is not None:
            return self._bind_rows(model, batch, h, w, ctx_len, list(row_ids))
        self._row_ids = None
        self.desc.slots = None; self.desc.slot_valid = None; self.desc.n_slots = 0
        if ctx_len is None:
            need = model._lib.mx_unet_block_cache_bytes(model._handle, batch, h, w)
        else:
            need = model._lib.mx_mmdit_block_cache_bytes(model._handle, batch, h, w, ctx_len)
        if need == 0:
            raise _lib.MxError("block_cache_bytes: " + model._lib.mx_last_error().decode())
        if self.state is None or self.state.numel() < need:
            self.state = torch.empty(need, dtype=torch.uint8, device=unet.device)
            self.desc.cached_valid = 0
        if not (self.desc.cached_valid and self.desc.cached_key == (batch_key & (2 ** 64 - 1)) and self.desc.cached_batch == batch):
            self.previous = {}
        self.desc.state = self.state.data_ptr()
        self.desc.state_bytes = self.state.numel()
        self.desc.batch_key = batch_key & (2 ** 64 - 1)
        self.decisions = []
        self.error = None
        return C.byref(self.desc)

    def _bind_rows(self, model, batch, h, w, ctx_len, row_ids):
        assert len(row_ids) == batch and len(set(row_ids)) == batch, "one distinct id per sample"
        if getattr(self, "_pending", None) is not None:        # the previous forward never reached after_forward(): it failed part-way, so
            self.invalidate()                                  # some blocks' rows were stored and others not -- nothing cached can be trusted
        geom = (h, w, ctx_len)
        if getattr(self, "_geom", None) != geom:              # another latent size: nothing cached applies
            self._geom, self._slot_of, self._cap = geom, {}, 0
            self.previous = {}
        if batch > self._cap:                                  # grow the state (rarely: sized for twice the largest batch seen)
            self._cap = max(2 * batch, 8)
            if ctx_len is None:
                need = model._lib.mx_unet_block_cache_bytes(model._handle, self._cap, h, w)
            else:
                need = model._lib.mx_mmdit_block_cache_bytes(model._handle, self._cap, h, w, ctx_len)
            if need == 0:
                raise _lib.MxError("block_cache_bytes: " + model._lib.mx_last_error().decode())
            self.state = torch.empty(need, dtype=torch.uint8, device=model.device)
            self._slot_of = {}
            self.previous = {}
        # "self.cache = {only the ids of this call}" (cache_manager.py:131,153): ids that left are forgotten, their rows are free again
        keep = {k: v for k, v in self._slot_of.items() if k in set(row_ids)}
        free = sorted(set(range(self._cap)) - set(keep.values()))
        valid = []
        for rid in row_ids:
            if rid in keep:
                valid.append(1)
            else:
                keep[rid] = free.pop(0)
                valid.append(0)
        # The new table is only STAGED here and committed by after_forward(): a forward that fails part-way (predictor exception, short
        # workspace) has not stored the new requests' rows for its later blocks, and must not leave them marked valid for the next step.
        self._pending = keep
        self._row_ids = row_ids
        self._slots_arr = (C.c_int32 * batch)(*[keep[r] for r in row_ids])
        self._valid_arr = (C.c_ubyte * batch)(*valid)
        for blk, counts in list(self.previous.items()):        # counters follow the requests
            self.previous[blk] = {k: v for k, v in counts.items() if k in keep}
        self.desc.state = self.state.data_ptr()
        self.desc.state_bytes = self.state.numel()
        self.desc.slots = C.cast(self._slots_arr, C.POINTER(C.c_int32))
        self.desc.slot_valid = C.cast(self._valid_arr, C.POINTER(C.c_ubyte))
        self.desc.n_slots = self._cap
        self.decisions = []
        self.error = None
        return C.byref(self.desc)

    def after_forward(self):
        """the forward succeeded: commit the request -> row table staged by bind()"""
        if getattr(self, "_pending", None) is not None:
            self._slot_of, self._pending = self._pending, None
        self.history.append(int(self.desc.blocks_run) | int(self.desc.blocks_run_hi) << 32)

    def invalidate(self):
        """forget everything cached (called when a forward failed, and by callers that change the model's inputs out of band)"""
        self.desc.cached_valid = 0
        self.previous = {}
        self._slot_of = {}
        self._pending = None

    @staticmethod
    def blocks_of(mask: int) -> Sequence[int]:
        return [i for i in range(64) if mask >> i & 1]


class PatchSkipCache:
    """The cache at the reference's own unit, the 256-px PATCH, over a mixed-resolution batch in ONE launch sequence: host half of
    ``mx_unet_forward_cached_mixed`` (include/mxdenoise.h).  What the reference does with ESYMRED_USE_CACHE=TRUE and is_sliced=True -- the only mode
    in which its cache works, and the one its mixed policies force: every CacheManager dictionary is keyed ``"<request id>-<h>-<w>"``
    (modules/utils.py:37,60), get_mask decides per patch (cache_manager.py:101-161) and a patch that reused a block four times in a row is
    forced to run it.  The library compares, gathers the asking patches, computes them and merges; this class owns the state tensor (one row per
    sample id, i.e. per request and CFG half), the per-patch reuse counters and the predictor callback (one call per block for the patches of
    all samples of all resolutions)."""

    def __init__(self, down, up=None, forced_after: int = FORCED_RUN_AFTER, max_latent: int = 128, mmdit_ctx_len: Optional[int] = None):
        """mmdit_ctx_len: set for the SD3 / SD3.5 transformer (mx_mmdit_forward_cached_mixed): the unit is then the token chunk
        "<request id>-<k>" (modules/utils.py:86-122), one cache point per joint block, forced run after two reuses (pass forced_after=2)."""
        self.mmdit_ctx_len = mmdit_ctx_len
        wrap = lambda p: CompiledForest(p) if hasattr(p, "estimators_") and hasattr(p, "n_features_in_") else p
        self.down, self.up = wrap(down), wrap(up if up is not None else down)
        self.forced_after = forced_after
        self.max_latent = max_latent
        self.state: Optional[torch.Tensor] = None
        self.desc = _lib.BlockCacheC()
        self._cb = _lib.SKIP_PREDICT_FN(self._predict)
        self.desc.predict = self._cb
        self._slot_of, self._cap, self._patch, self._pending = {}, 0, None, None
        self._keys = []                                      # patch keys of the forward in flight, in the library's row order
        self._prev = {}                                      # block -> int64[len(_keys)]: consecutive reuses per patch, aligned with _keys (remapped by bind())
        self.decisions, self.features, self.history = [], [], []
        self.record_features = False
        self.error: Optional[BaseException] = None
        self.patches_asked = self.patches_total = 0

    @property
    def previous(self):
        """block -> {patch key: consecutive reuses} (the reference's per-key dictionaries, cache_manager.py:128,150); kept as arrays aligned with the forward's
        patch order, because this callback runs seven times per forward on the critical path (the GPU idles while it decides)"""
        return {blk: {k: int(v) for k, v in zip(self._keys, arr)} for blk, arr in self._prev.items()}

    @previous.setter
    def previous(self, value):
        pos = {k: i for i, k in enumerate(self._keys)}
        self._prev = {}
        for blk, counts in value.items():
            arr = np.zeros(len(self._keys), dtype=np.int64)
            for k, v in counts.items():
                if k in pos:
                    arr[pos[k]] = v
            self._prev[blk] = arr

    def _predict(self, _ctx, block, is_up, n, nf, timesteps, mse, run_out):
        try:
            ts = np.ctypeslib.as_array(timesteps, shape=(n,))
            m = np.ctypeslib.as_array(mse, shape=(n, nf))
            feats = np.empty((n, 2 + nf), dtype=np.float64)
            feats[:, 0] = float(block); feats[:, 1] = ts; feats[:, 2:] = m
            if self.record_features:
                self.features.append(feats.copy())
            assert len(self._keys) == n
            uncached = feats[:, 2] >= MSE_UNCACHED * 0.5
            prev = self._prev.get(block)
            if prev is None or len(prev) != n:
                prev = np.zeros(n, dtype=np.int64)
            prev = np.where(uncached, 0, prev)               # "0 if not in the cache else previous" (cache_manager.py:128,150)
            raw = np.asarray((self.up if is_up else self.down).predict(feats))
            run, new_prev = decide(raw, prev, self.forced_after)
            run = run | uncached                              # nothing cached: nothing to reuse
            self._prev[block] = np.where(uncached, 0, new_prev)
            self.decisions.append((int(block), run.copy()))
            np.ctypeslib.as_array(run_out, shape=(n,))[:] = run
            return 0
        except BaseException as e:                           # never unwind through the C frame
            self.error = e
            return 1

    def bind(self, model, shapes: Sequence, row_ids: Sequence, gn_patch: int):
        """shapes: per group (batch, H, W) in the library's group order; row_ids: one hashable id per sample in row order (all groups).  Returns
        the descriptor for mx_unet_forward_cached_mixed; commits the request -> row table in after_forward()."""
        row_ids = list(row_ids)
        n = sum(b for b, _h, _w in shapes)
        assert len(row_ids) == n and len(set(row_ids)) == n, "one distinct id per sample"
        if self._pending is not None:                         # the previous forward failed part-way
            self.invalidate()
        if self._patch != gn_patch:
            self._patch, self._slot_of, self._cap, self.previous = gn_patch, {}, 0, {}
        ml = max([self.max_latent] + [max(h, w) for _b, h, w in shapes])
        ml = (ml + gn_patch - 1) // gn_patch * gn_patch
        if n > self._cap or ml != self.max_latent or self.state is None:
            self.max_latent = ml
            self._cap = max(2 * n, 8)
            if self.mmdit_ctx_len is None:
                need = model._lib.mx_unet_patch_cache_bytes(model._handle, self._cap, ml, ml, gn_patch)
            else:
                need = model._lib.mx_mmdit_patch_cache_bytes(model._handle, self._cap, ml, ml, gn_patch, self.mmdit_ctx_len)
            if need == 0:
                raise _lib.MxError("patch_cache_bytes: " + model._lib.mx_last_error().decode())
            self.state = None
            self.state = torch.empty(need, dtype=torch.uint8, device=model.device)
            self._slot_of, self.previous = {}, {}
        keep = {k: v for k, v in self._slot_of.items() if k in set(row_ids)}       # ids that left are forgotten (cache_manager.py:131,153)
        free = sorted(set(range(self._cap)) - set(keep.values()))
        valid = []
        for rid in row_ids:
            if rid in keep:
                valid.append(1)
            else:
                keep[rid] = free.pop(0)
                valid.append(0)
        self._pending = keep
        keys, i = [], 0
        for b, h, w in shapes:                                # the library's patch order: group, sample, patch row, patch column
            for _ in range(b):
                if self.mmdit_ctx_len is None:
                    keys += [f"{row_ids[i]}-{py}-{px}" for py in range(h // gn_patch) for px in range(w // gn_patch)]
                else:
                    keys += [f"{row_ids[i]}-{k}" for k in range((h // gn_patch) * (w // gn_patch))]
                i += 1
        if keys != self._keys:                                 # a new composition: the counters follow their patches, patches that left are forgotten
            old_pos = {k: i for i, k in enumerate(self._keys)}
            idx = np.array([old_pos.get(k, -1) for k in keys], dtype=np.int64)
            for blk, arr in list(self._prev.items()):
                self._prev[blk] = np.where(idx >= 0, arr[np.maximum(idx, 0)] if len(arr) else 0, 0).astype(np.int64)
            self._keys = keys
        self._slots_arr = (C.c_int32 * n)(*[keep[r] for r in row_ids])
        self._valid_arr = (C.c_ubyte * n)(*valid)
        d = self.desc
        d.state, d.state_bytes = self.state.data_ptr(), self.state.numel()
        d.slots = C.cast(self._slots_arr, C.POINTER(C.c_int32))
        d.slot_valid = C.cast(self._valid_arr, C.POINTER(C.c_ubyte))
        d.n_slots, d.max_h, d.max_w = self._cap, self.max_latent, self.max_latent
        self.decisions, self.error = [], None
        return C.byref(self.desc)

    def after_forward(self):
        if self._pending is not None:
            self._slot_of, self._pending = self._pending, None
        self.history.append(int(self.desc.blocks_run) | int(self.desc.blocks_run_hi) << 32)
        self.patches_asked += int(self.desc.patches_asked)
        self.patches_total += int(self.desc.patches_total)

    def invalidate(self):
        self._slot_of, self._pending, self.previous = {}, None, {}
