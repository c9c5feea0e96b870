"""Block-skip cache: the host half of ``mx_unet_forward_cached`` (include/mxdenoise.h).

The reference wraps each of the SDXL UNet's seven blocks with a ``CacheManager`` (modules/cache_manager.py:21-161): the block's
input is compared with the input it saw at its last run, a predictor sees ``[block index, timestep, mse (, mse of each skip)]``
per sample and answers run / reuse, and a sample that reused a block four times in a row is forced to run it
(cache_manager.py:134,154).  The comparison, the copies and the reuse live in the library; this module owns the decision:

  * ``BlockSkipCache`` keeps the device state, the per-sample reuse counters and the ctypes callback;
  * the predictors the reference loads are cuML random forests pickled with joblib (ESYMRED_UPSAMPLE_PATH /
    ESYMRED_DOWNSAMPLE_PATH) and cannot be loaded without cuML -- any object with ``.predict(features) -> 0/1 per row`` is
    accepted, and ``ThresholdPredictor`` is the rule shipped here.

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

    def __init__(self, down, up=None, forced_after: int = FORCED_RUN_AFTER):
        self.down, self.up = down, (up if up is not None else down)
        self.forced_after = forced_after
        self.state: Optional[torch.Tensor] = None
        self.desc = _lib.BlockCacheC()
        self._cb = _lib.SKIP_PREDICT_FN(self._predict)       # kept alive with the object
        self.desc.predict = self._cb
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
            prev = self.previous.get(block)
            if prev is None or prev.shape[0] != n or bool((m[:, 0] >= MSE_UNCACHED * 0.5).any()):
                prev = np.zeros(n, dtype=np.int64)           # "0 if not in the cache" (cache_manager.py:128,150)
            raw = np.asarray((self.up if is_up else self.down).predict(feats))
            run, self.previous[block] = decide(raw, prev, self.forced_after)
            self.decisions.append((block, run.copy()))
            for i in range(n):
                run_out[i] = 1 if run[i] else 0
            return 0
        except BaseException as e:                           # never unwind through the C frame
            self.error = e
            return 1

    def bind(self, model, batch: int, h: int, w: int, batch_key: int, ctx_len: Optional[int] = None):
        """size / (re)allocate the device state for `model` (MxUNet, or MxMMDiT when ctx_len is given) and return the descriptor"""
        unet = model
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

    def after_forward(self):
        self.history.append(int(self.desc.blocks_run) | int(self.desc.blocks_run_hi) << 32)

    def invalidate(self):
        self.desc.cached_valid = 0
        self.previous = {}

    @staticmethod
    def blocks_of(mask: int) -> Sequence[int]:
        return [i for i in range(64) if mask >> i & 1]
