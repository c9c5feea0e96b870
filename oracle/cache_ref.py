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
