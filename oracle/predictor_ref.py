"""Restatement of the feature construction of the reference's latency predictor (TEST INFRASTRUCTURE, see oracle/__init__.py).

``Predictor.predict`` (sduss/worker/scheduler/policy/ESyMReD.py:46-53): for a batch composition (n512, n768, n1024) the
5-feature row is [n512, n768, n1024, 4 n512 + 9 n768 + 16 n1024, #non-zero counts]; the sklearn model's output is the
50-step batch time in seconds and is divided by 50 to give seconds per step.
Pinned against the reference class itself: tests/golden/ref_predictor.npz (tests/golden/make_ref_fixtures.py).
"""
import numpy as np


def predictor_features(task_distribute) -> np.ndarray:
    t = np.asarray(task_distribute, dtype=np.float64)
    weighted = t[:, :1] * 4 + t[:, 1:2] * 9 + t[:, 2:3] * 16
    nonzero = np.count_nonzero(t, axis=1)[:, None]
    return np.concatenate([t, weighted, nonzero], axis=1)


def predict_step_seconds(model, task_distribute, steps: int = 50) -> np.ndarray:
    return model.predict(predictor_features(task_distribute)) / steps
