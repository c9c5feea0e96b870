"""Drop-in for the reference's JIT-built pybind module ``esymred_mp``
(sduss/model_executor/modules/groupnorm.py:17-27; entry points norm_silu_concat.cpp:66-106), same names and
argument order, backed by mx_groupnorm_halo / mx_halo_only of libmxdenoise.so.

    groupnorm(X, gamma, beta, N, C, H, W, group, eps, padding, latent_offset, patch_map, padding_idx) -> Y
    mock_groupnorm(X, N, C, H, W, group, padding_idx) -> Y

``group`` is channels-per-group (groupnorm.py:50,58 pass C / num_groups).  Raises ``MxError`` on a bad launch
instead of printing and continuing (norm_silu_concat.cu:434-437).
"""
from __future__ import annotations

from typing import Optional

import torch

from . import lib as _lib


def _i32(t: torch.Tensor, device) -> torch.Tensor:
    return t.to(device=device, dtype=torch.int32).contiguous()


def groupnorm(X: torch.Tensor, gamma: Optional[torch.Tensor], beta: Optional[torch.Tensor], N: int, C: int, H: int, W: int,
              group: int, eps: float, padding: bool, latent_offset: torch.Tensor, patch_map: torch.Tensor,
              padding_idx: Optional[torch.Tensor]) -> torch.Tensor:
    l = _lib.load()
    assert X.is_cuda and X.is_contiguous() and tuple(X.shape) == (N, C, H, W)
    dev = X.device
    p = 1 if padding else 0
    Y = torch.empty((N, C, H + 2 * p, W + 2 * p), dtype=X.dtype, device=dev)
    lo = _i32(latent_offset, dev)
    pm = _i32(patch_map, dev)
    pi = _i32(padding_idx, dev) if padding_idx is not None else None
    g = gamma.to(device=dev, dtype=X.dtype).contiguous() if gamma is not None else None
    b = beta.to(device=dev, dtype=X.dtype).contiguous() if beta is not None else None
    ws = torch.empty(l.mx_groupnorm_halo_workspace_bytes(N, C, group), dtype=torch.uint8, device=dev)
    _lib.check(l.mx_groupnorm_halo(_lib.current_stream(), X.data_ptr(), g.data_ptr() if g is not None else None,
                                   b.data_ptr() if b is not None else None, Y.data_ptr(), N, C, H, W, group, float(eps),
                                   int(bool(padding)), lo.data_ptr(), lo.numel() - 1, pm.data_ptr(),
                                   pi.data_ptr() if pi is not None else None, _lib.torch_dtype_code(X.dtype), ws.data_ptr()),
               "mx_groupnorm_halo")
    return Y


def mock_groupnorm(X: torch.Tensor, N: int, C: int, H: int, W: int, group: int, padding_idx: torch.Tensor) -> torch.Tensor:
    l = _lib.load()
    assert X.is_cuda and X.is_contiguous() and tuple(X.shape) == (N, C, H, W)
    Y = torch.empty((N, C, H + 2, W + 2), dtype=X.dtype, device=X.device)
    pi = _i32(padding_idx, X.device)
    _lib.check(l.mx_halo_only(_lib.current_stream(), X.data_ptr(), Y.data_ptr(), N, C, H, W, pi.data_ptr(),
                              _lib.torch_dtype_code(X.dtype)), "mx_halo_only")
    return Y


def get_adjacency(input: torch.Tensor, padding_idx: torch.Tensor = None) -> torch.Tensor:
    """groupnorm.py:31-33."""
    N, C, H, W = input.shape
    return mock_groupnorm(input, N, C, H, W, int(C / 32), padding_idx)
