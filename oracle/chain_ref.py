"""CPU restatement of the denoising LOOP: ``denoising_step`` called once per scheduled batch per timestep on a batch whose
requests sit at different step indices and carry different step counts (continuous batching, BASELINE configs[4]).

TEST INFRASTRUCTURE (see oracle/__init__.py).  PARITY UNPINNED.

Follows, per step and per resolution:
* gather of the per-request latents / embeddings / timesteps -- pipelines/stable_diffusion_xl/pipeline_stable_diffusion_xl_esymred.py:287-316
  (SD3: pipelines/stable_diffusion_3/pipeline_stable_diffusion_3_esymred.py:240-303);
* CFG duplication, rows [uncond..., cond...] -- :322-339 (SD3 :281-292);
* per-request sigma vectors built from ``sigmas[_step_index]`` of every request -- schedulers/scheduling_euler_discrete.py:171-175, 213-217
  (flow match: scheduling_flow_match_euler_discrete.py:171-183): a request that joined later, or has another ``num_inference_steps``, simply
  carries its own table and index;
* model call -> CFG combine -> scheduler step -> write back and ``_step_index += 1`` per request -- :369-403 (SD3 :312-388).
Requests keep their latents in the model dtype between steps (runner/wrappers.py:19-36), so the chain rounds to ``store_dtype`` after every step.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Dict, List, Optional

import numpy as np
import torch

from . import scheduler_ref


def sd3_flow_tables(num_inference_steps: int, num_train_timesteps: int = 1000, shift: float = 3.0):
    """diffusers==0.32.1 FlowMatchEulerDiscreteScheduler.set_timesteps with the SD3.5 scheduler config (shift 3.0, no dynamic shifting), as
    ``batch_set_timesteps`` stores it per request (scheduling_flow_match_euler_discrete.py:69-131).  Returns (timesteps[n], sigmas[n + 1])."""
    base = np.linspace(1, num_train_timesteps, num_train_timesteps, dtype=np.float32)[::-1].copy() / num_train_timesteps
    base = shift * base / (1 + (shift - 1) * base)
    ts = np.linspace(float(base[0]) * num_train_timesteps, float(base[-1]) * num_train_timesteps, num_inference_steps)
    sig = ts / num_train_timesteps
    sig = shift * sig / (1 + (shift - 1) * sig)
    ts = (sig * num_train_timesteps).astype(np.float32)
    sig = np.concatenate([sig, [0.0]]).astype(np.float32)
    return torch.from_numpy(ts), torch.from_numpy(sig)


@dataclass
class ChainRequest:
    """what RunnerRequest holds for one request (runner/wrappers.py:19-36), on the CPU in fp32"""
    request_id: int
    resolution: int
    num_inference_steps: int
    latents: torch.Tensor            # [1, C, h, w]
    cond: tuple                      # per-model conditioning rows, each [1, ...]: (prompt_embeds, pooled[, time_ids])
    uncond: tuple
    timesteps: torch.Tensor = None
    sigmas: torch.Tensor = None
    step_index: int = 0

    def done(self) -> bool:
        return self.step_index >= self.num_inference_steps


def denoising_step(reqs_by_res: Dict[str, List[ChainRequest]], model: Callable, kind: str, guidance_scale: float,
                   store_dtype: Optional[torch.dtype] = torch.bfloat16) -> None:
    """One timestep for every request, in place.  ``model(latents [2n, ...], timesteps [2n], *conditioning rows)`` is the fp32 oracle forward;
    ``kind`` = "sdxl" (input scaling + epsilon Euler step) or "sd3" (flow-match step, no input scaling)."""
    for res in sorted(reqs_by_res.keys(), key=lambda r: int(r)):                 # ascending resolution, :275-276
        reqs = reqs_by_res[res]
        if not reqs:
            continue
        lat = torch.cat([r.latents for r in reqs]).to(torch.float32)
        sig = torch.stack([r.sigmas[r.step_index] for r in reqs])
        sig_next = torch.stack([r.sigmas[r.step_index + 1] for r in reqs])
        ts = torch.stack([r.timesteps[r.step_index] for r in reqs])
        x2 = torch.cat([lat, lat])
        if kind == "sdxl":
            x2 = scheduler_ref.scale_model_input(x2, torch.cat([sig, sig]))
        ncond = len(reqs[0].cond)
        rows = [torch.cat([r.uncond[k] for r in reqs] + [r.cond[k] for r in reqs]) for k in range(ncond)]
        out = model(x2, torch.cat([ts, ts]), *rows)
        guided = scheduler_ref.cfg_combine(out, guidance_scale)
        new = (scheduler_ref.euler_step if kind == "sdxl" else scheduler_ref.flow_match_step)(guided, lat, sig, sig_next)
        if store_dtype is not None:
            new = new.to(store_dtype).to(torch.float32)
        for i, r in enumerate(reqs):
            r.latents = new[i:i + 1]
            r.step_index += 1
