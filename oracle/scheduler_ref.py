"""CPU restatement of the element-wise steps either side of the UNet call.

TEST INFRASTRUCTURE (see oracle/__init__.py).  PARITY UNPINNED.

* ``EulerDiscreteScheduler.batch_scale_model_input`` / ``batch_step`` --
  /root/reference/sduss/model_executor/diffusers/schedulers/scheduling_euler_discrete.py:161-274
  (epsilon prediction, gamma = 0, fp32 upcast :210, cast back :268).
* CFG combine -- pipelines/stable_diffusion_xl/pipeline_stable_diffusion_xl_esymred.py:382-385.
* sigma/timestep tables: diffusers==0.32.1 ``EulerDiscreteScheduler.set_timesteps`` with the SDXL-base scheduler
  config (scaled_linear betas 0.00085..0.012, 1000 train steps, timestep_spacing 'leading', steps_offset 1,
  interpolation 'linear', no karras) -- third-party, restated from its published definition.
* ``FlowMatchEulerDiscreteScheduler.batch_step`` -- schedulers/scheduling_flow_match_euler_discrete.py:159-202.
"""
from __future__ import annotations

import numpy as np
import torch


def sdxl_euler_tables(num_inference_steps: int, num_train_timesteps: int = 1000, beta_start: float = 0.00085,
                      beta_end: float = 0.012, steps_offset: int = 1):
    """Returns (timesteps[float32, n], sigmas[float32, n+1], init_noise_sigma)."""
    betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
    alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
    step_ratio = num_train_timesteps // num_inference_steps
    timesteps = (np.arange(0, num_inference_steps) * step_ratio).round()[::-1].copy().astype(np.float32)
    timesteps += steps_offset
    sig = (((1 - alphas_cumprod) / alphas_cumprod) ** 0.5).numpy()
    sigmas = np.interp(timesteps, np.arange(0, len(sig)), sig)
    sigmas = np.concatenate([sigmas, [0.0]]).astype(np.float32)
    init_noise_sigma = float((sigmas.max() ** 2 + 1) ** 0.5)  # 'leading' spacing
    return torch.from_numpy(timesteps), torch.from_numpy(sigmas), init_noise_sigma


def scale_model_input(samples: torch.Tensor, sigmas: torch.Tensor) -> torch.Tensor:
    """x / sqrt(sigma^2 + 1); ``sigmas`` [B] (repeated x2 by the caller under CFG, :176-178)."""
    s = sigmas.to(samples.dtype).reshape(-1, *([1] * (samples.ndim - 1)))
    return samples / ((s ** 2 + 1) ** 0.5)


def cfg_combine(noise_pred: torch.Tensor, guidance_scale: float) -> torch.Tensor:
    u, t = noise_pred.chunk(2)
    return u + guidance_scale * (t - u)


def euler_step(model_output: torch.Tensor, samples: torch.Tensor, sigma: torch.Tensor,
               sigma_next: torch.Tensor) -> torch.Tensor:
    """epsilon-prediction Euler step in fp32, result cast to the model-output dtype (:210-268)."""
    x = samples.to(torch.float32)
    shape = (-1, *([1] * (x.ndim - 1)))
    s = sigma.to(torch.float32).reshape(shape)
    sn = sigma_next.to(torch.float32).reshape(shape)
    pred_x0 = x - s * model_output
    d = (x - pred_x0) / s
    return (x + d * (sn - s)).to(model_output.dtype)


def flow_match_step(model_output: torch.Tensor, samples: torch.Tensor, sigma: torch.Tensor,
                    sigma_next: torch.Tensor) -> torch.Tensor:
    """x + (sigma_next - sigma) * v in fp32, cast back (scheduling_flow_match_euler_discrete.py:186-196)."""
    x = samples.to(torch.float32)
    shape = (-1, *([1] * (x.ndim - 1)))
    dt = (sigma_next.to(torch.float32) - sigma.to(torch.float32)).reshape(shape)
    return (x + dt * model_output).to(model_output.dtype)
