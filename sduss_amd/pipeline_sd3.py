"""SD3.5 caller of the model slot: ``denoising_step`` mirroring ``ESyMReDStableDiffusion3Pipeline.denoising_step``
(sduss/model_executor/diffusers/pipelines/stable_diffusion_3/pipeline_stable_diffusion_3_esymred.py:231-388) with the
batched flow-match Euler step (schedulers/scheduling_flow_match_euler_discrete.py:159-202).  Host side = bookkeeping only."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional

import numpy as np
import torch

from . import ops
from .step_state import StepCache
from .transformer_sd3 import MxSD3Transformer


def flow_match_tables(num_inference_steps: int, num_train_timesteps: int = 1000, shift: float = 3.0):
    """diffusers FlowMatchEulerDiscreteScheduler.set_timesteps (SD3.5 scheduler config: shift 3.0, no dynamic shifting).
    Returns (timesteps f32[n], sigmas f32[n+1])."""
    base = np.linspace(1, num_train_timesteps, num_train_timesteps, dtype=np.float32)[::-1].copy() / num_train_timesteps
    base = shift * base / (1 + (shift - 1) * base)
    sigma_max, sigma_min = float(base[0]), float(base[-1])
    timesteps = np.linspace(sigma_max * num_train_timesteps, sigma_min * num_train_timesteps, num_inference_steps)
    sigmas = timesteps / num_train_timesteps
    sigmas = shift * sigmas / (1 + (shift - 1) * sigmas)
    timesteps = (sigmas * num_train_timesteps).astype(np.float32)
    sigmas = np.concatenate([sigmas, [0.0]]).astype(np.float32)
    return timesteps, sigmas


@dataclass
class SD3Request:
    request_id: int
    resolution: int
    num_inference_steps: int
    latents: torch.Tensor                 # [1, 16, res/8, res/8]
    prompt_embeds: torch.Tensor           # [1, 333, 4096]
    negative_prompt_embeds: torch.Tensor
    pooled_prompt_embeds: torch.Tensor    # [1, 2048]
    negative_pooled_prompt_embeds: torch.Tensor
    timesteps: np.ndarray = None
    sigmas: np.ndarray = None
    step_index: int = 0
    arrival: float = 0.0
    start: Optional[float] = None
    finish: Optional[float] = None

    def done(self) -> bool:
        return self.step_index >= self.num_inference_steps


class SD3Denoiser:
    def __init__(self, transformer: MxSD3Transformer, guidance_scale: float = 7.0):
        self.transformer = transformer
        self.guidance_scale = guidance_scale     # reference default (pipeline_stable_diffusion_3_esymred.py:236)
        self._tables: Dict[int, tuple] = {}
        self.concurrent_resolutions = True
        self._streams: List[torch.cuda.Stream] = []
        self._cache = StepCache(transformer.device)
        self._mixed_cond: Dict[tuple, tuple] = {}   # per mixed composition: the conditioning of all resolutions concatenated

    def set_timesteps(self, req: SD3Request) -> None:
        if req.num_inference_steps not in self._tables:
            self._tables[req.num_inference_steps] = flow_match_tables(req.num_inference_steps)
        req.timesteps, req.sigmas = self._tables[req.num_inference_steps]
        req.step_index = 0

    @torch.inference_mode()
    def denoising_step(self, runner_reqs: Dict[str, List[SD3Request]], do_classifier_free_guidance: bool = True,
                       is_sliced: bool = False, patch_size: int = 256) -> None:
        """One timestep for every request, in place; the resolutions of a mixed batch run on separate streams (see
        SDXLDenoiser.denoising_step)."""
        res_list = [r for r in sorted(runner_reqs.keys(), key=lambda r: int(r)) if runner_reqs[r]]           # :240-241
        tr = self.transformer
        cached_chunk_unit = (getattr(tr, "_block_caches", None) is not None and is_sliced and 1 <= len(res_list) <= tr.max_mixed_groups
                             and all(int(r) % patch_size == 0 and int(r) > patch_size for r in res_list))
        if cached_chunk_unit or (1 < len(res_list) <= tr.max_mixed_groups and tr.mixed_one_sequence and getattr(tr, "_block_caches", None) is None):
            self._step_mixed(res_list, runner_reqs, do_classifier_free_guidance, cached_patch_size=patch_size if cached_chunk_unit else None)
            return
        if len(res_list) <= 1 or not self.concurrent_resolutions:
            for res in res_list:
                self._step_resolution(res, runner_reqs[res], do_classifier_free_guidance, is_sliced, patch_size)
            return
        cur = torch.cuda.current_stream()
        fork = torch.cuda.Event()
        fork.record(cur)
        while len(self._streams) < len(res_list):
            self._streams.append(torch.cuda.Stream(device=self.transformer.device))
        for i, res in enumerate(res_list):
            side = self._streams[i]
            side.wait_event(fork)
            with torch.cuda.stream(side):
                self._step_resolution(res, runner_reqs[res], do_classifier_free_guidance, is_sliced, patch_size)
            join = torch.cuda.Event()
            join.record(side)
            cur.wait_event(join)
        for res in res_list:
            for r in runner_reqs[res]:
                r.latents.record_stream(cur)

    def _step_mixed(self, res_list: List[str], runner_reqs: Dict[str, List[SD3Request]], do_classifier_free_guidance: bool,
                    cached_patch_size: Optional[int] = None) -> None:
        """All resolutions of the batch in ONE launch sequence (MxSD3Transformer.forward_mixed): the reference hands the transformer the dict
        of all resolutions (:312-322) and re-chunks their tokens into one batch (SD3Transformer.py:86).  Rows: ascending resolution,
        [uncond..., cond...] inside each (:240-241, 281-292)."""
        here = torch.cuda.current_stream()
        parts = []
        for res in res_list:
            reqs = runner_reqs[res]
            for r in reqs:
                r.latents.record_stream(here)

            def build_cond(reqs=reqs):
                if do_classifier_free_guidance:
                    return (torch.cat([r.negative_prompt_embeds for r in reqs] + [r.prompt_embeds for r in reqs], dim=0),
                            torch.cat([r.negative_pooled_prompt_embeds for r in reqs] + [r.pooled_prompt_embeds for r in reqs], dim=0))
                return torch.cat([r.prompt_embeds for r in reqs], dim=0), torch.cat([r.pooled_prompt_embeds for r in reqs], dim=0)
            e = self._cache.entry((res, do_classifier_free_guidance, tuple(r.request_id for r in reqs), tuple(id(r) for r in reqs)), reqs, build_cond)
            lat = self._cache.latents(e, reqs)
            sig, sig_next, ts = self._cache.step_scalars(e, reqs)
            if do_classifier_free_guidance:
                x_in, ts2 = ops.euler_scale_input(lat, torch.zeros_like(sig), 2 * len(reqs)), torch.cat([ts, ts], dim=0)
            else:
                x_in, ts2 = lat, ts
            parts.append((reqs, e, lat, sig, sig_next, ts2, x_in))
        key = tuple(id(p[1]) for p in parts)
        hit = self._mixed_cond.get(key)
        if hit is None or any(a is not b for a, b in zip(hit[0], [p[1] for p in parts])):
            if len(self._mixed_cond) > 32:
                self._mixed_cond.clear()
            hit = self._mixed_cond[key] = ([p[1] for p in parts], tuple(torch.cat([p[1].cond[k] for p in parts], dim=0) for k in range(2)))
        ehs, pooled = hit[1]
        if cached_patch_size is not None:     # ESYMRED_USE_CACHE=TRUE: the slot's own entry with the request ids (round 4: the chunk unit, one sequence)
            out = self.transformer.forward({res: p[6] for res, p in zip(res_list, parts)}, encoder_hidden_states=ehs, pooled_projections=pooled,
                                           timestep=torch.cat([p[5] for p in parts]), return_dict=False, is_sliced=True, patch_size=cached_patch_size,
                                           input_indices={res: [str(r.request_id) for r in p[0]] for res, p in zip(res_list, parts)})[0]
            noise = [out[res] for res in res_list]
        else:
            noise = self.transformer.forward_mixed([p[6] for p in parts], torch.cat([p[5] for p in parts]), ehs, pooled)
        g = self.guidance_scale if do_classifier_free_guidance else 0.0
        for (reqs, _e, lat, sig, sig_next, _ts, _x), nz in zip(parts, noise):
            ops.cfg_flow_step_(nz, lat, sig, sig_next, g)
            for i, r in enumerate(reqs):
                r.step_index += 1
                r.latents = lat[i:i + 1]

    def _step_resolution(self, res: str, reqs: List[SD3Request], do_classifier_free_guidance: bool, is_sliced: bool,
                         patch_size: int) -> None:
        n = len(reqs)
        here = torch.cuda.current_stream()
        for r in reqs:                                                   # latents may have been produced on another stream
            r.latents.record_stream(here)

        def build_cond():
            if do_classifier_free_guidance:                                  # :281-292 rows [uncond..., cond...]
                ehs = torch.cat([r.negative_prompt_embeds for r in reqs] + [r.prompt_embeds for r in reqs], dim=0)
                pooled = torch.cat([r.negative_pooled_prompt_embeds for r in reqs] + [r.pooled_prompt_embeds for r in reqs], dim=0)
            else:
                ehs = torch.cat([r.prompt_embeds for r in reqs], dim=0)
                pooled = torch.cat([r.pooled_prompt_embeds for r in reqs], dim=0)
            return ehs, pooled
        e = self._cache.entry((res, do_classifier_free_guidance, tuple(r.request_id for r in reqs), tuple(id(r) for r in reqs)), reqs, build_cond)
        ehs, pooled = e.cond
        lat = self._cache.latents(e, reqs)
        sig, sig_next, ts = self._cache.step_scalars(e, reqs)
        if do_classifier_free_guidance:
            ts2 = torch.cat([ts, ts], dim=0)
            x_in = ops.euler_scale_input(lat, torch.zeros_like(sig), 2 * n)  # exact x/1 copy == torch.cat([latents] * 2)
        else:
            ts2, x_in = ts, lat
        noise = self.transformer.forward({res: x_in}, encoder_hidden_states=ehs, pooled_projections=pooled, timestep=ts2,
                                         return_dict=False, is_sliced=is_sliced, patch_size=patch_size,
                                         input_indices={res: [str(r.request_id) for r in reqs]})[0][res]   # :312-322
        ops.cfg_flow_step_(noise, lat, sig, sig_next, self.guidance_scale if do_classifier_free_guidance else 0.0)  # :362-372
        for i, r in enumerate(reqs):
            r.step_index += 1
            r.latents = lat[i:i + 1]


def synthetic_sd3_request(rid: int, resolution: int, steps: int, cfg, denoiser: SD3Denoiser, device, dtype=torch.bfloat16,
                          seed: int = 10086, shared: Optional[dict] = None, ctx_len: int = 333) -> SD3Request:
    g = torch.Generator(device="cpu").manual_seed(seed + 17 * rid)
    if shared is None or "pe" not in shared:
        ge = torch.Generator(device="cpu").manual_seed(seed)
        pe = torch.randn(1, ctx_len, cfg.joint_attention_dim, generator=ge).to(device=device, dtype=dtype)
        ne = torch.randn(1, ctx_len, cfg.joint_attention_dim, generator=ge).to(device=device, dtype=dtype)
        pp = torch.randn(1, cfg.pooled_projection_dim, generator=ge).to(device=device, dtype=dtype)
        npp = torch.randn(1, cfg.pooled_projection_dim, generator=ge).to(device=device, dtype=dtype)
        if shared is not None:
            shared.update(pe=pe, ne=ne, pp=pp, npp=npp)
    else:
        pe, ne, pp, npp = shared["pe"], shared["ne"], shared["pp"], shared["npp"]
    lat = torch.randn(1, cfg.in_channels, resolution // 8, resolution // 8, generator=g).to(device=device, dtype=dtype)
    req = SD3Request(rid, resolution, steps, lat, pe, ne, pp, npp)
    denoiser.set_timesteps(req)
    return req
