"""The caller of the model slot: one ``denoising_step`` over a heterogeneous batch of requests, mirroring
``ESyMReDStableDiffusionXLPipeline.denoising_step``
(sduss/model_executor/diffusers/pipelines/stable_diffusion_xl/pipeline_stable_diffusion_xl_esymred.py:259-403)
and the batched Euler scheduler either side of it (schedulers/scheduling_euler_discrete.py:161-274).

What stays host-side is bookkeeping (which request is at which step); the arithmetic -- input scaling with the CFG
duplication, the UNet, the CFG combine and the Euler step -- runs in libmxdenoise.so.  Text encoders and the VAE
(prepare_inference / post_inference) are out of scope and stay on stock PyTorch.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np
import torch

from . import ops
from .step_state import StepCache, new_uid
from .unet import MxUNet


def euler_tables(num_inference_steps: int, num_train_timesteps: int = 1000, beta_start: float = 0.00085,
                 beta_end: float = 0.012, steps_offset: int = 1):
    """diffusers EulerDiscreteScheduler.set_timesteps for the SDXL-base scheduler config (scaled_linear betas,
    'leading' spacing, steps_offset 1, linear sigma interpolation) -- what ``batch_set_timesteps``
    (scheduling_euler_discrete.py:72-113) stores per request.  Returns (timesteps f32[n], sigmas f32[n+1], init_noise_sigma)."""
    betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
    alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
    step_ratio = num_train_timesteps // num_inference_steps
    timesteps = (np.arange(0, num_inference_steps) * step_ratio).round()[::-1].copy().astype(np.float32) + steps_offset
    sig = (((1 - alphas_cumprod) / alphas_cumprod) ** 0.5).numpy()
    sigmas = np.interp(timesteps, np.arange(0, len(sig)), sig)
    sigmas = np.concatenate([sigmas, [0.0]]).astype(np.float32)
    return timesteps, sigmas, float((sigmas.max() ** 2 + 1) ** 0.5)


@dataclass
class Request:
    """The per-request state the reference keeps on RunnerRequest (worker/runner/wrappers.py:19-36)."""
    request_id: int
    resolution: int
    num_inference_steps: int
    latents: torch.Tensor                 # [1, 4, res/8, res/8] model dtype, on device
    prompt_embeds: torch.Tensor           # [1, 77, 2048]
    negative_prompt_embeds: torch.Tensor  # [1, 77, 2048]
    pooled_prompt_embeds: torch.Tensor    # [1, 1280]
    negative_pooled_prompt_embeds: torch.Tensor
    add_time_ids: torch.Tensor            # [1, 6]
    negative_add_time_ids: torch.Tensor
    timesteps: np.ndarray = None
    sigmas: np.ndarray = None
    step_index: int = 0
    arrival: float = 0.0
    start: Optional[float] = None
    finish: Optional[float] = None

    def done(self) -> bool:
        return self.step_index >= self.num_inference_steps


class SDXLDenoiser:
    def __init__(self, unet: MxUNet, guidance_scale: float = 5.0):
        self.unet = unet
        self.guidance_scale = guidance_scale   # reference default (pipeline_..._esymred.py:265)
        self._tables: Dict[int, tuple] = {}
        self.concurrent_resolutions = True
        self._streams: List[torch.cuda.Stream] = []
        self._cache = StepCache(unet.device)      # per batch composition: conditioning cats, sigma/timestep tables (step_state.py)
        self._mixed_cond: Dict[tuple, tuple] = {}  # per mixed composition: the conditioning of all resolutions concatenated

    def set_timesteps(self, req: Request) -> None:
        if req.num_inference_steps not in self._tables:
            self._tables[req.num_inference_steps] = euler_tables(req.num_inference_steps)
        req.timesteps, req.sigmas, _ = self._tables[req.num_inference_steps]
        req.step_index = 0

    def init_noise_sigma(self, num_inference_steps: int) -> float:
        if num_inference_steps not in self._tables:
            self._tables[num_inference_steps] = euler_tables(num_inference_steps)
        return self._tables[num_inference_steps][2]

    @torch.inference_mode()
    def denoising_step(self, worker_reqs: Dict[str, List[Request]], do_classifier_free_guidance: bool = True,
                       is_sliced: bool = False, patch_size: int = 256) -> None:
        """One timestep for every request in ``worker_reqs`` ({str(res): [requests]}), in place.

        The reference runs the resolutions of a mixed batch as one patch batch; here each resolution is its own launch
        sequence, and with more than one resolution present the sequences are issued on separate streams: a small batch
        (one 512-1024 px request fills about a quarter of the CUs) leaves room for the other sequences to run beside it.
        The caller's stream waits for all of them; ``self.concurrent_resolutions = False`` serialises them."""
        res_list = [r for r in sorted(worker_reqs.keys(), key=lambda r: int(r)) if worker_reqs[r]]       # :275-276
        cached_patch_unit = (getattr(self.unet, "_block_caches", None) is not None and is_sliced and 1 <= len(res_list) <= self.unet.max_mixed_groups
                             and all(int(r) % patch_size == 0 and int(r) > patch_size for r in res_list))
        if cached_patch_unit or (1 < len(res_list) <= self.unet.max_mixed_groups and self.unet.mixed_one_sequence
                                 and getattr(self.unet, "_block_caches", None) is None):
            # ONE launch sequence for all resolutions -- also with ESYMRED_USE_CACHE=TRUE (round 4: the cache at its reference unit, the patch,
            # decides once per block for the patches of every resolution: MxUNet.forward -> forward_mixed_cached)
            self._step_mixed(res_list, worker_reqs, do_classifier_free_guidance, is_sliced, patch_size, cached=cached_patch_unit)
            return
        if len(res_list) <= 1 or not self.concurrent_resolutions:
            for res in res_list:
                self._step_resolution(res, worker_reqs[res], do_classifier_free_guidance, is_sliced, patch_size)
            return
        cur = torch.cuda.current_stream()
        fork = torch.cuda.Event()
        fork.record(cur)
        while len(self._streams) < len(res_list):
            self._streams.append(torch.cuda.Stream(device=self.unet.device))
        for i, res in enumerate(res_list):
            side = self._streams[i]
            side.wait_event(fork)
            with torch.cuda.stream(side):
                self._step_resolution(res, worker_reqs[res], do_classifier_free_guidance, is_sliced, patch_size)
            join = torch.cuda.Event()
            join.record(side)
            cur.wait_event(join)
        for res in res_list:                       # the new latents were allocated on side streams: consumers on `cur`
            for r in worker_reqs[res]:             # (post_inference / VAE) are now known to the allocator
                r.latents.record_stream(cur)

    def _gather(self, res: str, reqs: List[Request], do_classifier_free_guidance: bool):
        """the per-resolution gather of :287-339 through the per-composition cache: (entry, latents [n, ...], sigma, sigma_next, timesteps)"""
        def build_cond():
            if do_classifier_free_guidance:                              # :322-339 row order [uncond..., cond...]
                ehs = torch.cat([r.negative_prompt_embeds for r in reqs] + [r.prompt_embeds for r in reqs], dim=0)
                pooled = torch.cat([r.negative_pooled_prompt_embeds for r in reqs] + [r.pooled_prompt_embeds for r in reqs], dim=0)
                tids = torch.cat([t for r in reqs for t in (r.negative_add_time_ids, r.add_time_ids)], dim=0)    # interleaved neg/pos per request (:302-305)
            else:
                ehs = torch.cat([r.prompt_embeds for r in reqs], dim=0)
                pooled = torch.cat([r.pooled_prompt_embeds for r in reqs], dim=0)
                tids = torch.cat([r.add_time_ids for r in reqs], dim=0)
            return ehs, pooled, tids
        e = self._cache.entry((res, do_classifier_free_guidance, tuple(r.request_id for r in reqs), tuple(id(r) for r in reqs)), reqs, build_cond)
        lat = self._cache.latents(e, reqs)
        sig, sig_next, ts = self._cache.step_scalars(e, reqs)
        return e, lat, sig, sig_next, ts

    def _step_mixed(self, res_list: List[str], worker_reqs: Dict[str, List[Request]], do_classifier_free_guidance: bool, is_sliced: bool,
                    patch_size: int, cached: bool = False) -> None:
        """All resolutions of the batch in ONE launch sequence (MxUNet.forward_mixed): the reference runs them as one patch batch
        (:369-380 with the dict of all resolutions; modules/unet.py:242-260).  Conditioning rows: ascending resolution, [uncond..., cond...]
        inside each (:275-276, 327-339)."""
        here = torch.cuda.current_stream()
        parts = []
        for res in res_list:
            reqs = worker_reqs[res]
            for r in reqs:
                r.latents.record_stream(here)
            e, lat, sig, sig_next, ts = self._gather(res, reqs, do_classifier_free_guidance)
            rows = 2 * len(reqs) if do_classifier_free_guidance else len(reqs)
            parts.append((res, reqs, e, lat, sig, sig_next, torch.cat([ts, ts]) if do_classifier_free_guidance else ts,
                          ops.euler_scale_input(lat, sig, rows)))
        key = tuple(id(p[2]) for p in parts)                 # the concatenated conditioning lives as long as its per-resolution entries
        hit = self._mixed_cond.get(key)
        if hit is None or any(a is not b for a, b in zip(hit[0], [p[2] for p in parts])):
            cat = tuple(torch.cat([p[2].cond[k] for p in parts], dim=0) for k in range(3))
            if len(self._mixed_cond) > 32:
                self._mixed_cond.clear()
            hit = self._mixed_cond[key] = ([p[2] for p in parts], cat, new_uid())
        ehs, pooled, tids = hit[1]
        if cached:        # the model slot's own entry: the dict of all resolutions with the request ids the caches are keyed by (:369-380)
            out = self.unet.forward({p[0]: p[7] for p in parts}, torch.cat([p[6] for p in parts]), ehs, added_cond_kwargs={"text_embeds": pooled, "time_ids": tids},
                                    return_dict=False, is_sliced=True, patch_size=patch_size,
                                    input_indices={p[0]: [str(r.request_id) for r in p[1]] for p in parts})[0]
            noise = [out[p[0]] for p in parts]
        else:
            self.unet.set_context_key(hit[2])       # the composition's text embeddings are fixed: K / V^T of all layers once per composition
            noise = self.unet.forward_mixed([p[7] for p in parts], torch.cat([p[6] for p in parts]), ehs, pooled, tids,
                                            gn_patch=(patch_size // 8 if is_sliced else 0))
        g = self.guidance_scale if do_classifier_free_guidance else 0.0
        for (res, reqs, _e, lat, sig, sig_next, _ts, _x), nz in zip(parts, noise):
            ops.cfg_euler_step_(nz, lat, sig, sig_next, g)              # :382-397
            for i, r in enumerate(reqs):                                 # :399-403
                r.step_index += 1
                r.latents = lat[i:i + 1]

    def _step_resolution(self, res: str, reqs: List[Request], do_classifier_free_guidance: bool, is_sliced: bool,
                         patch_size: int) -> None:
        n = len(reqs)
        here = torch.cuda.current_stream()
        for r in reqs:                                                   # latents may have been produced on another stream
            r.latents.record_stream(here)

        def build_cond():
            if do_classifier_free_guidance:                              # :322-339 row order [uncond..., cond...]
                ehs = torch.cat([r.negative_prompt_embeds for r in reqs] + [r.prompt_embeds for r in reqs], dim=0)
                pooled = torch.cat([r.negative_pooled_prompt_embeds for r in reqs] + [r.pooled_prompt_embeds for r in reqs], dim=0)
                # add_time_ids are interleaved neg/pos per request in the reference (:302-305)
                tids = torch.cat([t for r in reqs for t in (r.negative_add_time_ids, r.add_time_ids)], dim=0)
            else:
                ehs = torch.cat([r.prompt_embeds for r in reqs], dim=0)
                pooled = torch.cat([r.pooled_prompt_embeds for r in reqs], dim=0)
                tids = torch.cat([r.add_time_ids for r in reqs], dim=0)
            return ehs, pooled, tids
        # embeddings are fixed for a request's lifetime: one cat per batch composition, not per step (:287-316 redo it every step)
        e = self._cache.entry((res, do_classifier_free_guidance, tuple(r.request_id for r in reqs), tuple(id(r) for r in reqs)), reqs, build_cond)
        ehs, pooled, tids = e.cond
        lat = self._cache.latents(e, reqs)                               # :287-312
        sig, sig_next, ts = self._cache.step_scalars(e, reqs)
        rows = 2 * n if do_classifier_free_guidance else n
        ts2 = torch.cat([ts, ts], dim=0) if do_classifier_free_guidance else ts
        x_in = ops.euler_scale_input(lat, sig, rows)                     # :357-360 (+ the cat of :327)
        self.unet.set_context_key(e.uid)         # the composition's text embeddings are fixed: K / V^T of all layers once per composition
        noise = self.unet.forward({res: x_in}, ts2, ehs, added_cond_kwargs={"text_embeds": pooled, "time_ids": tids},
                                  return_dict=False, is_sliced=is_sliced, patch_size=patch_size,
                                  input_indices={res: [str(r.request_id) for r in reqs]})[0][res]   # :369-380
        g = self.guidance_scale if do_classifier_free_guidance else 0.0
        ops.cfg_euler_step_(noise, lat, sig, sig_next, g)                # :382-397
        for i, r in enumerate(reqs):                                     # :399-403
            r.step_index += 1
            r.latents = lat[i:i + 1]


def synthetic_request(rid: int, resolution: int, steps: int, cfg, denoiser: SDXLDenoiser, device, dtype=torch.bfloat16,
                      seed: int = 10086, shared: Optional[dict] = None) -> Request:
    """Fixed-prompt synthetic request: embeddings ~N(0,1) from the reference seed, time ids
    (res, res, 0, 0, res, res) (pipeline_..._esymred.py:181-187), latents randn * init_noise_sigma."""
    g = torch.Generator(device="cpu").manual_seed(seed + 17 * rid)
    if shared is None or "pe" not in shared:
        ge = torch.Generator(device="cpu").manual_seed(seed)
        pe = torch.randn(1, 77, cfg.cross_attention_dim, generator=ge).to(device=device, dtype=dtype)
        ne = torch.randn(1, 77, cfg.cross_attention_dim, generator=ge).to(device=device, dtype=dtype)
        pp = torch.randn(1, cfg.text_embed_dim, generator=ge).to(device=device, dtype=dtype)
        npp = torch.randn(1, cfg.text_embed_dim, generator=ge).to(device=device, dtype=dtype)
        if shared is not None:
            shared.update(pe=pe, ne=ne, pp=pp, npp=npp)
    else:
        pe, ne, pp, npp = shared["pe"], shared["ne"], shared["pp"], shared["npp"]
    r = float(resolution)
    tid = torch.tensor([[r, r, 0.0, 0.0, r, r]], device=device, dtype=torch.float32)
    lat = torch.randn(1, cfg.in_channels, resolution // 8, resolution // 8, generator=g)
    lat = (lat * denoiser.init_noise_sigma(steps)).to(device=device, dtype=dtype)
    req = Request(rid, resolution, steps, lat, pe, ne, pp, npp, tid, tid.clone())
    denoiser.set_timesteps(req)
    return req
