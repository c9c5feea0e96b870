"""Generates the golden fixtures in this directory from the CPU oracle (oracle/*.py, fp32 torch on CPU).

    PYTHONPATH=. python tests/golden/make_golden.py

What the fixtures are.  The reference holds no golden vectors for this path and cannot be imported here
(diffusers / xformers / CUDA are absent -- SURVEY.md section 8c), so these are outputs of THIS repo's restatement:
they pin the oracle against drift (tests/test_cpu.py::test_oracle_reproduces_golden*) and give the GPU tests a
fixed expected output that does not depend on the oracle code running on the GPU box.  They do not upgrade the
oracle from "parity unpinned".  The only externally published known answers used anywhere are the scheduler
constants checked in tests/test_cpu.py::test_scheduler_published_constants.

Each .npz holds the seeded INPUTS (so a different torch RNG cannot silently change the case), a float64 checksum of
the seeded weights, and the expected OUTPUT.  Total size < 1 MB.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import patch_ref, scheduler_ref, sd3_mmdit_ref as sd3, sdxl_unet_ref as ref  # noqa: E402


def weights_checksum(P) -> float:
    return float(sum(v.double().abs().sum().item() for _, v in sorted(P.items())))


def f16(x):
    """inputs are stored in fp16-exact form so the file stays small and both sides read identical values"""
    return x.to(torch.float16).float()


def sdxl_cases():
    cfg = ref.UNetConfig.tiny()
    P = ref.init_params(cfg)
    s, t, e, te, ti = ref.make_inputs(cfg, 2, 32)
    s, e, te = f16(s), f16(e), f16(te)
    with torch.inference_mode():
        out = ref.unet_forward(P, cfg, s, t, e, te, ti)
        out_sliced = patch_ref.unet_forward_sliced(P, cfg, {"256": s}, t, e, te, ti, patch_size=64)["256"]
    np.savez_compressed(os.path.join(HERE, "sdxl_tiny_b2_32x32.npz"), sample=s.numpy().astype(np.float16),
                        timestep=t.numpy(), encoder_hidden_states=e.numpy().astype(np.float16),
                        text_embeds=te.numpy().astype(np.float16), time_ids=ti.numpy(),
                        weights_checksum=np.float64(weights_checksum(P)), out_unsliced=out.numpy(),
                        out_sliced_patch64=out_sliced.numpy())


def sd3_case():
    cfg = sd3.MMDiTConfig.tiny()
    P = sd3.init_params(cfg)
    lat, t, ehs, pooled = sd3.make_inputs(cfg, 2, 16, ctx_len=37)
    lat, ehs, pooled = f16(lat), f16(ehs), f16(pooled)
    with torch.inference_mode():
        out = sd3.mmdit_forward(P, cfg, lat, t, ehs, pooled)
    np.savez_compressed(os.path.join(HERE, "sd3_tiny_b2_16x16.npz"), latents=lat.numpy().astype(np.float16), timestep=t.numpy(),
                        encoder_hidden_states=ehs.numpy().astype(np.float16), pooled=pooled.numpy().astype(np.float16),
                        weights_checksum=np.float64(weights_checksum(P)), out=out.numpy())


def gn_halo_case():
    """inner boundary: groupnorm(..., padding=True) on two latents of 2x2 and 1x1 patches (SURVEY section 8b inner signature)"""
    g = torch.Generator().manual_seed(7)
    lat_a = f16(torch.randn(1, 8, 16, 16, generator=g) * 1.5 + 0.3)
    lat_b = f16(torch.randn(1, 8, 8, 8, generator=g) * 0.7 - 0.2)
    padding_idx, latent_offset, _, halod, patch_map = patch_ref.split_sample({"64": lat_b, "128": lat_a}, 64)
    patches = halod[:, :, 1:-1, 1:-1].contiguous()      # the op receives patch interiors [N, C, 8, 8]
    gamma = f16(torch.randn(8, generator=g)); beta = f16(torch.randn(8, generator=g))
    y = patch_ref.groupnorm(patches, gamma, beta, 2, 1e-5, True, latent_offset, patch_map, padding_idx)
    m = patch_ref.mock_groupnorm(patches, padding_idx)
    np.savez_compressed(os.path.join(HERE, "gn_halo_2latents.npz"), x=patches.numpy().astype(np.float16), gamma=gamma.numpy(),
                        beta=beta.numpy(), latent_offset=np.asarray(latent_offset, np.int32),
                        patch_map=np.asarray(patch_map, np.int32), padding_idx=np.asarray(padding_idx, np.int32).reshape(-1),
                        cpg=np.int32(2), eps=np.float64(1e-5), y=y.numpy(), y_mock=m.numpy())


def scheduler_case():
    ts, sig, init = scheduler_ref.sdxl_euler_tables(50)
    g = torch.Generator().manual_seed(11)
    x = f16(torch.randn(2, 4, 8, 8, generator=g) * init)
    eps = f16(torch.randn(4, 4, 8, 8, generator=g))
    idx = torch.tensor([3, 17])
    scaled = scheduler_ref.scale_model_input(torch.cat([x, x]), sig[idx].repeat(2))
    stepped = scheduler_ref.euler_step(scheduler_ref.cfg_combine(eps, 5.0), x, sig[idx], sig[idx + 1])
    v = f16(torch.randn(4, 4, 8, 8, generator=g))
    fs = torch.tensor([0.9, 0.4]); fn = torch.tensor([0.85, 0.3])
    flow = scheduler_ref.flow_match_step(scheduler_ref.cfg_combine(v, 7.0), x, fs, fn)
    np.savez_compressed(os.path.join(HERE, "scheduler_steps.npz"), timesteps50=ts.numpy(), sigmas50=sig.numpy(),
                        init_noise_sigma=np.float64(init), x=x.numpy(), eps=eps.numpy(), step_index=idx.numpy(),
                        scaled=scaled.numpy(), euler_out=stepped.numpy(), v=v.numpy(), flow_sigma=fs.numpy(),
                        flow_sigma_next=fn.numpy(), flow_out=flow.numpy())


if __name__ == "__main__":
    torch.set_num_threads(4)
    sdxl_cases(); sd3_case(); gn_halo_case(); scheduler_case()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))
