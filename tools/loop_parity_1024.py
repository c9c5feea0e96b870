"""The denoising LOOP at the headline size against the oracle chain, run once per round on the GPU box (too long for the suite: the fp32 oracle
costs ~70 s per CFG step at this size): SDXL-base widths, ONE 1024 x 1024 request of the 50-step schedule, the first N steps (default 5).
Every step: scale -> UNet (batch 2) -> CFG combine -> Euler on both sides, latents rounded to bf16 between steps as the runner keeps them.
Appends one line per step to the output file (argv[2], default gpurun_out/loop_parity_1024.txt).
Usage: python tools/loop_parity_1024.py [steps] [out file] [sdxl | sd3]
sd3: SD3.5-medium widths, ONE 1024 x 1024 request of the 28-step flow-match schedule, guidance 7 (the fp32 oracle costs ~75 s per sample there: ~150 s per step)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import chain_ref, scheduler_ref, sdxl_unet_ref as ref  # noqa: E402  (checker only)
from sduss_amd.config import UNetConfig  # noqa: E402
from sduss_amd.pipeline import SDXLDenoiser, synthetic_request  # noqa: E402
from sduss_amd.unet import MxUNet  # noqa: E402


def main_sd3(steps, log):
    from oracle import sd3_mmdit_ref
    from sduss_amd.config import MMDiTConfig
    from sduss_amd.pipeline_sd3 import SD3Denoiser, synthetic_sd3_request
    from sduss_amd.transformer_sd3 import MxSD3Transformer
    ocfg = sd3_mmdit_ref.MMDiTConfig.sd35_medium()
    P = sd3_mmdit_ref.init_params(ocfg)
    net = MxSD3Transformer(MMDiTConfig.sd35_medium(), P, device="cuda:0")
    den = SD3Denoiser(net, guidance_scale=7.0)
    r = synthetic_sd3_request(0, 1024, 28, MMDiTConfig.sd35_medium(), den, "cuda:0", ctx_len=333)
    ts, sig = chain_ref.sd3_flow_tables(28)
    f = lambda t: t.float().cpu()
    c = chain_ref.ChainRequest(0, 1024, 28, f(r.latents), (f(r.prompt_embeds), f(r.pooled_prompt_embeds)),
                               (f(r.negative_prompt_embeds), f(r.negative_pooled_prompt_embeds)), ts, sig)
    model = lambda x, t, e, p: sd3_mmdit_ref.mmdit_forward(P, ocfg, x, t, e, p)
    log(f"SD3.5-medium 1024 x 1024, one request (batch 2, guidance 7.0), 28-step flow-match schedule, first {steps} steps")
    for n in range(1, steps + 1):
        t0 = time.perf_counter()
        den.denoising_step({"1024": [r]})
        with torch.inference_mode():
            chain_ref.denoising_step({"1024": [c]}, model, "sd3", 7.0)
        d = r.latents.float().cpu() - c.latents
        log(f"after step {n}: rel L2 {float(d.norm() / c.latents.norm()):.4f}  max err {float(d.abs().max() / c.latents.abs().max()):.4f} of range  "
            f"({time.perf_counter() - t0:.0f} s)")


def main():
    torch.set_num_threads(min(torch.get_num_threads(), 32))     # the oracles run fastest on ~32 host threads on the GPU box (profiles/r04_oracle_threads.txt)
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "loop_parity_1024.txt")
    log = lambda m: (print(m, flush=True), open(out, "a").write(m + "\n"))
    if len(sys.argv) > 3 and sys.argv[3] == "sd3":
        return main_sd3(steps, log)
    ocfg = ref.UNetConfig.sdxl_base()
    P = ref.fast_params(ocfg)
    net = MxUNet(UNetConfig.sdxl_base(), P, device="cuda:0")
    den = SDXLDenoiser(net, guidance_scale=5.0)
    r = synthetic_request(0, 1024, 50, UNetConfig.sdxl_base(), den, "cuda:0")
    ts, sig, _ = scheduler_ref.sdxl_euler_tables(50)
    f = lambda t: t.float().cpu()
    c = chain_ref.ChainRequest(0, 1024, 50, f(r.latents), (f(r.prompt_embeds), f(r.pooled_prompt_embeds), f(r.add_time_ids)),
                               (f(r.negative_prompt_embeds), f(r.negative_pooled_prompt_embeds), f(r.negative_add_time_ids)), ts, sig)
    P32 = {k: v.float() for k, v in P.items()}
    model = lambda x, t, e, te, ti: ref.unet_forward(P32, ocfg, x, t, e, te, ti)
    log(f"SDXL-base 1024 x 1024, one request (UNet batch 2, CFG 5.0), 50-step Euler schedule, first {steps} steps; oracle on the ORIGINAL weights")
    for n in range(1, steps + 1):
        t0 = time.perf_counter()
        den.denoising_step({"1024": [r]})
        with torch.inference_mode():
            chain_ref.denoising_step({"1024": [c]}, model, "sdxl", 5.0)
        d = r.latents.float().cpu() - c.latents
        log(f"after step {n}: rel L2 {float(d.norm() / c.latents.norm()):.4f}  max err {float(d.abs().max() / c.latents.abs().max()):.4f} of range  "
            f"({time.perf_counter() - t0:.0f} s)")


if __name__ == "__main__":
    main()
