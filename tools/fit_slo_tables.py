"""Regenerate, for MI355X, the two SLO inputs of the reference's worker scheduler that the latency predictor does not cover (SURVEY.md section
8f rank 3, second half):

  * sduss/worker/scheduler/configs/esymred.json  "STANDALONE": seconds a request takes ALONE on a GPU per stage and resolution -- denoising
    (the whole loop) and postprocessing (the VAE decode) -- which esymred_utils.py:27-43 multiplies by the SLO factor to get each request's
    deadline, and which Hyper_Parameter.postprocessing_ratio weighs;
  * exp/profile/sm_util_<model>_<res>.csv  "sm util, unet time, post time": one row per batch size 1..8 -- seconds for the denoising loop of a
    batch of n requests of that resolution and for the VAE decode of its n images (the first column is a constant 10 in the reference's files:
    its SmUtilMonitor returns None, engine/utils.py:58-68; kept as is).

Everything is measured through the HIP library on random-init weights of the real architectures, with the step counts of the reference's
profiles (50 steps; the tables are per-stage totals, not per-step).  Outputs (profiles/): esymred_mi355x.json (the reference's file with its
STANDALONE block replaced; Hyper_Parameter / DISCARD_SLACK copied from the values the reference ships) and exec_time_mi355x/sm_util_<model>_<res>.csv
-- the reference's file names, so that ESYMRED_EXEC_TIME_DIR can point at the directory (policy/ESyMReD.py:81, 105-118 reads the LAST column).

Usage (GPU box): python tools/fit_slo_tables.py --out-dir gpurun_out/slo [--models sdxl,sd3] [--steps 10] [--max-batch 8]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

REF_HYPER = {  # sduss/worker/scheduler/configs/esymred.json:2-19, 46 -- policy knobs, not measurements: shipped unchanged
    "Hyper_Parameter": {"sd3": {"postprocessing": {"256": 1, "512": 1, "768": 1}}, "sdxl": {"postprocessing": {"512": 1, "768": 1, "1024": 1}},
                        "get_best_tp_th": 1, "active_queue_timeout_th": 0.1, "postprocessing_ratio": 0.9},
    "DISCARD_SLACK": 500}
STEPS_PER_LOOP = 50


def timed(fn, n):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def vae_for(model, device):
    from sduss_amd.vae import MxVAEDecoder, VAEConfig
    from vae_bench import shapes
    cfg = VAEConfig.sdxl() if model == "sdxl" else VAEConfig.sd3()
    g = torch.Generator().manual_seed(1)
    P = {}
    for k, s in shapes(cfg).items():
        if len(s) > 1:
            P[k] = torch.randn(s, generator=g) * float(np.prod(s[1:])) ** -0.5
        else:
            P[k] = (1.0 if k.endswith("weight") else 0.0) + 0.05 * torch.randn(s, generator=g)
    return cfg, MxVAEDecoder(cfg, P, device=device, out_dtype=torch.bfloat16)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--models", default="sdxl,sd3")
    ap.add_argument("--steps", type=int, default=10, help="timed denoising steps per batch (the loop time is steps_per_loop x the mean step)")
    ap.add_argument("--max-batch", type=int, default=8)
    ap.add_argument("--out-dir", default="gpurun_out/slo")
    args = ap.parse_args()
    os.makedirs(args.out_dir, exist_ok=True)
    sys.path.insert(0, ROOT)
    import bench
    device = torch.device("cuda:0")
    standalone = {}
    for model in args.models.split(","):
        cfg, net, den, P = bench.build_model(model, device)
        del P
        vcfg, vae = vae_for(model, device)
        standalone[model] = {"denoising": {}, "postprocessing": {}}
        for res in (512, 768, 1024):
            rows = []
            for n in range(1, args.max_batch + 1):
                bench.STEPS_PER_IMAGE = 1000                    # never finishes inside the timing loop
                reqs = bench.make_batch(den, cfg, n, res, device, {}, base_id=res * 100 + n * 10)
                step_s = timed(lambda: den.denoising_step({str(res): reqs}, is_sliced=False), args.steps)
                lat = torch.randn(n, vcfg.latent_channels, res // 8, res // 8, device=device, dtype=torch.bfloat16)
                # (the decoder's 128-channel full-resolution level indexes its activations with 32 bits: images go through it four at a time)
                post_s = timed(lambda: [vae.decode(lat[k:k + 4]) for k in range(0, n, 4)], 3)
                rows.append((10, STEPS_PER_LOOP * step_s, post_s))
                print(f"{model} {res} px batch {n}: {1e3 * step_s:.2f} ms/step -> {STEPS_PER_LOOP * step_s:.3f} s per {STEPS_PER_LOOP}-step loop, "
                      f"VAE decode {1e3 * post_s:.1f} ms", flush=True)
                del reqs, lat
            os.makedirs(os.path.join(args.out_dir, "exec_time_mi355x"), exist_ok=True)
            with open(os.path.join(args.out_dir, "exec_time_mi355x", f"sm_util_{model}_{res}.csv"), "w") as f:
                f.write("sm util, unet time, post time\n")          # header of exp/profile/sm_util_sdxl_1024.csv
                for u, a, b in rows:
                    f.write(f"{u},{a},{b}\n")
            standalone[model]["denoising"][str(res)] = round(rows[0][1], 3)
            standalone[model]["postprocessing"][str(res)] = round(rows[0][2], 4)
        del net, den, vae
        torch.cuda.empty_cache()
    out = dict(REF_HYPER)
    out = {"Hyper_Parameter": REF_HYPER["Hyper_Parameter"], "STANDALONE": standalone, "DISCARD_SLACK": REF_HYPER["DISCARD_SLACK"]}
    with open(os.path.join(args.out_dir, "esymred_mi355x.json"), "w") as f:
        json.dump(out, f, indent=4)
    print(json.dumps(standalone))


if __name__ == "__main__":
    main()
