"""Re-fit the reference's step-latency predictor for MI355X.

The reference's worker scheduler ranks request slack with constants measured on H100:
  * exp/profile/unet_time_{sdxl,sd3}.csv   "512 num, 768 num, 1024 num, avg unet time"  (seconds for 50 steps of one batch)
  * exp/schedule_predictor_{sdxl,sd3}.pkl  sklearn MLPRegressor(32, 32, 16) on the 5 features of
    Predictor.predict (sduss/worker/scheduler/policy/ESyMReD.py:48-53): a, b, c, 4a + 9b + 16c, #non-zero
  * Predictor.latency  per-resolution single-request step time (ESyMReD.py:30-41)
With a different denoiser those constants mis-rank slack.  This tool measures the same table on this GPU with the same
batch compositions the reference sampled (counts of 512 / 768 / 1024 px requests, mixed batches sliced at patch 256 as
FCFS_Mixed.py:69-70 forces) and fits the same model.  Outputs (profiles/):
  unet_time_<model>_mi355x.csv, schedule_predictor_<model>_mi355x.pkl, predictor_<model>_mi355x.txt (fit report)

Usage (GPU box): python tools/fit_predictor.py [--model sdxl|sd3] [--steps 5] [--max-total 8] [--caps 12,8,5] --out-dir gpurun_out/pred
The composition list is seeded (numpy seed 10086) and capped by --max-total requests per batch.
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def compositions(max_total: int, n: int, seed: int = 10086, caps=(12, 8, 5)):
    """every single-resolution batch up to its cap plus a seeded sample of mixed ones.  The reference's table
    (exp/profile/unet_time_sdxl.csv) has 221 rows with up to 12 / 8 / 5 requests of 512 / 768 / 1024 px: `caps`; max_total bounds the
    pixel load of a mixed batch in units of 1024 px requests (4 a + 9 b + 16 c <= 16 max_total), as a batch the scheduler would form."""
    out = []
    for res_idx in range(3):
        for k in range(1, caps[res_idx] + 1):
            c = [0, 0, 0]
            c[res_idx] = k
            out.append(tuple(c))
    rng = np.random.RandomState(seed)
    seen = set(out)
    tries = 0
    while len(out) < n and tries < 100000:
        tries += 1
        a, b, c = (int(rng.randint(0, caps[i] + 1)) for i in range(3))
        if a + b + c > 0 and 4 * a + 9 * b + 16 * c <= 16 * max_total and (a, b, c) not in seen:
            seen.add((a, b, c))
            out.append((a, b, c))
    return out


def features(t):
    t = np.asarray(t, dtype=np.float64)
    return np.concatenate([t, t[:, :1] * 4 + t[:, 1:2] * 9 + t[:, 2:3] * 16, np.count_nonzero(t, axis=1)[:, None]], axis=1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", choices=["sdxl", "sd3"], default="sdxl")
    ap.add_argument("--steps", type=int, default=5, help="timed denoising steps per composition (after one warm-up step)")
    ap.add_argument("--max-total", type=int, default=8)
    ap.add_argument("--rows", type=int, default=221)
    ap.add_argument("--caps", default="12,8,5", help="largest single-resolution batch per resolution (the reference table's range)")
    ap.add_argument("--out-dir", default=None, help="where the table and the fit go (default profiles/; on the GPU box use gpurun_out/pred)")
    ap.add_argument("--fit-only", action="store_true", help="re-fit from the committed profiles/unet_time_<model>_mi355x.csv (no GPU)")
    args = ap.parse_args()
    out_dir = args.out_dir or os.path.join(ROOT, "profiles")
    os.makedirs(out_dir, exist_ok=True)
    csv_path = os.path.join(out_dir, f"unet_time_{args.model}_mi355x.csv")
    if args.fit_only:
        rows = [tuple(float(x) for x in l.split(",")) for l in open(csv_path).read().splitlines()[1:]]
        rows = [(int(a), int(b), int(c), t) for a, b, c, t in rows]
        return fit(args, rows, out_dir)
    dev = "cuda:0"
    torch.cuda.set_device(0)
    if args.model == "sdxl":
        from sduss_amd.config import UNetConfig
        from sduss_amd.pipeline import SDXLDenoiser, synthetic_request as mk
        from sduss_amd.unet import MxUNet
        from sduss_amd.weights import synthetic_params
        cfg = UNetConfig.sdxl_base()
        net = MxUNet(cfg, synthetic_params(cfg, device=dev), device=dev)
        den = SDXLDenoiser(net, guidance_scale=5.0)
    else:
        from sduss_amd.config import MMDiTConfig
        from sduss_amd.pipeline_sd3 import SD3Denoiser, synthetic_sd3_request as mk
        from sduss_amd.transformer_sd3 import MxSD3Transformer
        from sduss_amd.weights import synthetic_mmdit_params
        cfg = MMDiTConfig.sd35_medium()
        net = MxSD3Transformer(cfg, synthetic_mmdit_params(cfg, device=dev), device=dev)
        den = SD3Denoiser(net, guidance_scale=7.0)
    shared = {}
    comps = compositions(args.max_total, args.rows, caps=tuple(int(x) for x in args.caps.split(",")))
    rows = []
    t_start = time.time()
    for idx, (a, b, c) in enumerate(comps):
        reqs = {}
        rid = 0
        for res, n in ((512, a), (768, b), (1024, c)):
            if n:
                reqs[str(res)] = [mk(rid + i, res, 50, cfg, den, dev, shared=shared) for i in range(n)]
                rid += n
        sliced = True                                  # FCFS_Mixed.py:69-70: is_sliced=True, patch_size=256 always
        den.denoising_step(reqs, is_sliced=sliced, patch_size=256)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            den.denoising_step(reqs, is_sliced=sliced, patch_size=256)
        torch.cuda.synchronize()
        step = (time.perf_counter() - t0) / args.steps
        rows.append((a, b, c, step * 50.0))
        if idx % 10 == 0:
            print(f"[{idx + 1}/{len(comps)}] {a},{b},{c}: {step * 1e3:.1f} ms/step  (elapsed {time.time() - t_start:.0f} s)", flush=True)
        del reqs
    with open(csv_path, "w") as f:
        f.write("512 num, 768 num, 1024 num, avg unet time\n")
        for a, b, c, t in rows:
            f.write(f"{a},{b},{c},{t}\n")
    fit(args, rows, out_dir)


def fit(args, rows, out_dir):
    # the reference's model class and features
    from sklearn.neural_network import MLPRegressor
    from sklearn.pipeline import make_pipeline
    from sklearn.preprocessing import StandardScaler
    import joblib
    X = features([r[:3] for r in rows]); y = np.asarray([r[3] for r in rows])
    rng = np.random.RandomState(10086)
    perm = rng.permutation(len(rows))
    n_test = max(8, len(rows) // 6)
    te, tr = perm[:n_test], perm[n_test:]
    # same network as the reference's pickles; inputs standardised and the small table fitted with L-BFGS (the object keeps
    # the .predict(features) interface Predictor.predict calls)
    def make():
        return make_pipeline(StandardScaler(), MLPRegressor(hidden_layer_sizes=(32, 32, 16), solver="lbfgs", alpha=1e-3, max_iter=5000,
                                                            random_state=10086))
    model = make()
    model.fit(X[tr], y[tr])
    err_tr = np.abs(model.predict(X[tr]) - y[tr]) / y[tr]
    err_te = np.abs(model.predict(X[te]) - y[te]) / y[te]
    model = make()
    model.fit(X, y)                                    # the shipped predictor is fitted on every row
    joblib.dump(model, os.path.join(out_dir, f"schedule_predictor_{args.model}_mi355x.pkl"))
    singles = {res: next(t for a, b, c, t in rows if (a, b, c) == tuple(1 if i == j else 0 for j in range(3))) / 50.0
               for i, res in enumerate((512, 768, 1024))}
    with open(os.path.join(out_dir, f"predictor_{args.model}_mi355x.txt"), "w") as f:
        f.write(f"{args.model} on MI355X: {len(rows)} batch compositions (<= {max(sum(r[:3]) for r in rows)} requests), {args.steps} timed steps each, "
                f"is_sliced=True patch 256, CFG, bf16, synthetic weights\n")
        f.write("features: a, b, c, 4a+9b+16c, #non-zero (ESyMReD.py:48-53); target: seconds per 50 steps; StandardScaler + MLPRegressor(32,32,16)\n")
        f.write(f"hold-out ({n_test} rows): mean relative error {err_te.mean():.4f}, max {err_te.max():.4f}; "
                f"train: mean {err_tr.mean():.4f}, max {err_tr.max():.4f}\n")
        f.write("Predictor.latency (single request, seconds per step): " + ", ".join(f'"{r}": {v:.4f}' for r, v in singles.items()) + "\n")
        f.write("reference (H100) values: sdxl 0.04 / 0.045 / 0.054, sd3 0.0414 / 0.0574 / 0.065 (ESyMReD.py:30-41)\n")
    print(open(os.path.join(out_dir, f"predictor_{args.model}_mi355x.txt")).read())


if __name__ == "__main__":
    main()
