"""Re-fit the block-skip predictors on this stack.  The reference loads cuML random forests
(exp/sdxl-downsample-threshold0.01.pkl, exp/sdxl-upsample-threshold0.01.pkl, exp/sd3-state-threshold0.01.pkl; cache_manager.py:36-44) that
need cuML to unpickle.  This tool regenerates the same kind of object with scikit-learn from traces of THIS denoiser:

  1. run requests through the cached entry with every block running and the observer on: per block and step the library reports the feature row
     the predictor would see -- [block, timestep, input mse (, mse of each skip)] -- and how far the block's output moved since its last run;
  2. label = 1 (run) when the output moved by more than --threshold (the reference's file names carry it: 0.01), else 0 (reuse);
  3. fit one RandomForestClassifier for the down + mid blocks and one for the up blocks (SDXL), or one for the joint blocks (SD3), dump with joblib;
  4. replay the same requests with the fitted predictors: fraction of blocks reused, step time, distance of the final latents from the exact run.

With the random-init weights of this repository the traces are not those of a trained model: the numbers show the mechanism, the files are a
format check.  With real weights (MxUNet(load_safetensors_dir(...))) the same command fits usable predictors.
Usage on the GPU box: python tools/fit_skip_predictor.py --model sdxl --res 512 --requests 2 --steps 30 --out-dir gpurun_out"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sduss_amd.block_cache import MSE_UNCACHED, FORCED_RUN_AFTER, FORCED_RUN_AFTER_SD3  # noqa: E402


class AlwaysRun:
    def predict(self, f):
        return np.ones(len(f))


def rows_and_labels(cache, threshold):
    """pair the feature rows of the blocks that had a cached input with the observed output movement (same order: a block that ran while
    cached reports once to each)"""
    feats = [f for f in cache.features if not (f[:, 2] >= MSE_UNCACHED * 0.5).any()]
    assert len(feats) == len(cache.observed), (len(feats), len(cache.observed))
    X, y = [], []
    for f, (block, om) in zip(feats, cache.observed):
        assert int(f[0, 0]) == block
        X.append(f); y.append((om > threshold).astype(np.int64))
    return X, y


def fit(X, y, n_feat):
    from sklearn.ensemble import RandomForestClassifier
    rows = np.concatenate([x for x in X if x.shape[1] == n_feat])
    lab = np.concatenate([l for x, l in zip(X, y) if x.shape[1] == n_feat])
    rf = RandomForestClassifier(n_estimators=32, max_depth=8, random_state=0)
    rf.fit(rows, lab)
    return rf, rows, lab


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", choices=["sdxl", "sd3"], default="sdxl")
    ap.add_argument("--res", type=int, default=512)
    ap.add_argument("--requests", type=int, default=2)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--threshold", type=float, default=0.01)
    ap.add_argument("--threshold-quantile", type=float, default=0.0,
                    help="instead of --threshold: label this fraction of the observed block outputs reusable (random-init weights move far more "
                         "per step than a trained model, so 0.01 labels nothing)")
    ap.add_argument("--tiny", action="store_true", help="the tiny test configuration (plumbing check)")
    ap.add_argument("--out-dir", default="gpurun_out")
    args = ap.parse_args()
    import joblib
    dev = torch.device("cuda:0")
    if args.model == "sdxl":
        from sduss_amd.config import UNetConfig
        from sduss_amd.pipeline import SDXLDenoiser, synthetic_request
        from sduss_amd.unet import MxUNet
        from sduss_amd.weights import synthetic_params
        cfg = UNetConfig.tiny() if args.tiny else UNetConfig.sdxl_base()
        net = MxUNet(cfg, synthetic_params(cfg, device=dev), device=dev)
        den, make, forced = SDXLDenoiser(net), synthetic_request, FORCED_RUN_AFTER
    else:
        from sduss_amd.config import MMDiTConfig
        from sduss_amd.pipeline_sd3 import SD3Denoiser, synthetic_sd3_request
        from sduss_amd.transformer_sd3 import MxSD3Transformer
        from sduss_amd.weights import synthetic_mmdit_params
        cfg = MMDiTConfig.tiny() if args.tiny else MMDiTConfig.sd35_medium()
        net = MxSD3Transformer(cfg, synthetic_mmdit_params(cfg, device=dev), device=dev)
        den, make, forced = SD3Denoiser(net), synthetic_sd3_request, FORCED_RUN_AFTER_SD3
    res = str(args.res)

    def run(label):
        shared = {}
        reqs = {res: [make(i, args.res, args.steps, cfg, den, dev, shared=shared) for i in range(args.requests)]}
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            den.denoising_step(reqs)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / args.steps * 1e3
        print(f"{label}: {ms:.2f} ms/step", flush=True)
        return torch.cat([r.latents for r in reqs[res]]).float(), ms

    exact, ms_exact = run("exact path")
    # 1. traces
    if args.model == "sdxl":
        net.enable_block_cache(AlwaysRun(), AlwaysRun(), observe=True)
    else:
        net.enable_block_cache(AlwaysRun(), observe=True)
    traced, _ = run("cached entry, every block run, observer on")
    assert torch.equal(traced, exact), "every block run must be the exact path"
    cache = net._block_caches[res]
    moved = np.concatenate([m for _b, m in cache.observed])
    print("observed output movement (mse) percentiles 5/25/50/75/95: " + " ".join(f"{v:.3g}" for v in np.percentile(moved, [5, 25, 50, 75, 95])))
    if args.threshold_quantile > 0:
        args.threshold = float(np.quantile(moved, args.threshold_quantile))
        print(f"threshold from the {args.threshold_quantile:.2f} quantile: {args.threshold:.4g}")
    X, y = rows_and_labels(cache, args.threshold)
    net.disable_block_cache()
    # 2-3. fit
    os.makedirs(args.out_dir, exist_ok=True)
    preds = {}
    for name, n_feat in (("downsample", 3), ("upsample", 6)) if args.model == "sdxl" else (("state", 3),):
        rf, rows, lab = fit(X, y, n_feat)
        path = os.path.join(args.out_dir, f"{args.model}-{name}-threshold{args.threshold:.3g}-mi355x.pkl")
        joblib.dump(rf, path)
        preds[name] = joblib.load(path)
        print(f"{name}: {len(rows)} rows, {100 * (1 - lab.mean()):.1f} % labelled reusable, training accuracy {rf.score(rows, lab):.3f}, "
              f"input mse range [{rows[:, 2].min():.3g}, {rows[:, 2].max():.3g}] -> {path}", flush=True)
    # 4. replay with the fitted predictors
    if args.model == "sdxl":
        net.enable_block_cache(preds["downsample"], preds["upsample"], forced_after=forced)
    else:
        net.enable_block_cache(preds["state"], forced_after=forced)
    approx, ms_cached = run("cached entry, fitted predictors")
    hist = net._block_caches[res].history
    n_blocks = 7 if args.model == "sdxl" else cfg.num_layers
    ran = sum(bin(h).count("1") for h in hist)
    rel = float((approx - exact).norm() / exact.norm())
    print(f"blocks run: {ran} of {n_blocks * len(hist)} ({100 * (1 - ran / (n_blocks * len(hist))):.1f} % reused); step {ms_exact:.2f} -> {ms_cached:.2f} ms; "
          f"final latents: relative L2 to the exact run {rel:.4f}")


if __name__ == "__main__":
    main()
