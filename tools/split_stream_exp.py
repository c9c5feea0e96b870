"""Experiment: one launch sequence over B requests vs the same requests split over concurrent streams (2 x B/2, 4 x B/4).
Usage: python tools/split_stream_exp.py [sdxl|sd3] [batch]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
model = sys.argv[1] if len(sys.argv) > 1 else "sdxl"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = "cuda:0"
if model == "sdxl":
    from sduss_amd.config import UNetConfig
    from sduss_amd.pipeline import SDXLDenoiser, synthetic_request as mk
    from sduss_amd.unet import MxUNet
    from sduss_amd.weights import synthetic_params
    cfg = UNetConfig.sdxl_base()
    den = SDXLDenoiser(MxUNet(cfg, synthetic_params(cfg, device=dev), device=dev), 5.0)
else:
    from sduss_amd.config import MMDiTConfig
    from sduss_amd.pipeline_sd3 import SD3Denoiser, synthetic_sd3_request as mk
    from sduss_amd.transformer_sd3 import MxSD3Transformer
    from sduss_amd.weights import synthetic_mmdit_params
    cfg = MMDiTConfig.sd35_medium()
    den = SD3Denoiser(MxSD3Transformer(cfg, synthetic_mmdit_params(cfg, device=dev), device=dev), 7.0)
shared = {}
reqs = [mk(i, 1024, 1000, cfg, den, dev, shared=shared) for i in range(B)]
streams = [torch.cuda.Stream(device=dev) for _ in range(4)]


def step(groups):
    if groups == 1:
        den._step_resolution("1024", reqs, True, False, 256)
        return
    cur = torch.cuda.current_stream()
    fork = torch.cuda.Event(); fork.record(cur)
    per = B // groups
    for g in range(groups):
        s = streams[g]
        s.wait_event(fork)
        with torch.cuda.stream(s):
            den._step_resolution("1024", reqs[g * per:(g + 1) * per], True, False, 256)
        j = torch.cuda.Event(); j.record(s); cur.wait_event(j)


for groups in (1, 2, 4, 1, 2):
    if B % groups:
        continue
    for _ in range(2):
        step(groups)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(8):
        step(groups)
    torch.cuda.synchronize()
    print(f"{model} B={B} split into {groups} stream(s): {(time.perf_counter() - t0) / 8 * 1e3:.2f} ms/step", flush=True)
