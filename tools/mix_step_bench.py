"""Mixed-resolution step (configs[4] shape): what the per-resolution launch sequences on concurrent streams reach
against (a) the same sequences back to back and (b) the time the step's arithmetic would take at the rate of the headline batch
(4 x 1024 px: the chip full, no launch-bound tail) -- the bound a single variable-length launch sequence could approach.
Usage on the GPU box: python tools/mix_step_bench.py > gpurun_out/mix_step_bench.log"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sduss_amd.config import UNetConfig  # noqa: E402
from sduss_amd.pipeline import SDXLDenoiser, synthetic_request  # noqa: E402
from sduss_amd.unet import MxUNet  # noqa: E402
from sduss_amd.weights import synthetic_params  # noqa: E402


def timed(fn, n=8):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    dev = torch.device("cuda:0")
    cfg = UNetConfig.sdxl_base()
    net = MxUNet(cfg, synthetic_params(cfg, device=dev), device=dev)
    den = SDXLDenoiser(net)
    shared = {}
    rid = [0]

    def reqs(res, n):
        out = []
        for _ in range(n):
            out.append(synthetic_request(rid[0], res, 1000, cfg, den, dev, shared=shared)); rid[0] += 1
        return out
    head = {"1024": reqs(1024, 4)}
    t_head = timed(lambda: den.denoising_step(head))
    per_px2 = t_head / (4 * 1024 * 1024)                       # ms per pixel at the headline rate (attention's L^2 term aside)
    print(f"headline 4 x 1024: {t_head:.2f} ms/step")
    print(f"{'512/768/1024':>14s} {'alone ms (each)':>24s} {'serial':>8s} {'concurrent':>10s} {'ONE sequence':>12s} {'(sliced)':>9s} {'at headline rate':>16s}")
    for mix in ((1, 1, 1), (2, 2, 2), (4, 2, 1), (1, 2, 4), (4, 4, 4), (8, 0, 2)):
        batch = {str(r): reqs(r, n) for r, n in zip((512, 768, 1024), mix) if n}
        alone = [timed(lambda r=r: den.denoising_step({r: batch[r]})) for r in batch]
        net.mixed_one_sequence = False                          # the round-2 forms: one launch sequence per resolution
        den.concurrent_resolutions = False
        serial = timed(lambda: den.denoising_step(batch))
        den.concurrent_resolutions = True
        conc = timed(lambda: den.denoising_step(batch))
        net.mixed_one_sequence = True                           # round 3: all resolutions in ONE launch sequence (grouped launches)
        one = timed(lambda: den.denoising_step(batch))
        one_sliced = timed(lambda: den.denoising_step(batch, is_sliced=True, patch_size=256))
        ideal = per_px2 * sum(n * r * r for r, n in zip((512, 768, 1024), mix))
        print(f"{'/'.join(map(str, mix)):>14s} {' '.join(f'{a:7.2f}' for a in alone):>24s} {serial:8.2f} {conc:10.2f} {one:12.2f} {one_sliced:9.2f} {ideal:16.2f}", flush=True)


if __name__ == "__main__":
    main()
