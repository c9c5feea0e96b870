"""GPU diagnostic (not a test): per-stage comparison of the HIP step plan with the oracle's trace on the tiny config.
Usage on the GPU box: python tools/gpu_diag.py > gpurun_out/diag.log"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import sdxl_unet_ref as ref  # noqa: E402
from sduss_amd.config import UNetConfig  # noqa: E402
from sduss_amd.unet import MxUNet  # noqa: E402


def main():
    ocfg = ref.UNetConfig.tiny()
    P = ref.init_params(ocfg)
    b, hw = 2, 32
    sample, t, ehs, text, tids = ref.make_inputs(ocfg, b, hw)
    for gn_patch, corners in ((0, False), (16, True)):
        trace = {}
        want = ref.unet_forward(P, ocfg, sample, t, ehs, text, tids, gn_patch=gn_patch or None, trace=trace,
                                sliced_corners=corners)
        net = MxUNet(UNetConfig.tiny(), P)
        args = (sample.cuda().to(torch.bfloat16), t.cuda(), ehs.cuda(), text.cuda(), tids.cuda())
        print(f"== gn_patch={gn_patch}")
        for name, tw in trace.items():
            n, c, h, w = tw.shape
            try:
                got = net.forward_one(*args, gn_patch=gn_patch, stage=name, stage_shape=(n, h, w, c))
                torch.cuda.synchronize()
                g = got.float().cpu().permute(0, 3, 1, 2)
                err = (g - tw).abs().max().item()
                print(f"{name:40s} shape {tuple(tw.shape)} max|ref| {tw.abs().max().item():8.4f} err {err:8.5f} rel {err / (tw.abs().max().item() + 1e-9):.5f}")
            except Exception as e:  # noqa
                print(f"{name:40s} FAILED: {e}")
        got = net.forward_one(*args, gn_patch=gn_patch).float().cpu()
        err = (got - want).abs().max().item()
        print(f"FINAL err {err:.5f} rel {err / want.abs().max().item():.5f} finite {torch.isfinite(got).all().item()}")
        t0 = time.time()
        for _ in range(5):
            net.forward_one(*args, gn_patch=gn_patch)
        torch.cuda.synchronize()
        print(f"tiny forward {1e3 * (time.time() - t0) / 5:.2f} ms")


if __name__ == "__main__":
    main()
