"""GPU diagnostic (not a test): per-stage comparison of the MMDiT step plan with the oracle's trace (tiny config).
Usage on the GPU box: python tools/gpu_diag_mmdit.py > gpurun_out/diag_mmdit.log"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import sd3_mmdit_ref as ref  # noqa: E402
from sduss_amd.config import MMDiTConfig  # noqa: E402
from sduss_amd.transformer_sd3 import MxSD3Transformer  # noqa: E402


def main():
    ocfg = ref.MMDiTConfig.tiny()
    P = ref.init_params(ocfg)
    lat, t, e, p = ref.make_inputs(ocfg, 2, 16, ctx_len=37)
    trace = {}
    want = ref.mmdit_forward(P, ocfg, lat, t, e, p, trace=trace)
    net = MxSD3Transformer(MMDiTConfig.tiny(), P)
    args = (lat.cuda().to(torch.bfloat16), t.cuda(), e.cuda(), p.cuda())
    for name, tw in trace.items():
        if name == "temb":
            continue
        try:
            got = net.forward_one(*args, stage=name, stage_shape=tuple(tw.reshape(-1, tw.shape[-1]).shape))
            torch.cuda.synchronize()
            g = got.float().cpu().reshape(tw.shape)
            err = (g - tw).abs().max().item()
            print(f"{name:40s} shape {tuple(tw.shape)} max|ref| {tw.abs().max().item():8.4f} err {err:8.5f} rel {err / (tw.abs().max().item() + 1e-9):.5f}")
        except Exception as ex:  # noqa
            print(f"{name:40s} FAILED: {ex}")
    got = net.forward_one(*args).float().cpu()
    err = (got - want).abs().max().item()
    print(f"FINAL err {err:.5f} rel {err / want.abs().max().item():.5f} finite {torch.isfinite(got).all().item()}")
    t0 = time.time()
    for _ in range(5):
        net.forward_one(*args)
    torch.cuda.synchronize()
    print(f"tiny mmdit forward {1e3 * (time.time() - t0) / 5:.2f} ms")


if __name__ == "__main__":
    main()
