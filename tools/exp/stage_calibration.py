"""Diagnostic: per-stage error of the HIP plan and of a STOCK bf16 evaluation of the oracle's graph, both against the fp32 oracle (GPU evaluation), SDXL-base width.
Finds the stage at which the HIP path starts to lose more than stock bf16 does.  Usage on the GPU box: python tools/exp/stage_calibration.py [batch hw row]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import sdxl_unet_ref as ref  # noqa: E402  (diagnostic tool, not the product path)
from sduss_amd.config import UNetConfig  # noqa: E402
from sduss_amd.unet import MxUNet  # noqa: E402


def main():
    batch, hw, row = (int(x) for x in (sys.argv[1:4] + ["8", "32", "1"][len(sys.argv) - 1:]))
    ocfg = ref.UNetConfig.sdxl_base()
    P = ref.fast_params(ocfg)
    net = MxUNet(UNetConfig.sdxl_base(), P, device="cuda:0")
    s, t, e, te, ti = ref.make_inputs(ocfg, batch, hw)
    s = s + 0.3 * torch.randn(s.shape, generator=torch.Generator().manual_seed(8))
    t32, t16 = {}, {}
    P32 = {k: v.float().cuda() for k, v in P.items()}
    P16 = {k: v.to(torch.bfloat16).cuda() for k, v in P.items()}
    with torch.inference_mode():
        ref.unet_forward(P32, ocfg, s, t, e, te, ti, trace=t32, device="cuda")
        ref.unet_forward(P16, ocfg, s, t, e, te, ti, trace=t16, compute_dtype=torch.bfloat16, device="cuda", sdpa=True)
    x = s.cuda().to(torch.bfloat16)
    print(f"batch {batch}, {hw}x{hw} latents, row {row}: rel L2 vs the fp32 oracle per stage")
    for name, want in t32.items():
        if want.ndim != 4:
            continue
        b, c, h, w = want.shape
        got = net.forward_one(x, t.cuda(), e.cuda(), te.cuda(), ti.cuda(), stage=name, stage_shape=(b, h, w, c)).float().permute(0, 3, 1, 2)
        wr = want[row].float()
        l2 = ((got[row] - wr).norm() / wr.norm()).item()
        sl2 = ((t16[name][row].float() - wr).norm() / wr.norm()).item()
        print(f"  {name:40s} HIP {l2:.4f}  stock bf16 {sl2:.4f}  ratio {l2 / max(sl2, 1e-9):5.2f}")
        last_name, last_hip = name, got
    # the final stage (conv_norm_out + SiLU + conv_out) in isolation: the fp32 oracle's last stage applied to each path's OWN last-resnet output
    import torch.nn.functional as F
    with torch.inference_mode():
        o32 = ref.unet_forward(P32, ocfg, s, t, e, te, ti, device="cuda")
        b16 = ref.unet_forward(P16, ocfg, s, t, e, te, ti, compute_dtype=torch.bfloat16, device="cuda", sdpa=True).float()
        hip = net.forward_one(x, t.cuda(), e.cuda(), te.cuda(), ti.cuda()).float()

        def tail32(xin):
            y = F.silu(F.group_norm(xin.float(), ocfg.norm_num_groups, P32["conv_norm_out.weight"], P32["conv_norm_out.bias"], ocfg.norm_eps))
            return F.conv2d(y, P32["conv_out.weight"], P32["conv_out.bias"], padding=1)
        exp_hip, exp_b16 = tail32(last_hip), tail32(t16[last_name])
    n = lambda a, b: ((a[row] - b[row]).norm() / b[row].norm()).item()
    print(f"  FINAL OUTPUT                             HIP {n(hip, o32):.4f}  stock bf16 {n(b16, o32):.4f}")
    print(f"  final stage alone (own input -> fp32 tail): HIP {n(hip, exp_hip):.4f}  stock bf16 {n(b16, exp_b16):.4f}")
    print(f"  fp32 tail of own last-resnet output vs oracle: HIP {n(exp_hip, o32):.4f}  stock bf16 {n(exp_b16, o32):.4f}   (how the last stage amplifies the incoming error)")
    from sduss_amd.weights import params_as_held
    held = params_as_held(UNetConfig.sdxl_base(), P)
    with torch.inference_mode():
        h32 = ref.unet_forward({k: v.float().cuda() for k, v in held.items()}, ocfg, s, t, e, te, ti, device="cuda")
        h16 = ref.unet_forward(held, ocfg, s, t, e, te, ti, compute_dtype=torch.bfloat16, device="cuda", sdpa=True).float()
    for r in range(batch):
        nn = lambda a, b: ((a[r] - b[r]).norm() / b[r].norm()).item()
        print(f"  row {r}: vs the oracle on the ORIGINAL weights: HIP {nn(hip, o32):.4f} stock bf16 {nn(b16, o32):.4f} | vs the oracle on the weights AS HELD (W * gamma rounded once): "
              f"HIP {nn(hip, h32):.4f} stock bf16 on those weights {nn(h16, h32):.4f} ratio {nn(hip, h32) / nn(h16, h32):.2f}")
    print(f"  output range {o32[row].abs().max().item():.3f}, rms {o32[row].pow(2).mean().sqrt().item():.4f}; last-resnet rms {t32[last_name][row].pow(2).mean().sqrt().item():.3f}")


if __name__ == "__main__":
    main()
