"""GPU parity of the VAE decoder step plan (mx_vae_decode) against the CPU oracle (oracle/vae_ref.py), seeded bf16-representable weights
shared by both sides.  The reference decodes in fp32 (force_upcast); here activations are stored in bf16 between fused kernels, so the
bound is the UNet tests' one: max error <= 4 % of the output range, relative L2 <= 2 %."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import vae_ref as ref  # noqa: E402  (checker only)


def _check(got, want, what, max_rel=0.04, l2_rel=0.02):
    got = got.float().cpu()
    assert torch.isfinite(got).all(), f"{what}: non-finite"
    scale = want.abs().max().item()
    err = (got - want).abs().max().item()
    l2 = ((got - want).norm() / want.norm()).item()
    print(f"{what}: max err {err:.4f} ({err / scale:.4f} of range), rel L2 {l2:.4f}")
    assert err <= max_rel * scale and l2 <= l2_rel, f"{what}: max err {err} (range {scale}), rel L2 {l2}"


@pytest.mark.parametrize("batch,hw", [(2, 16), (1, 32)])
def test_vae_decode_tiny(cuda_device, batch, hw):
    from sduss_amd.vae import MxVAEDecoder, VAEConfig
    ocfg = ref.VAEConfig.tiny()
    P = ref.init_params(ocfg)
    g = torch.Generator().manual_seed(hw)
    lat = (torch.randn(batch, 4, hw, hw, generator=g) * 0.8).to(torch.bfloat16)
    want = ref.decode(P, ocfg, lat.float())
    vae = MxVAEDecoder(VAEConfig.tiny(), P, device="cuda:0")
    got = vae.decode(lat.cuda())
    assert got.shape == want.shape and got.dtype == torch.float32
    _check(got, want, f"vae decode tiny b{batch} {hw}x{hw}")


def test_vae_decode_sd3_latents(cuda_device):
    """the SD3 form: 16 latent channels, latents / scaling_factor + shift_factor, no post_quant_conv (pipeline_stable_diffusion_3_esymred.py:408)"""
    from sduss_amd.vae import MxVAEDecoder, VAEConfig
    from dataclasses import replace
    ocfg = ref.VAEConfig.tiny_sd3()
    P = ref.init_params(ocfg)
    lat = (torch.randn(2, 16, 16, 16, generator=torch.Generator().manual_seed(8)) * 1.5).to(torch.bfloat16)
    want = ref.decode(P, ocfg, lat.float())
    cfg = replace(VAEConfig.sd3(), block_out_channels=(64, 64, 128), layers_per_block=1)
    got = MxVAEDecoder(cfg, P, device="cuda:0").decode(lat.cuda())
    _check(got, want, "vae decode tiny SD3 (16 ch, shift)")


def test_vae_decode_sdxl_widths(cuda_device):
    """the real SDXL VAE widths (128 / 256 / 512 / 512, three resnets per up block, 512-wide single-head attention) on a 32 x 32 latent
    (256 px image): exercises the GEMM -> softmax -> GEMM attention at L = 1024 and every conv width"""
    from sduss_amd.vae import MxVAEDecoder, VAEConfig
    ocfg = ref.VAEConfig.sdxl()
    P = ref.init_params(ocfg)
    lat = (torch.randn(1, 4, 32, 32, generator=torch.Generator().manual_seed(3)) * 0.8).to(torch.bfloat16)
    with torch.inference_mode():
        want = ref.decode(P, ocfg, lat.float())
    vae = MxVAEDecoder(VAEConfig.sdxl(), P, device="cuda:0")
    got = vae.decode(lat.cuda())
    _check(got, want, "vae decode SDXL widths 32x32")


def test_post_inference_images(cuda_device):
    from types import SimpleNamespace
    from sduss_amd.vae import MxVAEDecoder, VAEConfig, post_inference
    ocfg = ref.VAEConfig.tiny()
    P = ref.init_params(ocfg)
    vae = MxVAEDecoder(VAEConfig.tiny(), P, device="cuda:0")
    reqs = [SimpleNamespace(latents=(torch.randn(1, 4, 16, 16, generator=torch.Generator().manual_seed(i)) * 0.8).to(torch.bfloat16).cuda()) for i in range(3)]
    out = post_inference(vae, {"64": reqs})
    want = ref.postprocess(ref.decode(P, ocfg, torch.cat([r.latents for r in reqs]).float().cpu()))
    assert out["64"].shape == want.shape and float(out["64"].min()) >= 0.0 and float(out["64"].max()) <= 1.0
    assert (out["64"].float().cpu() - want).abs().max().item() <= 0.03
