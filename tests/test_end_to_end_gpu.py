"""prepare_inference -> denoising_step x N -> post_inference on the device, small configurations: the three stages the reference's SDXL pipeline
runs per request (pipeline_stable_diffusion_xl_esymred.py:55-256, 259-403, 406-463) connected through the same tensors -- prompt_embeds
[n, 77, 2 x hidden], pooled text_embeds, latents, images -- with every stage checked against its checker on the way (transformers' CLIP,
the UNet oracle chain, the VAE oracle)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import scheduler_ref, sdxl_unet_ref as ref, vae_ref  # noqa: E402  (checker only)


def test_text_to_image_stages_connect(cuda_device):
    from dataclasses import replace
    from transformers import CLIPTextConfig as HFCfg, CLIPTextModel, CLIPTextModelWithProjection
    from sduss_amd.clip import CLIPTextConfig, MxCLIPTextEncoder, encode_prompt_sdxl
    from sduss_amd.config import UNetConfig
    from sduss_amd.pipeline import Request, SDXLDenoiser
    from sduss_amd.unet import MxUNet
    from sduss_amd.vae import MxVAEDecoder, VAEConfig
    from sduss_amd.weights import synthetic_params
    dev = "cuda:0"
    ucfg = UNetConfig.tiny()                                   # cross_attention_dim 128 = 2 x 64, text_embed_dim 64
    c1 = replace(CLIPTextConfig.tiny(), hidden_size=64, intermediate_size=128, num_attention_heads=1)
    c2 = replace(CLIPTextConfig.tiny(projection_dim=ucfg.text_embed_dim, hidden_act="gelu"), hidden_size=64, intermediate_size=128, num_attention_heads=1)

    def hf(c, cls, seed):
        torch.manual_seed(seed)
        m = cls(HFCfg(vocab_size=c.vocab_size, hidden_size=c.hidden_size, intermediate_size=c.intermediate_size, num_hidden_layers=c.num_hidden_layers,
                      num_attention_heads=c.num_attention_heads, max_position_embeddings=77, hidden_act=c.hidden_act, projection_dim=max(c.projection_dim, 64),
                      eos_token_id=2, bos_token_id=0, pad_token_id=1)).eval()
        with torch.no_grad():
            for p in m.parameters():
                p.copy_((p if p.ndim == 1 else torch.randn_like(p) * p.shape[1] ** -0.5).to(torch.bfloat16).float())
        return m
    m1, m2 = hf(c1, CLIPTextModel, 1), hf(c2, CLIPTextModelWithProjection, 2)
    g = torch.Generator().manual_seed(4)
    ids = torch.randint(3, 990, (2, 77), generator=g); ids[:, 0] = 0; ids[0, 9:] = 999; ids[1, 30:] = 999      # row 0: prompt, row 1: negative prompt
    # ---- prepare_inference: text encoders ----
    e1, e2 = MxCLIPTextEncoder(c1, m1.state_dict(), dev), MxCLIPTextEncoder(c2, m2.state_dict(), dev)
    embeds, pooled = encode_prompt_sdxl(e1, e2, ids, ids)
    with torch.no_grad():
        o1, o2 = m1(ids, output_hidden_states=True), m2(ids, output_hidden_states=True)
    want_embeds = torch.cat([o1.hidden_states[-2], o2.hidden_states[-2]], dim=-1)
    assert embeds.shape == (2, 77, ucfg.cross_attention_dim) and pooled.shape == (2, ucfg.text_embed_dim)
    assert (embeds.float().cpu() - want_embeds).abs().max().item() <= 0.03 * want_embeds.abs().max().item()
    # ---- denoising loop: 4 Euler steps of one 256 px request under CFG ----
    P = synthetic_params(ucfg)
    den = SDXLDenoiser(MxUNet(ucfg, P, device=dev), guidance_scale=5.0)
    steps, res = 4, 256
    lat0 = torch.randn(1, 4, res // 8, res // 8, generator=g) * den.init_noise_sigma(steps)
    tid = torch.tensor([[float(res), float(res), 0.0, 0.0, float(res), float(res)]], device=dev)
    bf = torch.bfloat16
    req = Request(0, res, steps, lat0.to(dev, bf), embeds[0:1], embeds[1:2], pooled[0:1].to(bf), pooled[1:2].to(bf), tid, tid.clone())
    den.set_timesteps(req)
    for _ in range(steps):
        den.denoising_step({str(res): [req]})
    assert req.done() and torch.isfinite(req.latents.float()).all()
    # the same loop on the oracle, from the device's own conditioning (so this bound is the loop's, not the encoders')
    ocfg = ref.UNetConfig.tiny()
    lat = lat0.to(bf).float()
    ts, sig, _ = scheduler_ref.sdxl_euler_tables(steps)
    pe, ne = embeds[0:1].float().cpu(), embeds[1:2].float().cpu()
    pp, npp = pooled[0:1].to(bf).float().cpu(), pooled[1:2].to(bf).float().cpu()
    for i in range(steps):
        x2 = scheduler_ref.scale_model_input(torch.cat([lat, lat]), sig[[i] * 2])
        noise = ref.unet_forward(P, ocfg, x2, ts[[i] * 2], torch.cat([ne, pe]), torch.cat([npp, pp]), tid.cpu().repeat(2, 1))
        lat = scheduler_ref.euler_step(scheduler_ref.cfg_combine(noise, 5.0), lat, sig[[i]], sig[[i + 1]])
        lat = lat.to(bf).float()
    err = (req.latents.float().cpu() - lat).abs().max().item() / lat.abs().max().item()
    print(f"4-step loop vs oracle: {err:.4f} of range")
    assert err <= 0.10
    # ---- post_inference: VAE decode of the final latents ----
    vcfg = vae_ref.VAEConfig.tiny()
    VP = vae_ref.init_params(vcfg)
    img = MxVAEDecoder(VAEConfig.tiny(), VP, device=dev).decode(req.latents)
    want_img = vae_ref.decode(VP, vcfg, req.latents.float().cpu())
    assert img.shape == (1, 3, res // 2, res // 2)             # the tiny VAE upsamples 4x (three levels)
    assert (img.float().cpu() - want_img).abs().max().item() <= 0.04 * want_img.abs().max().item()
