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


def test_sd3_text_to_image_stages_connect(cuda_device):
    """the SD3 chain: CLIP-L + CLIP-G (both with projection) + T5 -> prompt_embeds [n, 77 + 64, width] / pooled [n, 2 x proj] -> 3 flow-match
    steps of one 128 px request under CFG -> SD3-form VAE decode (16 latent channels, shift factor); tiny configurations, every stage against
    its checker (transformers' CLIP / T5, the MMDiT oracle chain, the VAE oracle)."""
    from dataclasses import replace
    from transformers import CLIPTextConfig as HFClip, CLIPTextModelWithProjection, T5Config as HFT5, T5EncoderModel
    from oracle import sd3_mmdit_ref as mref
    from sduss_amd.clip import CLIPTextConfig, MxCLIPTextEncoder
    from sduss_amd.config import MMDiTConfig
    from sduss_amd.pipeline_sd3 import SD3Denoiser, SD3Request, flow_match_tables
    from sduss_amd.t5 import MxT5Encoder, T5Config, encode_prompt_sd3
    from sduss_amd.transformer_sd3 import MxSD3Transformer
    from sduss_amd.vae import MxVAEDecoder, VAEConfig
    from sduss_amd.weights import synthetic_mmdit_params
    dev = "cuda:0"
    bf = torch.bfloat16
    mcfg = MMDiTConfig.tiny()                                  # joint_attention_dim 128, pooled_projection_dim 64
    cl = replace(CLIPTextConfig.tiny(projection_dim=32, hidden_act="quick_gelu"), hidden_size=64, intermediate_size=128, num_attention_heads=1)
    cg = replace(CLIPTextConfig.tiny(projection_dim=32, hidden_act="gelu"), hidden_size=64, intermediate_size=128, num_attention_heads=1)
    tcfg = T5Config.tiny()                                     # d_model 128 = the joint width

    def clip(c, seed):
        torch.manual_seed(seed)
        m = CLIPTextModelWithProjection(HFClip(vocab_size=c.vocab_size, hidden_size=64, intermediate_size=128, num_hidden_layers=c.num_hidden_layers,
                                               num_attention_heads=1, max_position_embeddings=77, hidden_act=c.hidden_act, projection_dim=32, eos_token_id=2,
                                               bos_token_id=0, pad_token_id=1)).eval()
        with torch.no_grad():
            for p in m.parameters():
                p.copy_((p if p.ndim == 1 else torch.randn_like(p) * p.shape[1] ** -0.5).to(bf).float())
        return m
    ml, mg = clip(cl, 1), clip(cg, 2)
    torch.manual_seed(3)
    mt = T5EncoderModel(HFT5(vocab_size=tcfg.vocab_size, d_model=128, d_kv=64, d_ff=256, num_layers=3, num_heads=2, feed_forward_proj="gated-gelu",
                             layer_norm_epsilon=1e-6, dropout_rate=0.0)).eval()
    with torch.no_grad():
        for n_, p in mt.named_parameters():
            if p.ndim == 2 and "embed" not in n_ and "shared" not in n_ and "relative" not in n_:
                p.copy_((torch.randn_like(p) * p.shape[1] ** -0.5 * 0.5).to(bf).float())
            else:
                p.copy_(p.to(bf).float())
    g = torch.Generator().manual_seed(6)
    ids = torch.randint(3, 990, (2, 77), generator=g); ids[:, 0] = 0; ids[0, 12:] = 999; ids[1, 40:] = 999
    ids5 = torch.randint(0, tcfg.vocab_size, (2, 64), generator=g)
    embeds, pooled = encode_prompt_sd3(MxCLIPTextEncoder(cl, ml.state_dict(), dev), MxCLIPTextEncoder(cg, mg.state_dict(), dev),
                                       MxT5Encoder(tcfg, mt.state_dict(), dev, seq_lens=(64,)), ids, ids, ids5)
    with torch.no_grad():
        ol, og, ot = ml(ids, output_hidden_states=True), mg(ids, output_hidden_states=True), mt(ids5)[0]
    want = torch.cat([torch.cat([ol.hidden_states[-2], og.hidden_states[-2]], dim=-1), ot], dim=-2)
    assert embeds.shape == (2, 77 + 64, mcfg.joint_attention_dim) and pooled.shape == (2, mcfg.pooled_projection_dim)
    assert (embeds.float().cpu() - want).abs().max().item() <= 0.03 * want.abs().max().item()
    assert (pooled.cpu() - torch.cat([ol.text_embeds, og.text_embeds], dim=-1)).abs().max().item() <= 0.03 * og.text_embeds.abs().max().item()
    # ---- 3 flow-match steps under CFG (row 0 = prompt, row 1 = negative prompt) ----
    P = synthetic_mmdit_params(mcfg)
    den = SD3Denoiser(MxSD3Transformer(mcfg, P, device=dev), guidance_scale=7.0)
    steps, res = 3, 128
    lat0 = torch.randn(1, mcfg.in_channels, res // 8, res // 8, generator=g).to(bf)
    req = SD3Request(0, res, steps, lat0.to(dev), embeds[0:1], embeds[1:2], pooled[0:1].to(bf), pooled[1:2].to(bf))
    den.set_timesteps(req)
    for _ in range(steps):
        den.denoising_step({str(res): [req]})
    ocfg = mref.MMDiTConfig.tiny()
    ts, sig = flow_match_tables(steps)
    ts, sig = torch.from_numpy(ts), torch.from_numpy(sig)
    lat = lat0.float()
    pe, ne = embeds[0:1].float().cpu(), embeds[1:2].float().cpu()
    pp, npp = pooled[0:1].to(bf).float().cpu(), pooled[1:2].to(bf).float().cpu()
    for i in range(steps):
        v = mref.mmdit_forward(P, ocfg, torch.cat([lat, lat]), ts[[i] * 2], torch.cat([ne, pe]), torch.cat([npp, pp]))
        lat = scheduler_ref.flow_match_step(scheduler_ref.cfg_combine(v, 7.0), lat, sig[[i]], sig[[i + 1]]).to(bf).float()
    err = (req.latents.float().cpu() - lat).abs().max().item() / lat.abs().max().item()
    print(f"SD3 3-step loop vs oracle: {err:.4f} of range")
    assert req.done() and err <= 0.10
    # ---- post_inference: latents / scaling + shift -> decode ----
    vo = vae_ref.VAEConfig.tiny_sd3()
    VP = vae_ref.init_params(vo)
    img = MxVAEDecoder(replace(VAEConfig.sd3(), block_out_channels=(64, 64, 128), layers_per_block=1), VP, device=dev).decode(req.latents)
    want_img = vae_ref.decode(VP, vo, req.latents.float().cpu())
    assert (img.float().cpu() - want_img).abs().max().item() <= 0.04 * want_img.abs().max().item()
