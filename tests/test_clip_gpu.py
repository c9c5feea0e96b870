"""GPU parity of the CLIP text-encoder step plan (mx_clip_encode) against transformers' own CLIPTextModel / CLIPTextModelWithProjection on CPU
in fp32 -- the library the reference's encode_prompt calls (a third-party dependency present in this image, so this checker is the real
implementation, not a restatement).  Random-init weights rounded to bf16 on both sides; bf16 activation storage on the device: max error
<= 3 % of the output range for the hidden states, 3 % for the pooled embedding."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _model(cfg, seed):
    from transformers import CLIPTextConfig, CLIPTextModel, CLIPTextModelWithProjection
    torch.manual_seed(seed)
    hf = CLIPTextConfig(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, intermediate_size=cfg.intermediate_size,
                        num_hidden_layers=cfg.num_hidden_layers, num_attention_heads=cfg.num_attention_heads,
                        max_position_embeddings=cfg.max_position_embeddings, hidden_act=cfg.hidden_act, projection_dim=max(cfg.projection_dim, 64),
                        eos_token_id=cfg.eos_token_id, bos_token_id=0, pad_token_id=1, layer_norm_eps=cfg.layer_norm_eps)
    m = (CLIPTextModelWithProjection if cfg.projection_dim > 0 else CLIPTextModel)(hf).eval()
    with torch.no_grad():
        for p in m.parameters():                               # livelier than the default init, and bf16-representable on both sides
            if p.ndim == 2:
                p.copy_((torch.randn_like(p) * p.shape[1] ** -0.5 * 1.5).to(torch.bfloat16).float())
            else:
                p.copy_((p + 0.1 * torch.randn_like(p)).to(torch.bfloat16).float())
    return m


def _ids(cfg, b, seed, eos):
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(3, cfg.vocab_size - 2, (b, cfg.max_position_embeddings), generator=g)
    ids[:, 0] = 0
    for i in range(b):
        n = int(torch.randint(4, cfg.max_position_embeddings - 1, (1,), generator=g))
        ids[i, n:] = eos                                       # EOS then padding with EOS (the SDXL tokenizers pad with EOS / 0)
    return ids


def _close(got, want, rel, what):
    got = got.float().cpu()
    assert torch.isfinite(got).all(), f"{what}: non-finite"
    scale = want.abs().max().item()
    err = (got - want).abs().max().item()
    l2 = ((got - want).norm() / want.norm()).item()
    print(f"{what}: max err {err:.4f} of range {scale:.3f}, rel L2 {l2:.4f}")
    assert err <= rel * scale, f"{what}: max err {err} > {rel} * {scale}"


@pytest.mark.parametrize("act,proj,eos", [("quick_gelu", 0, 2), ("gelu", 64, 2), ("gelu", 64, 997)])
def test_clip_tiny_vs_transformers(cuda_device, act, proj, eos):
    """tiny encoder (3 layers, 2 heads of 64): hidden_states[-2], and with projection the pooled text_embeds -- both EOS conventions
    (legacy eos_token_id == 2: position of the largest id; otherwise the first EOS)"""
    from dataclasses import replace
    from sduss_amd.clip import CLIPTextConfig, MxCLIPTextEncoder
    cfg = replace(CLIPTextConfig.tiny(projection_dim=proj, hidden_act=act), eos_token_id=eos)
    m = _model(cfg, 5)
    ids = _ids(cfg, 3, 11, eos if eos != 2 else cfg.vocab_size - 1)
    with torch.no_grad():
        o = m(ids, output_hidden_states=True)
    enc = MxCLIPTextEncoder(cfg, m.state_dict(), device="cuda:0")
    hidden, pooled = enc.encode(ids)
    _close(hidden, o.hidden_states[-2], 0.03, f"clip tiny {act} hidden_states[-2]")
    if proj:
        _close(pooled, o.text_embeds, 0.03, f"clip tiny {act} text_embeds (eos {eos})")
    else:
        assert pooled is None


def test_clip_causal_mask_matters(cuda_device):
    """a token change at position t must leave the hidden states of positions < t unchanged (and change those >= t)"""
    from sduss_amd.clip import CLIPTextConfig, MxCLIPTextEncoder
    cfg = CLIPTextConfig.tiny()
    m = _model(cfg, 7)
    enc = MxCLIPTextEncoder(cfg, m.state_dict(), device="cuda:0")
    ids = _ids(cfg, 1, 3, cfg.vocab_size - 1)
    h0, _ = enc.encode(ids)
    ids2 = ids.clone(); ids2[0, 40] = (ids2[0, 40] + 17) % (cfg.vocab_size - 4) + 3
    h1, _ = enc.encode(ids2)
    assert torch.equal(h0[:, :40], h1[:, :40])
    assert not torch.equal(h0[:, 40:], h1[:, 40:])


def test_clip_sdxl_text_encoders(cuda_device):
    """the two SDXL text encoders at full size (ViT-L: 12 layers x 768, quick_gelu; bigG: 32 layers x 1280, gelu, projection), batch 2:
    prompt_embeds = cat(hidden_states[-2]) [2, 77, 2048] and pooled_prompt_embeds [2, 1280] as encode_prompt assembles them"""
    from sduss_amd.clip import CLIPTextConfig, MxCLIPTextEncoder, encode_prompt_sdxl
    c1, c2 = CLIPTextConfig.sdxl_text_encoder(), CLIPTextConfig.sdxl_text_encoder_2()
    m1, m2 = _model(c1, 1), _model(c2, 2)
    ids = _ids(c1, 2, 9, c1.vocab_size - 1)
    with torch.no_grad():
        o1 = m1(ids, output_hidden_states=True); o2 = m2(ids, output_hidden_states=True)
    e1 = MxCLIPTextEncoder(c1, m1.state_dict(), device="cuda:0"); e2 = MxCLIPTextEncoder(c2, m2.state_dict(), device="cuda:0")
    embeds, pooled = encode_prompt_sdxl(e1, e2, ids, ids)
    assert embeds.shape == (2, 77, 2048) and pooled.shape == (2, 1280)
    _close(embeds[..., :768], o1.hidden_states[-2], 0.03, "SDXL text_encoder hidden_states[-2]")
    _close(embeds[..., 768:], o2.hidden_states[-2], 0.04, "SDXL text_encoder_2 hidden_states[-2]")
    _close(pooled, o2.text_embeds, 0.04, "SDXL pooled_prompt_embeds")
