"""GPU parity of the T5 v1.1 encoder step plan (mx_t5_encode) against transformers' own T5EncoderModel on CPU in fp32 -- the library the
reference's SD3 encode_prompt calls for text_encoder_3 (present in this image: the checker is the real implementation).  Random-init weights
rounded to bf16 on both sides, no attention mask (as _get_t5_prompt_embeds calls it); bf16 activation storage on the device."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _model(cfg, seed, scale=1.0):
    from transformers import T5Config as HF, T5EncoderModel
    torch.manual_seed(seed)
    m = T5EncoderModel(HF(vocab_size=cfg.vocab_size, d_model=cfg.d_model, d_kv=64, d_ff=cfg.d_ff, num_layers=cfg.num_layers, num_heads=cfg.num_heads,
                          relative_attention_num_buckets=cfg.relative_attention_num_buckets, relative_attention_max_distance=cfg.relative_attention_max_distance,
                          feed_forward_proj="gated-gelu", layer_norm_epsilon=cfg.layer_norm_epsilon, dropout_rate=0.0)).eval()
    with torch.no_grad():
        for n, p in m.named_parameters():
            if "relative_attention_bias" in n:
                p.copy_((torch.randn_like(p) * 2.0).to(torch.bfloat16).float())          # a bias that matters
            elif p.ndim == 2 and "embed" not in n and "shared" not in n:
                p.copy_((torch.randn_like(p) * p.shape[1] ** -0.5 * scale).to(torch.bfloat16).float())
            else:
                p.copy_(p.to(torch.bfloat16).float())
    return m


def _close(got, want, rel, what):
    got = got.float().cpu()
    assert torch.isfinite(got).all(), f"{what}: non-finite"
    scale = want.abs().max().item()
    err = (got - want).abs().max().item()
    l2 = ((got - want).norm() / want.norm()).item()
    print(f"{what}: max err {err:.4f} of range {scale:.3f}, rel L2 {l2:.4f}")
    assert err <= rel * scale, f"{what}: max err {err} > {rel} * {scale}"


@pytest.mark.parametrize("L,b", [(256, 2), (64, 3), (200, 1)])
def test_t5_tiny_vs_transformers(cuda_device, L, b):
    from sduss_amd.t5 import MxT5Encoder, T5Config
    cfg = T5Config.tiny()
    m = _model(cfg, 3, scale=0.5)              # q . k is NOT scaled by 1/sqrt(d) in T5: unit-variance projections would make 64-term logits of
    ids = torch.randint(0, cfg.vocab_size, (b, L), generator=torch.Generator().manual_seed(L))
    with torch.no_grad():                      # std 8 and a near one-hot softmax that amplifies every bf16 rounding of q and k
        want = m(ids)[0]
    got = MxT5Encoder(cfg, m.state_dict(), device="cuda:0", seq_lens=(L,)).encode(ids)
    _close(got, want, 0.03, f"t5 tiny L{L} b{b}")


def test_t5_xxl_widths(cuda_device):
    """the XXL widths of SD3's text_encoder_3 (d_model 4096, 64 heads, d_ff 10240) with two blocks, 256 tokens, batch 2"""
    from dataclasses import replace
    from sduss_amd.t5 import MxT5Encoder, T5Config
    cfg = replace(T5Config.xxl(), num_layers=2, vocab_size=1000)
    m = _model(cfg, 4)
    ids = torch.randint(0, cfg.vocab_size, (2, 256), generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        want = m(ids)[0]
    got = MxT5Encoder(cfg, m.state_dict(), device="cuda:0").encode(ids)
    _close(got, want, 0.03, "t5 XXL widths, 2 blocks")


def test_attention_with_bias_and_rmsnorm_ops(cuda_device):
    """the two new ops on their own: softmax(q k^T + bias) v through mx_attention_prescaled_bias (ragged key count, bias that dominates the
    scores) and mx_rmsnorm, against torch"""
    import ctypes as C
    import math
    from sduss_amd import lib, ops
    l = lib.load()
    g = torch.Generator().manual_seed(2)
    b, h, lq = 2, 3, 200
    c = h * 64
    rt = lambda t: t.to(torch.bfloat16).float()
    q = rt(torch.randn(b, lq, c, generator=g)); k = rt(torch.randn(b, lq, c, generator=g)); v = rt(torch.randn(b, lq, c, generator=g))
    bias = torch.randn(h, lq, lq, generator=g) * 3.0
    sp = lambda t: t.reshape(b, lq, h, 64).transpose(1, 2)
    want = torch.softmax(sp(q) @ sp(k).transpose(-1, -2) + bias[None], dim=-1) @ sp(v)
    want = want.transpose(1, 2).reshape(b, lq, c)
    LOG2E = 1.4426950408889634
    ldb = (lq + 63) // 64 * 64
    bias_d = torch.full((h, lq, ldb), float("nan")); bias_d[:, :, :lq] = bias * LOG2E
    qs = (q * LOG2E).to(torch.bfloat16).reshape(-1, c).cuda(); kd = k.to(torch.bfloat16).reshape(-1, c).cuda()
    vt = ops.pack_vt(v, pad=float("nan")).to(torch.bfloat16).cuda(); bd = bias_d.cuda()
    o = torch.empty_like(qs)
    lib.check(l.mx_attention_prescaled_bias(lib.current_stream(), qs.data_ptr(), c, kd.data_ptr(), c, vt.data_ptr(), vt.shape[2], vt.shape[1] * vt.shape[2],
                                            o.data_ptr(), c, b, h, lq, lq, bd.data_ptr(), ldb), "attention bias")
    _close(o.reshape(b, lq, c), want, 2.0 ** -6, "attention with bias")
    x = rt(torch.randn(300, 4096, generator=g) * 2 + 0.5); w = 1 + 0.2 * torch.randn(4096, generator=g)
    y = torch.empty(300, 4096, dtype=torch.bfloat16, device="cuda")
    xd = x.to(torch.bfloat16).cuda(); wd = w.cuda()
    lib.check(l.mx_rmsnorm(lib.current_stream(), xd.data_ptr(), y.data_ptr(), wd.data_ptr(), 300, 4096, 1e-6), "rmsnorm")
    _close(y, x * torch.rsqrt((x * x).mean(dim=1, keepdim=True) + 1e-6) * w, 2.0 ** -7, "rmsnorm rows")
