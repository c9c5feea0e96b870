"""mx_attn_tail: the attention tail of a BasicTransformerBlock (attn1.to_out + residual -> norm2 folded into attn2.to_q -> the 77-key cross-attention ->
attn2.to_out + residual; modules/transformer.py:204-262, modules/attention.py:59-110) as ONE launch against the four separate launches on the SAME descriptors.

The chained launch runs the tiles of the same kernels in the same order of summation, so the bar is BIT equality -- of the new hidden state, of the
intermediate q2 / ao2, of the slab and finalised row statistics -- at the two step shapes of the headline batch, at a shape whose panel count is not a
multiple of the queue count, repeatedly (every launch must leave its counters zero), and under an uneven load on another stream (the hand-offs must not depend
on which workgroup runs where or when).  A torch fp32 evaluation of the same four ops bounds the arithmetic itself."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _rt(t):
    return t.to(torch.bfloat16).to(torch.float32)


def _problem(b, heads, L, ctx_len=77, seed=0):
    from sduss_amd import ops
    from sduss_amd.weights import fold_layernorm
    g = torch.Generator().manual_seed(seed + b + heads + L)
    c = heads * 64
    m = b * L
    ao = _rt(torch.randn(m, c, generator=g))
    y = _rt(torch.randn(m, c, generator=g) * (0.5 + torch.rand(m, 1, generator=g)) + 0.5 * torch.randn(m, 1, generator=g))
    lin = lambda: (_rt(torch.randn(c, c, generator=g) * c ** -0.5), 0.1 * torch.randn(c, generator=g))
    (w1, b1), (wq, bq), (w2, b2) = lin(), lin(), lin()
    gamma, beta = 1.0 + 0.2 * torch.randn(c, generator=g), 0.1 * torch.randn(c, generator=g)
    wqf, colsum, bqf = fold_layernorm(wq, bq, gamma, beta)
    k = _rt(torch.randn(b, ctx_len, c, generator=g))
    v = _rt(torch.randn(b, ctx_len, c, generator=g))
    dev = lambda t: t.cuda()
    bf = lambda t: t.to(torch.bfloat16).cuda()
    host = dict(ao=ao, y=y, w1=w1, b1=b1, wq=wq, bq=bq, gamma=gamma, beta=beta, w2=w2, b2=b2, k=k, v=v)
    args = dict(ao=bf(ao), y=bf(y), w1=bf(w1), b1=dev(b1), wq=wqf.cuda(), bq=dev(bqf), colsum_q=dev(colsum), k=bf(k.reshape(b * ctx_len, c)),
                vt=ops.pack_vt(bf(v), pad=float("nan")), w2=bf(w2), b2=dev(b2), heads=heads, L=L, ctx_len=ctx_len)
    return host, args


def _same(a, b, what):
    assert a.dtype == b.dtype and a.shape == b.shape
    eq = torch.equal(a, b) if a.dtype != torch.float32 else torch.equal(a.view(torch.int32), b.view(torch.int32))
    if not eq:
        d = (a.float() - b.float()).abs()
        raise AssertionError(f"{what}: chained launch differs from the separate launches: {int((d > 0).sum())} elements, max {float(d.max()):.3e}")


def _compare(args, finalise=True):
    from sduss_amd import ops
    ys, (sts, slabs), fs, q2s, ao2s, _ = ops.attn_tail(**args, chained=False, finalise=finalise)
    yc, (stc, slabc), fc, q2c, ao2c, sync = ops.attn_tail(**args, chained=True, finalise=finalise)
    assert ops.attn_tail_status(sync) == 0, "a wait inside the chained launch gave up"
    assert slabs == slabc
    _same(q2c, q2s, "q2"); _same(ao2c, ao2s, "ao2"); _same(yc, ys, "y")
    _same(stc[:, :slabs].contiguous(), sts[:, :slabs].contiguous(), "slab statistics")
    if finalise:
        _same(fc, fs, "finalised statistics")
    s = sync.clone()
    s[257] = 0                                   # (the error word is the one word a launch does not clear)
    assert int(s.abs().sum()) == 0, "the launch must leave its counters zero"
    return yc, sync


@pytest.mark.parametrize("b,heads,L", [(8, 20, 1024),      # the 60 layers at 32 x 32 of the headline batch: 32 panels x 8 tiles
                                       (8, 10, 4096),      # the 10 layers at 64 x 64: 128 panels x 4 tiles, two rounds per queue
                                       (9, 20, 768),       # 27 panels: queues of 4 and 3 panels, a short round
                                       (13, 20, 1024),     # 52 panels: queues of 7 and 6 panels, two rounds, the second short
                                       (19, 10, 768),      # 57 panels x 4 tiles: queues of 8 and 7 panels
                                       (7, 10, 4096)])     # 112 panels x 4 tiles: 14 per queue, a short second round
def test_attn_tail_equals_four_launches(cuda_device, b, heads, L):
    host, args = _problem(b, heads, L)
    y, _sync = _compare(args)
    # the arithmetic itself against torch fp32 (the bound of three bf16-stored linears and one attention: 2^-6 of the range)
    c = heads * 64
    y1 = host["ao"] @ host["w1"].t() + host["b1"] + host["y"]
    y1 = _rt(y1)
    q2 = F.layer_norm(y1, (c,), host["gamma"], host["beta"], 1e-5) @ host["wq"].t() + host["bq"]
    qh = q2.reshape(b, L, heads, 64).transpose(1, 2)
    kh = host["k"].reshape(b, -1, heads, 64).transpose(1, 2)
    vh = host["v"].reshape(b, -1, heads, 64).transpose(1, 2)
    a2 = F.scaled_dot_product_attention(qh, kh, vh).transpose(1, 2).reshape(b * L, c)
    want = a2 @ host["w2"].t() + host["b2"] + y1
    err = (y.float().cpu() - want).abs().max().item() / want.abs().max().item()
    print(f"attn_tail B{b} H{heads} L{L}: max err vs torch fp32 {err:.5f} of range")
    assert err <= 2.0 ** -6


def test_attn_tail_without_finalised_statistics(cuda_device):
    _host, args = _problem(16, 20, 512, seed=3)
    _compare(args, finalise=False)


def test_attn_tail_repeated_and_under_uneven_load(cuda_device):
    """40 launches on ONE sync buffer while a second stream keeps part of the chip busy with launches of changing size (so that workgroups of the chain
    arrive late, in changing order and on changing CUs) and the consumer CUs' L1 hold lines of the buffers from the previous launch: every result equals
    the separate launches' bit for bit, and no wait gives up."""
    from sduss_amd import ops
    _host, args = _problem(8, 20, 1024, seed=11)
    want, _st, _f, _q2, _ao2, _ = ops.attn_tail(**args, chained=False)
    sync = None
    side = torch.cuda.Stream()
    noise_a = torch.randn(4096, 4096, device="cuda", dtype=torch.bfloat16)
    stop = torch.cuda.Event()
    for it in range(40):
        with torch.cuda.stream(side):
            for j in range(1 + it % 4):
                n = 512 * (1 + (it + j) % 8)
                (noise_a[:n, :n] @ noise_a[:n, :n]).sum()
        got, _s, _f, _q, _a, sync = ops.attn_tail(**args, chained=True, sync=sync)
        _same(got, want, f"launch {it}")
    stop.record()
    torch.cuda.synchronize()
    assert ops.attn_tail_status(sync) == 0


def test_attn_tail_supported_says_no(cuda_device):
    """shapes the chained launch does not serve are refused by the query (the step plan then issues the four launches)"""
    from sduss_amd import lib, ops
    l = lib.load()
    for b, heads, L, why in ((8, 20, 384, "L % 256 != 0"),
                             (2, 20, 1024, "M = 2048 takes 128-row tiles (one request): the chained launch runs 256 x 160 tiles only")):
        _host, args = _problem(b, heads, L, seed=5)
        with pytest.raises(AssertionError):
            ops.attn_tail(**args, chained=True, finalise=False)
        ops.attn_tail(**args, chained=False, finalise=False)     # ... while the four launches serve it
    assert l.mx_attn_tail_supported(None) == 0
