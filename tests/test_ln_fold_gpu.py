"""LayerNorm folded into its consumer GEMM (mx_gemm_desc.ln_stats / stats_out; include/mxdenoise.h), every kernel family, against
LayerNorm -> linear in fp32 (the reference's BasicTransformerBlock norm1 / norm2 / norm3 -> attn / ff linears, modules/transformer.py:191,
239, 266).  Inputs with a row mean well away from zero (the cancellation the rank-one correction has to survive) and per-row scales."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _bf(t):
    return t.to(torch.bfloat16)


def _rt(t):
    return t.to(torch.bfloat16).to(torch.float32)


def _close(got, want, rel, what):
    got = got.float().cpu()
    assert torch.isfinite(got).all(), f"{what}: non-finite"
    scale = want.abs().max().item()
    err = (got - want).abs().max().item()
    print(f"{what}: max err {err:.5f} of range {scale:.3f}")
    assert err <= rel * scale, f"{what}: max err {err:.5f} > {rel} * {scale:.4f}"


def _hidden(g, m, c):
    """hidden states with row means up to 3 sigma away from zero and row scales over two octaves"""
    x = torch.randn(m, c, generator=g) * (0.5 + 1.5 * torch.rand(m, 1, generator=g)) + 3.0 * torch.randn(m, 1, generator=g)
    return _rt(x)


def _ln_params(g, c):
    return 1.0 + 0.2 * torch.randn(c, generator=g), 0.1 * torch.randn(c, generator=g)


@pytest.mark.parametrize("m,c,n", [(100, 128, 192),        # generic kernel
                                   (600, 640, 640),        # 256 x 160
                                   (256, 1280, 1280),      # 128-row tiles
                                   (4096, 1280, 1280),     # 256 x 160 at a step shape's width
                                   (2048, 640, 1024)])     # 256 x 256 persistent kernel
def test_ln_folded_linear(cuda_device, m, c, n):
    from sduss_amd import ops
    from sduss_amd.weights import fold_layernorm
    g = torch.Generator().manual_seed(m + c + n)
    x = _hidden(g, m, c)
    gamma, beta = _ln_params(g, c)
    w = _rt(torch.randn(n, c, generator=g) * c ** -0.5); bias = 0.1 * torch.randn(n, generator=g)
    want = F.layer_norm(x, (c,), gamma, beta, 1e-5) @ w.t() + bias
    wf, colsum, bf_ = fold_layernorm(w, bias, gamma, beta)
    st = ops.row_stats(_bf(x).cuda())
    got = ops.gemm(_bf(x).cuda(), wf.cuda(), bf_.cuda(), ln_stats=st, ln_colsum=colsum.cuda(), ln_eps=1e-5)
    _close(got, want, 2.0 ** -7, f"ln-folded linear {m}x{c}->{n}")


@pytest.mark.parametrize("m,dim", [(192, 64), (600, 128), (2048, 640)])
def test_ln_folded_geglu(cuda_device, m, dim):
    from sduss_amd import ops
    from sduss_amd.weights import _geglu_interleave, fold_layernorm
    g = torch.Generator().manual_seed(5 + m)
    x = _hidden(g, m, dim)
    gamma, beta = _ln_params(g, dim)
    w = _rt(torch.randn(8 * dim, dim, generator=g) * dim ** -0.5); b = torch.randn(8 * dim, generator=g)
    hid, gate = (F.layer_norm(x, (dim,), gamma, beta, 1e-5) @ w.t() + b).chunk(2, dim=-1)
    want = hid * F.gelu(gate)
    wf, colsum, bf_ = fold_layernorm(_geglu_interleave(w), _geglu_interleave(b), gamma, beta)
    st = ops.row_stats(_bf(x).cuda())
    got = ops.gemm(_bf(x).cuda(), wf.cuda(), bf_.cuda(), geglu=True, ln_stats=st, ln_colsum=colsum.cuda())
    _close(got, want, 2.0 ** -7, f"ln-folded geglu {m}x{dim}")


@pytest.mark.parametrize("rows,dim", [(256, 64), (512, 640), (1024, 1280)])
def test_ln_folded_qkv(cuda_device, rows, dim):
    """norm1 folded into the fused q / k / v projection: q scaled, k plain, V^T transposed -- all from the un-normalised hidden state"""
    from sduss_amd import ops
    from sduss_amd.weights import fold_layernorm
    g = torch.Generator().manual_seed(rows + dim)
    b = 2
    x = _hidden(g, b * rows, dim)
    gamma, beta = _ln_params(g, dim)
    w = _rt(torch.randn(3 * dim, dim, generator=g) * dim ** -0.5)
    y = F.layer_norm(x, (dim,), gamma, beta, 1e-5) @ w.t()
    wf, colsum, bf_ = fold_layernorm(w, None, gamma, beta)
    st = ops.row_stats(_bf(x).cuda())
    c, vt = ops.gemm_qkv(_bf(x).cuda(), wf.cuda(), dim, 3, rows, q_scale=0.25, ln_stats=st, ln_colsum=colsum.cuda(), bias=bf_.cuda())
    _close(c[:, :dim], y[:, :dim] * 0.25, 2.0 ** -7, "ln-folded q")
    _close(c[:, dim:], y[:, dim:2 * dim], 2.0 ** -7, "ln-folded k")
    _close(ops.unpack_vt(vt, rows).reshape(b * rows, dim), y[:, 2 * dim:], 2.0 ** -7, "ln-folded v^T")


@pytest.mark.parametrize("m,k,n", [(600, 640, 640), (256, 1280, 1280), (8192, 1280, 1280), (100, 128, 192), (2048, 5120, 1280), (333, 640, 1280)])
def test_row_stats_from_the_producing_epilogue(cuda_device, m, k, n):
    """stats_out: the (sum, sum of squares) slabs a producing GEMM (to_out / ff.net.2 with the residual) leaves equal those of the rows it
    stored, and feed a folded consumer to the same result as LayerNorm -> linear on the stored rows"""
    from sduss_amd import ops
    from sduss_amd.weights import fold_layernorm
    g = torch.Generator().manual_seed(m + k)
    a = _rt(torch.randn(m, k, generator=g)); w = _rt(torch.randn(n, k, generator=g) * k ** -0.5); bias = torch.randn(n, generator=g)
    res = _hidden(g, m, n)
    y, st = ops.gemm(_bf(a).cuda(), _bf(w).cuda(), bias.cuda(), residual=_bf(res).cuda(), want_stats=True)
    yf = y.float().cpu()
    buf, slabs = st
    s = buf.cpu()[:, :slabs].sum(dim=1)          # entries past the last slab are not initialised (NaN-filled by ops.gemm)
    print(f"stats slabs {slabs} for {m}x{k}->{n}")
    ref1, ref2 = yf.sum(dim=1), (yf * yf).sum(dim=1)
    # the epilogue sums the fp32 values before their rounding to bf16: agreement to the rounding noise of a row
    assert (s[:, 0] - ref1).abs().max().item() <= 2.0 ** -8 * yf.abs().sum(dim=1).max().item()
    assert ((s[:, 1] - ref2).abs() / ref2).max().item() <= 2.0 ** -7
    gamma, beta = _ln_params(g, n)
    w2 = _rt(torch.randn(256, n, generator=g) * n ** -0.5)
    wf, colsum, bf_ = fold_layernorm(w2, None, gamma, beta)
    got = ops.gemm(y, wf.cuda(), bf_.cuda(), ln_stats=st, ln_colsum=colsum.cuda())
    want = F.layer_norm(yf, (n,), gamma, beta, 1e-5) @ w2.t()
    _close(got, want, 2.0 ** -7, f"producer stats -> folded consumer {m}x{k}->{n}")


# ----------------------------------------------------------------------------------------------------------------------
# FINALISED row statistics (round 4, mx_gemm_desc.ln_final): the producer's last workgroup per 256-row panel leaves (mean, rstd) per row and the
# 256 x 256 kernel folds the LayerNorm from those 8 bytes
# ----------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("m,k,n", [(8192, 1280, 1280), (8192, 5120, 1280), (9000, 640, 640), (32768, 2560, 640)])
def test_finalised_row_statistics_of_the_producer(cuda_device, m, k, n):
    """(mean, rstd) per row from the producing launch itself: equal to the moments of the rows it stored (to the bf16 rounding noise of a row),
    bit-identical run to run (the slabs are added in slab order whichever workgroup arrives last), tickets back at zero, slabs still written"""
    from sduss_amd import ops
    g = torch.Generator().manual_seed(m + k + n)
    a = _rt(torch.randn(m, k, generator=g)); w = _rt(torch.randn(n, k, generator=g) * k ** -0.5); bias = torch.randn(n, generator=g)
    res = _hidden(g, m, n)
    ag, wg, bg, rg = _bf(a).cuda(), _bf(w).cuda(), bias.cuda(), _bf(res).cuda()
    y, st, fin = ops.gemm(ag, wg, bg, residual=rg, want_stats=True, want_final=True, ln_eps=1e-5)      # (asserts the tickets are zero again)
    assert fin is not None, "a 256-row tile serves this shape: the launch must be able to finalise"
    yf = y.float().cpu()
    mean, rstd = yf.mean(dim=1), (yf.var(dim=1, unbiased=False) + 1e-5).rsqrt()
    f = fin.cpu()
    assert torch.isfinite(f).all()
    assert (f[:, 0] - mean).abs().max().item() <= 2.0 ** -8 * yf.abs().mean(dim=1).max().item()
    assert ((f[:, 1] - rstd).abs() / rstd).max().item() <= 2.0 ** -8
    buf, slabs = st
    assert torch.isfinite(buf[:, :slabs]).all()
    for _ in range(3):
        y2, _st2, fin2 = ops.gemm(ag, wg, bg, residual=rg, want_stats=True, want_final=True, ln_eps=1e-5)
        assert torch.equal(fin2, fin) and torch.equal(y2, y)


@pytest.mark.parametrize("m,c,kind", [(8192, 1280, "geglu"), (8192, 1280, "qkv"), (16384, 640, "geglu"), (8192, 1280, "plain"), (9000, 640, "plain")])
def test_ln_folded_from_finalised_statistics(cuda_device, m, c, kind):
    """producer (to_out / ff.net.2 shape, with the residual) -> finalised statistics -> consumer on the 256 x 256 kernel, against LayerNorm -> linear in
    fp32 on the rows the producer stored; GEGLU (norm3), the fused q | k | v projection (norm1) and a plain linear"""
    from sduss_amd import lib, ops
    from sduss_amd.weights import _geglu_interleave, fold_layernorm
    g = torch.Generator().manual_seed(m + c + len(kind))
    a = _rt(torch.randn(m, c, generator=g)); w0 = _rt(torch.randn(c, c, generator=g) * c ** -0.5); b0 = torch.randn(c, generator=g)
    res = _hidden(g, m, c)
    y, _st, fin = ops.gemm(_bf(a).cuda(), _bf(w0).cuda(), b0.cuda(), residual=_bf(res).cuda(), want_stats=True, want_final=True, ln_eps=1e-5)
    assert fin is not None
    yf = y.float().cpu()
    gamma, beta = _ln_params(g, c)
    ln = F.layer_norm(yf, (c,), gamma, beta, 1e-5)
    from test_headline_shapes_gpu import _records
    rec, out = [], {}
    def run(fn):
        rec.extend(_records(lambda: out.__setitem__("o", fn())))
        return out["o"]
    if kind == "geglu":
        w = _rt(torch.randn(8 * c, c, generator=g) * c ** -0.5); b = torch.randn(8 * c, generator=g)
        hid, gate = (ln @ w.t() + b).chunk(2, dim=-1)
        want = hid * F.gelu(gate)
        wf, colsum, bf_ = fold_layernorm(_geglu_interleave(w), _geglu_interleave(b), gamma, beta)
        got = run(lambda: ops.gemm(y, wf.cuda(), bf_.cuda(), geglu=True, ln_final=fin, ln_colsum=colsum.cuda()))
        _close(got, want, 2.0 ** -7, f"finalised ln -> geglu {m}x{c}")
    elif kind == "qkv":
        rows = 1024
        w = _rt(torch.randn(3 * c, c, generator=g) * c ** -0.5)
        want = ln @ w.t()
        wf, colsum, bf_ = fold_layernorm(w, None, gamma, beta)
        cq, vt = run(lambda: ops.gemm_qkv(y, wf.cuda(), c, 3, rows, q_scale=0.25, ln_final=fin, ln_colsum=colsum.cuda(), bias=bf_.cuda()))
        _close(cq[:, :c], want[:, :c] * 0.25, 2.0 ** -7, "finalised ln -> q")
        _close(cq[:, c:], want[:, c:2 * c], 2.0 ** -7, "finalised ln -> k")
        _close(ops.unpack_vt(vt, rows).reshape(m, c), want[:, 2 * c:], 2.0 ** -7, "finalised ln -> v^T")
    else:
        n = 1024
        w = _rt(torch.randn(n, c, generator=g) * c ** -0.5); b = 0.1 * torch.randn(n, generator=g)
        want = ln @ w.t() + b
        wf, colsum, bf_ = fold_layernorm(w, b, gamma, beta)
        got = run(lambda: ops.gemm(y, wf.cuda(), bf_.cuda(), ln_final=fin, ln_colsum=colsum.cuda()))
        _close(got, want, 2.0 ** -7, f"finalised ln -> linear {m}x{c}->{n}")
    assert len(rec) == 1 and rec[0][0] == "gemm_256x256", rec       # the fold ran on the 256 x 256 kernel


def test_ln_final_is_refused_where_the_256x256_kernel_does_not_run(cuda_device):
    from sduss_amd import lib, ops
    g = torch.Generator().manual_seed(3)
    # a small launch takes 128-row tiles, which do not finalise: the caller is told (None) and keeps the slabs
    a = _bf(torch.randn(2048, 1280, generator=g)).cuda(); w0 = _bf(torch.randn(1280, 1280, generator=g) * 0.03).cuda()
    _y, (_buf, slabs), fin0 = ops.gemm(a, w0, None, want_stats=True, want_final=True)
    assert fin0 is None and slabs > 0
    y = _bf(torch.randn(512, 640, generator=g)).cuda()
    w = _bf(torch.randn(1920, 640, generator=g)).cuda()            # 1920 % 256 != 0: a 160-wide tile serves it, which reads the slabs
    fin = torch.zeros(512, 2, device="cuda"); cs = torch.zeros(1920, device="cuda")
    with pytest.raises(lib.MxError, match="ln_final"):
        ops.gemm(y, w, None, ln_final=fin, ln_colsum=cs)


def test_inlaunch_handoffs_long_run_under_uneven_load(cuda_device):
    """Advisor (round 4): the finalised row statistics and the split-K combine hand data between workgroups INSIDE a launch (relaxed agent-scope tickets, write-through
    stores, no fence) and rely on every launch leaving its tickets at zero.  A long run under an uneven load on a second stream (launches of changing size, so that
    the last arriver, the CUs and their L1 contents change from launch to launch): 60 finalising launches and 60 split launches, every one bit-identical to the first
    and to the slab / unsplit form, tickets zero at the end."""
    from sduss_amd import ops
    g = torch.Generator().manual_seed(2025)
    m, k, n = 8192, 1280, 1280
    a = _rt(torch.randn(m, k, generator=g)); w = _rt(torch.randn(n, k, generator=g) * k ** -0.5); bias = torch.randn(n, generator=g)
    res = _hidden(g, m, n)
    ag, wg, bg, rg = _bf(a).cuda(), _bf(w).cuda(), bias.cuda(), _bf(res).cuda()
    final = torch.full((m, 2), float("nan"), dtype=torch.float32, device="cuda")
    cnt = torch.zeros((m + 255) // 256, dtype=torch.int32, device="cuda")
    y0, (st0, slabs), f0 = ops.gemm(ag, wg, bg, residual=rg, want_stats=True, want_final=True, ln_eps=1e-5)
    # the finalised pair must be what the slabs add up to (slab order), not only what an earlier launch left
    s = st0[:, :slabs].double().sum(dim=1)
    mean = s[:, 0] / n
    rstd = ((s[:, 1] / n - mean * mean).clamp_min(0) + 1e-5).rsqrt()
    assert (f0[:, 0].double() - mean).abs().max().item() < 1e-5 and ((f0[:, 1].double() - rstd).abs() / rstd).max().item() < 1e-5
    ms, ks = 2048, 5120                                     # a split-K shape (one request's ff.net.2)
    a2 = _bf(_rt(torch.randn(ms, ks, generator=g))).cuda(); w2 = _bf(_rt(torch.randn(n, ks, generator=g) * ks ** -0.5)).cuda(); r2 = _bf(_hidden(g, ms, n)).cuda()
    s0 = ops.gemm(a2, w2, bg, residual=r2, splitk=2)
    u0 = ops.gemm(a2, w2, bg, residual=r2, splitk=1)
    assert ((s0.float() - u0.float()).abs().max() <= 2.0 ** -6 * u0.float().abs().max()).item()
    side = torch.cuda.Stream()
    noise = torch.randn(4096, 4096, device="cuda", dtype=torch.bfloat16)
    for it in range(60):
        with torch.cuda.stream(side):
            for j in range(1 + it % 3):
                q = 512 * (1 + (it + 3 * j) % 8)
                (noise[:q, :q] @ noise[:q, :q]).sum()
        y, _st, f = ops.gemm(ag, wg, bg, residual=rg, want_stats=True, want_final=True, ln_eps=1e-5, final_buffers=(final, cnt))
        assert torch.equal(f.view(torch.int32), f0.view(torch.int32)) and torch.equal(y, y0), f"finalising launch {it} differs"
        sk = ops.gemm(a2, w2, bg, residual=r2, splitk=2)
        assert torch.equal(sk, s0), f"split launch {it} differs"
    torch.cuda.synchronize()
    assert int(cnt.abs().sum()) == 0, "the panel tickets must be zero after every launch"
