"""GPU parity AT THE SHAPES bench.py TIMES (BASELINE configs[1] SDXL-base 1024^2 UNet batch 8, configs[2] SD3.5-medium).

Every case runs through the C ABI, asserts through the profiler which kernel family served it (so a tile-chooser change
cannot silently move the case off the kernel the headline rests on), and is compared with a torch fp32 CPU reference of
the same op on the same seeded, bf16-representable inputs.  One full-width 1024^2 sample-forward per model is compared
with the oracle (oracle/sdxl_unet_ref.py, oracle/sd3_mmdit_ref.py; ~1 min of host time each).

Reference call sites: PatchUNet.forward modules/unet.py:205-530, PatchSD3Transformer2DModel.forward
modules/SD3Transformer.py:60-262, PatchSelfAttention / PatchCrossAttention modules/attention.py:59-232.

Tolerance: one bf16-stored op = 2^-7 of the output range (fp32 accumulate, one rounding at 2^-9 relative plus the
range-vs-value slack); attention 2^-6; whole forwards: max error <= 4 % of range and relative L2 <= 2 % (measured values
are printed per case)."""
import ctypes as C
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import sd3_mmdit_ref, sdxl_unet_ref as ref  # noqa: E402  (checker only)

# profiler kinds (csrc/common.h ProfKind)
KIND = {0: "gemm128", 1: "gemm64", 2: "conv128", 3: "conv64", 4: "attn", 5: "norm", 6: "gemm_v2_160", 7: "conv_v2_160",
        8: "gemm_v2_128", 9: "conv_v2_128", 10: "gemm_256x256", 11: "attn_cross", 12: "attn_tail"}
LARGE_GEMM = {"gemm_256x256"}
LARGE_CONV = {"conv_v2_160", "conv_v2_128"}


def _bf(t):
    return t.to(torch.bfloat16)


def _rt(t):
    return t.to(torch.bfloat16).to(torch.float32)


def _close(got, want, rel, what):
    got = got.float().cpu()
    scale = want.abs().max().item() + 1e-6
    err = (got - want).abs().max().item()
    l2 = ((got - want).norm() / (want.norm() + 1e-12)).item()
    print(f"{what}: max err {err:.5f} = {err / scale:.5f} of range, rel L2 {l2:.5f}")
    assert math.isfinite(err), f"{what}: non-finite"
    assert err <= rel * scale, f"{what}: max err {err:.5f} > {rel} * {scale:.4f}"


def _records(fn):
    """runs fn() under mx_profile; returns [(kind name, M, N, K)] of every profiled launch"""
    from sduss_amd import lib
    l = lib.load()
    l.mx_profile_enable(1)
    try:
        fn()
        torch.cuda.synchronize()
        buf = (C.c_double * 64)()
        lib.check(l.mx_profile_collect(buf), "mx_profile_collect")
        rec = (C.c_double * (6 * 64))()
        n = l.mx_profile_records(rec, 64)
    finally:
        l.mx_profile_enable(0)
    return [(KIND.get(int(rec[6 * i]), str(int(rec[6 * i]))), int(rec[6 * i + 1]), int(rec[6 * i + 2]), int(rec[6 * i + 3])) for i in range(n)]


def _gemm_desc(lib, a, w, c, **kw):
    d = lib.GemmDesc()
    d.a, d.w, d.c = a.data_ptr(), w.data_ptr(), c.data_ptr()
    d.M, d.K = a.shape[0] if "M" not in kw else kw.pop("M"), w.shape[1]
    d.N, d.lda, d.ldc = w.shape[0], a.shape[1], c.shape[1]
    for k, v in kw.items():
        setattr(d, k, v.data_ptr() if torch.is_tensor(v) else v)
    return d


def _rand(g, *shape, scale=1.0):
    return _rt(torch.randn(*shape, generator=g) * scale)


# ----------------------------------------------------------------------------------------------------------------------
# the dominant kernel: 256 x 256 GEMM at the K depths of the step (nk = 10 ... 96), persistent and per-sample-vector forms
# ----------------------------------------------------------------------------------------------------------------------
def test_geglu_up_projection_sdxl(cuda_device):
    """60 launches / step: M 8192 (8 x 1024 tokens), N 10240, K 1280; and the 64^2 level: M 32768, N 5120, K 640."""
    from sduss_amd import ops
    from sduss_amd.weights import _geglu_interleave
    for m, dim in ((8192, 1280), (32768, 640)):
        g = torch.Generator().manual_seed(m + dim)
        a = _rand(g, m, dim); w = _rand(g, 8 * dim, dim, scale=dim ** -0.5); b = torch.randn(8 * dim, generator=g)
        hid, gate = (a @ w.t() + b).chunk(2, dim=-1)
        want = hid * F.gelu(gate)
        out = {}
        rec = _records(lambda: out.setdefault("c", ops.gemm(_bf(a).cuda(), _bf(_geglu_interleave(w)).cuda(), _geglu_interleave(b).cuda(), geglu=True)))
        assert [r[0] for r in rec] == ["gemm_256x256"], rec
        _close(out["c"], want, 2.0 ** -7, f"GEGLU M{m} N{8 * dim} K{dim}")


def test_qkv_projection_sdxl(cuda_device):
    """60 launches / step: fused to_q|to_k|to_v, M 8192, N 3840, K 1280, q scaled for mx_attention_prescaled, V transposed."""
    from sduss_amd import ops
    g = torch.Generator().manual_seed(3840)
    nb, rows, dim = 8, 1024, 1280
    a = _rand(g, nb * rows, dim); w = _rand(g, 3 * dim, dim, scale=dim ** -0.5)
    full = (a @ w.t()).reshape(nb * rows, 3, dim)
    out = {}
    rec = _records(lambda: out.setdefault("r", ops.gemm_qkv(_bf(a).cuda(), _bf(w).cuda(), dim, 3, rows, q_scale=ops.ATTN_QSCALE)))
    assert [r[0] for r in rec] == ["gemm_256x256"], rec
    c, vt = out["r"]
    _close(c, torch.cat([full[:, 0] * ops.ATTN_QSCALE, full[:, 1]], dim=1), 2.0 ** -7, "QKV q|k M8192 N3840 K1280")
    _close(ops.unpack_vt(vt, rows), full[:, 2].reshape(nb, rows, dim), 2.0 ** -7, "QKV V^T")


def test_qkv_rmsnorm_sd3(cuda_device):
    """SD3.5 image-stream QKV: M 32768, N 4608, K 1536 with the fused RMSNorm(64) of q / k heads, written into the joint
    [image ; text] sequence (the per-sample-vector-free persistent variant)."""
    from sduss_amd import lib, ops
    l = lib.load()
    g = torch.Generator().manual_seed(4608)
    b, rows, lt, dm, k = 8, 4096, 333, 1536, 1536
    lj = rows + lt
    a = _rand(g, b * rows, k); w = _rand(g, 3 * dm, k, scale=k ** -0.5); bias = torch.randn(3 * dm, generator=g)
    wq = 1 + 0.2 * torch.randn(64, generator=g); wk = 1 + 0.2 * torch.randn(64, generator=g)
    ag, wg, bg, wqg, wkg = _bf(a).cuda(), _bf(w).cuda(), bias.cuda(), wq.cuda(), wk.cuda()
    ldvt = ops.vt_ld(lj)
    qk = torch.zeros(b * lj, 2 * dm, dtype=torch.bfloat16, device="cuda")
    vt = torch.zeros(b, dm, ldvt, dtype=torch.bfloat16, device="cuda")

    def run():
        d = _gemm_desc(lib, ag, wg, qk, bias=bg, vt=vt, flags=lib.EPI_QKV | lib.EPI_RMSNORM, seg=dm, period=3, ldvt=ldvt, rows_per_batch=rows,
                       c_batch_rows=lj, c_row_off=0, rms_wq=wqg, rms_wk=wkg, rms_eps=1e-6, out_scale=ops.ATTN_QSCALE)
        lib.check(l.mx_gemm(lib.current_stream(), C.byref(d)))
    rec = _records(run)
    assert [r[0] for r in rec] == ["gemm_256x256"], rec
    full = (a @ w.t() + bias).reshape(b, rows, 3, dm // 64, 64)

    def rms(x, wgt):
        return x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-6) * wgt
    qkc = qk.float().cpu().reshape(b, lj, 2 * dm)
    _close(qkc[:, :rows, :dm], (rms(full[:, :, 0], wq) * ops.ATTN_QSCALE).reshape(b, rows, dm), 2.0 ** -7, "SD3 QKV rmsnorm q")
    _close(qkc[:, :rows, dm:], rms(full[:, :, 1], wk).reshape(b, rows, dm), 2.0 ** -7, "SD3 QKV rmsnorm k")
    assert qkc[:, rows:].abs().max() == 0, "text rows of the joint buffer were touched"
    _close(ops.unpack_vt(vt.float().cpu(), lj)[:, :rows], full[:, :, 2].reshape(b, rows, dm), 2.0 ** -7, "SD3 QKV V^T")


def test_gated_ff_down_sd3(cuda_device):
    """SD3.5 FF down-projection with the AdaLN-Zero gated residual: M 32768, N 1536, K 6144 (nk = 96, per-sample gate)."""
    from sduss_amd import lib
    l = lib.load()
    g = torch.Generator().manual_seed(6144)
    b, rows, n, k = 8, 4096, 1536, 6144
    a = _rand(g, b * rows, k); w = _rand(g, n, k, scale=k ** -0.5); bias = torch.randn(n, generator=g)
    gate = torch.randn(b, n, generator=g); r = _rand(g, b * rows, n)
    ag, wg, rg, gg, bg = _bf(a).cuda(), _bf(w).cuda(), _bf(r).cuda(), gate.cuda(), bias.cuda()
    out = torch.empty(b * rows, n, dtype=torch.bfloat16, device="cuda")

    def run():
        d = _gemm_desc(lib, ag, wg, out, bias=bg, residual=rg, ldr=n, gate=gg, ldg=n, rows_per_batch=rows)
        lib.check(l.mx_gemm(lib.current_stream(), C.byref(d)))
    rec = _records(run)
    assert [r[0] for r in rec] == ["gemm_256x256"], rec
    _close(out, r + gate.repeat_interleave(rows, dim=0) * (a @ w.t() + bias), 2.0 ** -7, "gated FF2 M32768 N1536 K6144")


def test_out_projection_residual_sdxl(cuda_device):
    """attention out-projection / FF down with the residual add at the step's shapes: M 8192 N 1280 K 1280 and K 5120,
    M 32768 N 640 K 640 and K 2560 (the 256 x 160 / 256 x 128 family, 324 launches / step)."""
    from sduss_amd import ops
    for m, n, k in ((8192, 1280, 1280), (8192, 1280, 5120), (32768, 640, 640), (32768, 640, 2560)):
        g = torch.Generator().manual_seed(m + n + k)
        a = _rand(g, m, k); w = _rand(g, n, k, scale=k ** -0.5); bias = torch.randn(n, generator=g); r = _rand(g, m, n)
        out = {}
        rec = _records(lambda: out.setdefault("c", ops.gemm(_bf(a).cuda(), _bf(w).cuda(), bias.cuda(), residual=_bf(r).cuda())))
        assert len(rec) == 1 and rec[0][0] in {"gemm_v2_160", "gemm_v2_128", "gemm_256x256"}, rec
        _close(out["c"], a @ w.t() + bias + r, 2.0 ** -7, f"out-proj+residual M{m} N{n} K{k} [{rec[0][0]}]")


# ----------------------------------------------------------------------------------------------------------------------
# implicit-GEMM conv, 256-row instantiation, at UNet batch 8 (the bench batch: smaller batches choose 128-row tiles)
# ----------------------------------------------------------------------------------------------------------------------
def _nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def _conv_pack(w):
    return w.permute(0, 2, 3, 1).reshape(w.shape[0], -1).contiguous()


@pytest.mark.parametrize("hw,cin,cout,stride,up", [(128, 320, 320, 1, 0), (64, 1920, 640, 1, 0), (32, 2560, 1280, 1, 0),
                                                   (128, 320, 320, 2, 0), (32, 1280, 1280, 1, 1)])
def test_conv3x3_bench_shapes(cuda_device, hw, cin, cout, stride, up):
    """resnet convs 320->320 @128^2, 1920->640 @64^2, 2560->1280 @32^2 (K = 2880 ... 23040), the stride-2 downsample
    320 @128->64 and the fused nearest-x2 upsample conv 1280 @32->64 (resnet.py:249-378), all at batch 8."""
    from sduss_amd import ops
    g = torch.Generator().manual_seed(hw + cin + cout + stride + up)
    b = 8
    x = _rand(g, b, cin, hw, hw); w = _rand(g, cout, cin, 3, 3, scale=(9 * cin) ** -0.5); bias = torch.randn(cout, generator=g)
    xin = F.interpolate(x, scale_factor=2.0, mode="nearest") if up else x
    with torch.inference_mode():
        want = F.conv2d(xin, w, bias, stride=stride, padding=1)
    out = {}
    xg, wg, bg = _bf(_nhwc(x)).cuda(), _bf(_conv_pack(w)).cuda(), bias.cuda()
    rec = _records(lambda: out.setdefault("c", ops.conv3x3(xg, wg, bg, stride=stride, up=up)))
    assert len(rec) == 1 and rec[0][0] in LARGE_CONV, rec
    _close(out["c"].permute(0, 3, 1, 2), want, 2.0 ** -7, f"conv3x3 {cin}->{cout} @{hw} s{stride} up{up} [{rec[0][0]}]")


# ----------------------------------------------------------------------------------------------------------------------
# attention at the step's shapes: the XCD-aware workgroup remap runs with gridDim.x > 1
# ----------------------------------------------------------------------------------------------------------------------
def _sdpa_ref(q, k, v, heads):
    b, lq, c = q.shape
    sp = lambda t: t.reshape(b, t.shape[1], heads, 64).transpose(1, 2)
    with torch.inference_mode():
        o = F.scaled_dot_product_attention(sp(q), sp(k), sp(v))
    return o.transpose(1, 2).reshape(b, lq, c)


@pytest.mark.parametrize("b,heads,lq,lk", [(2, 4, 1024, 1024), (8, 20, 1024, 1024), (8, 10, 4096, 4096), (8, 10, 4096, 77), (8, 20, 1024, 77),
                                           (2, 24, 4429, 4429)])
def test_attention_bench_shapes(cuda_device, b, heads, lq, lk):
    """self-attention (L 4096 x 10 heads, L 1024 x 20 heads), cross-attention (77 keys) and the SD3 joint sequence (4429), all
    with B*H % 8 == 0 and several query blocks per head, prescaled q as the step plans call it."""
    from sduss_amd import ops
    g = torch.Generator().manual_seed(b * 7 + heads + lq + lk)
    c = heads * 64
    q = _rand(g, b, lq, c); k = _rand(g, b, lk, c); v = _rand(g, b, lk, c)
    qs = _rt(q * ops.ATTN_QSCALE)
    want = _sdpa_ref(qs * (8.0 * math.log(2.0)), k, v, heads)
    out = {}
    qg, kg, vg = _bf(qs.reshape(-1, c)).cuda(), _bf(k.reshape(-1, c)).cuda(), _bf(ops.pack_vt(v, pad=float("nan"))).cuda()
    rec = _records(lambda: out.setdefault("o", ops.attention(qg, kg, vg, heads, lq, lk, prescaled=True)))
    assert len(rec) == 1 and rec[0][0] in {"attn", "attn_cross"}, rec
    _close(out["o"].reshape(b, lq, c), want, 2.0 ** -6, f"attention b{b} h{heads} {lq}x{lk} [{rec[0][0]}]")


# ----------------------------------------------------------------------------------------------------------------------
# one full-width 1024^2 sample-forward per model against the oracle
# ----------------------------------------------------------------------------------------------------------------------
def _check_forward(got, want, what, max_rel=0.04, l2_rel=0.02):
    got = got.float().cpu()
    assert torch.isfinite(got).all(), f"{what}: non-finite output"
    scale = want.abs().max().item()
    err = (got - want).abs().max().item()
    l2 = ((got - want).norm() / want.norm()).item()
    print(f"{what}: max err {err:.4f} ({err / scale:.4f} of range), rel L2 {l2:.4f}")
    assert err <= max_rel * scale and l2 <= l2_rel, f"{what}: max err {err} (range {scale}), rel L2 {l2}"


@pytest.fixture(scope="module")
def sdxl_1024_oracle(full_width_sdxl):
    """The oracle's answer for TWO 1024 px sample-forwards of SDXL-base, shared by the batch-2 and the batch-8 test below (one fp32 CPU
    forward of this size costs ~35 s on the GPU box's host cores): on the ORIGINAL weights for both rows, and on the weights as the device
    holds them (weights.params_as_held: the LayerNorm-folded linears hold bf16(W * gamma)) for row 0."""
    ocfg, P, held, _net = full_width_sdxl
    s, t, e, te, ti = ref.make_inputs(ocfg, 2, 128)
    t = torch.tensor([801.0, 341.0])                       # two different timesteps: the rows must not be interchangeable
    with torch.inference_mode():
        want_orig = ref.unet_forward(P, ocfg, s, t, e, te, ti)
        want_held0 = ref.unet_forward(held, ocfg, s[:1], t[:1], e[:1], te[:1], ti[:1])
    return (s, t, e, te, ti), want_orig, want_held0


def test_sdxl_base_forward_1024(full_width_sdxl, sdxl_1024_oracle):
    """SDXL-base (2.57 B params, random init, bf16-representable) on 128 x 128 latents, UNet batch 2 (one request under
    CFG): the configuration of BASELINE configs[1] at a quarter of the bench batch.  Asserted twice: against the oracle on the weights as the
    device holds them (the bound is then on the kernels' arithmetic alone) and against the oracle on the ORIGINAL weights (real checkpoints have
    LayerNorm gamma != 1, so the fold's extra weight rounding is part of the end-to-end error: 1.9 % -> 2.4 % rel L2)."""
    _ocfg, _P, _held, net = full_width_sdxl
    (s, t, e, te, ti), want_orig, want_held0 = sdxl_1024_oracle
    got = net.forward_one(s.cuda().to(torch.bfloat16), t.cuda(), e.cuda(), te.cuda(), ti.cuda())
    # error budget of tests/test_unet_gpu.py::test_unet_per_stage_error_budget (0.35 % * sqrt(n) + 0.2 % after n bf16-stored stages) at this
    # model's depth -- 17 resnets + 70 transformer layers + 9 convs, n ~ 96 -> 3.6 %; measured 1.9-2.0 % over the round's kernel changes, so the
    # bound is set at 2.5 %, not at the first measurement (a 2.0 % bound flipped on a change of summation order in the GroupNorm statistics)
    _check_forward(got[:1], want_held0, "SDXL-base 1024^2 forward, batch 2, row 0, weights as held", l2_rel=0.025)
    _check_forward(got, want_orig, "SDXL-base 1024^2 forward, batch 2, original weights", max_rel=0.05, l2_rel=0.03)


# ----------------------------------------------------------------------------------------------------------------------
# CALIBRATED tolerance (round 5).  north_star: "within a stated fp16 L-inf tolerance".  The absolute bounds above (2.5-3 % rel L2, 4-5 % of range)
# come from this repo's own error-growth model; what they do not say is how much of that error ANY bf16 / fp16 evaluation of the same graph makes.
# Here the oracle's own graph is evaluated by STOCK torch ops on the GPU box (hipBLASLt / MIOpen / SDPA) in bf16 and in fp16 (the reference's dtype),
# same weights and inputs, and the statement becomes a ratio:
#     err(HIP vs fp32 oracle)  <=  1.5 x err(stock bf16 vs fp32 oracle)          in relative L2 AND in L-inf (fraction of the output range)
# so a kernel that loses 2x the accuracy a stock bf16 pipeline keeps fails whatever the absolute level is.  The fp16 figure is printed next to it.
# The fp32 oracle evaluated on the GPU by the same code (compute_dtype fp32, device cuda) must agree with the CPU oracle to 1e-3: the longer loop
# tests use that form of the oracle (seconds instead of half a minute per forward).
# ----------------------------------------------------------------------------------------------------------------------
CALIBRATION_RATIO = 1.5


def _err2(got, want):
    got = got.float().cpu()
    want = want.float().cpu()
    return ((got - want).norm() / want.norm()).item(), ((got - want).abs().max() / want.abs().max()).item()


def _calibrate(what, got_hip, want, lowp):
    """lowp(dtype, sdpa) -> the oracle graph by stock torch ops at that dtype on the GPU"""
    with torch.inference_mode():
        o32 = lowp(torch.float32, False)
        l2, mx = _err2(o32, want)
        print(f"{what}: fp32 oracle on the GPU (stock ops) vs the CPU oracle: rel L2 {l2:.2e}, max {mx:.2e} of range")
        assert l2 <= 1e-3 and mx <= 2e-3, f"{what}: the oracle evaluated on the GPU in fp32 differs from the CPU oracle (rel L2 {l2})"
        del o32
        res = {}
        for name, dt, sdpa in (("bf16 sdpa", torch.bfloat16, True), ("bf16 math", torch.bfloat16, False), ("fp16 sdpa", torch.float16, True)):
            y = lowp(dt, sdpa)
            res[name] = _err2(y, want) if torch.isfinite(y.float()).all() else (float("inf"), float("inf"))
            del y
            torch.cuda.empty_cache()
    hip = _err2(got_hip, want)
    print(f"{what}: vs the fp32 oracle -- HIP rel L2 {hip[0]:.4f} max {hip[1]:.4f} of range | " +
          " | ".join(f"stock {k} rel L2 {v[0]:.4f} max {v[1]:.4f}" for k, v in res.items()) +
          f" | HIP / stock bf16 sdpa = {hip[0] / res['bf16 sdpa'][0]:.2f} (L2) {hip[1] / res['bf16 sdpa'][1]:.2f} (L-inf)")
    ref_l2, ref_mx = res["bf16 sdpa"]
    assert hip[0] <= CALIBRATION_RATIO * ref_l2, f"{what}: HIP rel L2 {hip[0]:.4f} > {CALIBRATION_RATIO} x stock bf16 {ref_l2:.4f}"
    assert hip[1] <= CALIBRATION_RATIO * ref_mx, f"{what}: HIP L-inf {hip[1]:.4f} of range > {CALIBRATION_RATIO} x stock bf16 {ref_mx:.4f}"


def test_sdxl_base_forward_1024_tolerance_calibrated(full_width_sdxl, sdxl_1024_oracle):
    """SDXL-base 1024^2, UNet batch 2, ORIGINAL weights: the HIP forward may lose at most 1.5 x what a stock bf16 evaluation of the oracle's graph loses."""
    ocfg, P, _held, net = full_width_sdxl
    (s, t, e, te, ti), want_orig, _want_held0 = sdxl_1024_oracle
    got = net.forward_one(s.cuda().to(torch.bfloat16), t.cuda(), e.cuda(), te.cuda(), ti.cuda())
    _calibrate("SDXL-base 1024^2 forward, batch 2", got, want_orig,
               lambda dt, sdpa: ref.unet_forward(P, ocfg, s, t, e, te, ti, compute_dtype=dt, device="cuda", sdpa=sdpa))


def test_sdxl_base_forward_1024_headline_batch8_rows(full_width_sdxl, sdxl_1024_oracle):
    """The HEADLINE batch itself (BASELINE configs[1]: 4 requests under CFG = UNet batch 8 at 128 x 128 latents -- the launch shapes bench.py
    times: 256-row tiles everywhere, the 256 x 256 ping-pong kernel for QKV / GEGLU, the normalisation pass in front of it).  Samples are
    independent, so rows 0 and 7 of the batch-8 forward are compared with the 2-row oracle run; rows 1-6 carry other latents, timesteps and
    embeddings so that a row mix-up cannot pass."""
    ocfg, _P, _held, net = full_width_sdxl
    (s2, t2, e2, te2, ti2), want_orig, want_held0 = sdxl_1024_oracle
    s, t, e, te, ti = ref.make_inputs(ocfg, 8, 128, seed=777)
    t = torch.tensor([0.0, 961.0, 741.0, 521.0, 301.0, 81.0, 21.0, 0.0])
    for src, dst in ((0, 0), (1, 7)):
        s[dst], t[dst], e[dst], te[dst], ti[dst] = s2[src], t2[src], e2[src], te2[src], ti2[src]
    got = net.forward_one(s.cuda().to(torch.bfloat16), t.cuda(), e.cuda(), te.cuda(), ti.cuda())
    assert torch.isfinite(got.float()).all()
    _check_forward(got[0:1], want_held0, "SDXL-base 1024^2 headline batch 8, row 0, weights as held", l2_rel=0.025)
    _check_forward(got[0:1], want_orig[0:1], "SDXL-base 1024^2 headline batch 8, row 0, original weights", max_rel=0.05, l2_rel=0.03)
    _check_forward(got[7:8], want_orig[1:2], "SDXL-base 1024^2 headline batch 8, row 7, original weights", max_rel=0.05, l2_rel=0.03)
    # the other rows did something else (different inputs): a forward that broadcast one row would fail here
    assert (got[3].float() - got[0].float()).abs().max() > 0.1 * got[0].float().abs().max()


def test_sdxl_base_mixed_forward_512_768_1024_one_sequence(full_width_sdxl):
    """The configs[4] launch shapes (bench.py `mixed_stream`): ONE request each at 512 / 768 / 1024 px through mx_unet_forward_mixed at SDXL-base
    width, is_sliced=True / patch 256 as the mixed policies force -- every group against its per-resolution forward (two bf16 evaluations with
    differently grouped fp32 sums) and the 512 px group against the oracle's LITERAL PATCH PIPELINE on the weights as the device holds them (four
    halo'd 256-px patches, patch-averaged GroupNorm statistics, regrouped self-attention)."""
    from oracle import patch_ref
    ocfg, _P, held, net = full_width_sdxl
    spec = [(1, 64), (1, 96), (1, 128)]
    ins = [list(ref.make_inputs(ocfg, b, hw, seed=30 + i)) for i, (b, hw) in enumerate(spec)]
    for i, x in enumerate(ins):
        x[1] = torch.full_like(x[1], 801.0 - 200.0 * i)
    cat = lambda k: torch.cat([x[k] for x in ins]).cuda()
    xs = [x[0].cuda().to(torch.bfloat16) for x in ins]
    got = net.forward_mixed(xs, cat(1), cat(2), cat(3), cat(4), gn_patch=32)
    for i, (b, hw) in enumerate(spec):
        s_, t_, e_, te_, ti_ = ins[i]
        alone = net.forward_one(xs[i], t_.cuda(), e_.cuda(), te_.cuda(), ti_.cuda(), gn_patch=32)
        _check_forward(got[i], alone.float().cpu(), f"SDXL-base mixed 512/768/1024, group {hw * 8} px vs its own sequence", max_rel=0.05, l2_rel=0.03)
    s_, t_, e_, te_, ti_ = ins[0]
    with torch.inference_mode():
        want = patch_ref.unet_forward_sliced(held, ocfg, {"512": s_}, t_, e_, te_, ti_, patch_size=256)["512"]
    _check_forward(got[0], want, "SDXL-base mixed 512/768/1024, 512 px group vs the literal patch pipeline (weights as held)", max_rel=0.05, l2_rel=0.03)


@pytest.fixture(scope="module")
def sd35_1024(cuda_device):
    """SD3.5-medium at full width: parameters, ONE 1024^2 sample with its CPU fp32 oracle answer (~40-75 s of host time) and the packed model"""
    from sduss_amd.config import MMDiTConfig
    from sduss_amd.transformer_sd3 import MxSD3Transformer
    ocfg = sd3_mmdit_ref.MMDiTConfig.sd35_medium()
    P = sd3_mmdit_ref.init_params(ocfg)
    lat, t, e, p = sd3_mmdit_ref.make_inputs(ocfg, 1, 128, ctx_len=333)
    with torch.inference_mode():
        want = sd3_mmdit_ref.mmdit_forward(P, ocfg, lat, t, e, p)
    net = MxSD3Transformer(MMDiTConfig.sd35_medium(), P, device="cuda:0")
    yield ocfg, P, (lat, t, e, p), want, net
    del net
    torch.cuda.empty_cache()


def test_sd35_medium_forward_1024_tolerance_calibrated(sd35_1024):
    """SD3.5-medium 1024^2, one sample: the HIP forward may lose at most 1.5 x what a stock bf16 evaluation of the oracle's graph loses."""
    ocfg, P, (lat, t, e, p), want, net = sd35_1024
    got = net.forward_one(lat.cuda().to(torch.bfloat16), t.cuda(), e.cuda(), p.cuda())
    _calibrate("SD3.5-medium 1024^2 forward, batch 1", got, want,
               lambda dt, sdpa: sd3_mmdit_ref.mmdit_forward(P, ocfg, lat, t, e, p, compute_dtype=dt, device="cuda", sdpa=sdpa))


def test_sd35_medium_forward_1024(sd35_1024):
    """SD3.5-medium (24 joint blocks, 13 dual) on 128 x 128 latents with 333 text tokens: BASELINE configs[2].  The fp32 CPU oracle of this
    forward costs 75 s per sample, so ONE oracle sample serves two comparisons: the batch-1 forward, and -- round 4 -- ROW 7 OF THE HEADLINE
    BATCH OF 8 (4 requests under CFG: the launch shapes bench.py's `sd3` block times, 256 x 256 tiles for every image-stream GEMM and the 64-row
    joint attention), whose other rows carry different latents / timesteps / embeddings so that a row mix-up cannot pass; row 0 of the batch of 8
    is compared with the batch-1 HIP forward of its own inputs (two bf16 evaluations with different tile selections)."""
    ocfg, _P, (lat, t, e, p), want, net = sd35_1024
    got = net.forward_one(lat.cuda().to(torch.bfloat16), t.cuda(), e.cuda(), p.cuda())
    _check_forward(got, want, "SD3.5-medium 1024^2 forward, batch 1")
    lat8, t8, e8, p8 = sd3_mmdit_ref.make_inputs(ocfg, 8, 128, ctx_len=333, seed=4242)
    t8 = torch.tensor([901.0, 41.0, 741.0, 521.0, 301.0, 81.0, 621.0, 0.0])
    lat8[7], t8[7], e8[7], p8[7] = lat[0], t[0], e[0], p[0]
    dev = lambda x: x.cuda().to(torch.bfloat16) if x.dtype == torch.float32 and x.dim() > 1 else x.cuda()
    got8 = net.forward_one(dev(lat8), t8.cuda(), dev(e8), dev(p8))
    assert torch.isfinite(got8.float()).all()
    _check_forward(got8[7:8], want, "SD3.5-medium 1024^2 headline batch 8, row 7")
    got0 = net.forward_one(dev(lat8[:1]), t8[:1].cuda(), dev(e8[:1]), dev(p8[:1]))
    _check_forward(got8[0:1], got0.float().cpu(), "SD3.5-medium 1024^2 headline batch 8, row 0 vs its batch-1 forward", max_rel=0.04, l2_rel=0.02)
    assert (got8[3].float() - got8[7].float()).abs().max() > 0.1 * got8[7].float().abs().max()
