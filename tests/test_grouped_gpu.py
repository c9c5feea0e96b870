"""Grouped launches (include/mxdenoise.h: mx_gemm_seg, mx_attention_grouped, mx_groupnorm_nhwc_grouped): ONE launch over the problems of all
resolutions present in a mixed batch -- the MI355X form of what the reference does by cutting every latent into one patch batch
(modules/unet.py:104-185) and regrouping per latent before attention (attention.py:152-203).  Each test runs the grouped launch and the same
problems as separate launches through the C ABI and checks (a) every problem against an fp32 torch reference of the op and (b) grouped ==
separate bit for bit wherever both ran the same tile shape (output tiles never straddle two problems, so the arithmetic per element is the
same instruction sequence)."""
import ctypes as C
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
    scale = want.abs().max().item() + 1e-6
    err = (got - want).abs().max().item()
    print(f"{what}: max err {err:.5f} = {err / scale:.5f} of range")
    assert math.isfinite(err) and err <= rel * scale, f"{what}: max err {err:.5f} > {rel} * {scale:.4f}"


def _kinds(fn):
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
    return [int(rec[6 * i]) for i in range(n)]


def _desc(lib, **kw):
    d = lib.GemmDesc()
    for k, v in kw.items():
        setattr(d, k, v.data_ptr() if torch.is_tensor(v) else v)
    return d


def _segs(lib, rows):
    arr = (lib.GemmSeg * len(rows))()
    for i, kw in enumerate(rows):
        for k, v in kw.items():
            setattr(arr[i], k, v.data_ptr() if torch.is_tensor(v) else v)
    return arr


@pytest.mark.parametrize("ms,n,k", [((300, 1000, 2304), 640, 640), ((128, 64, 576), 1280, 1280), ((4096, 9216, 16384), 640, 2560), ((2048,), 320, 320)])
def test_grouped_gemm_bias_residual(cuda_device, ms, n, k):
    """plain linear with bias + residual over problems of different row counts (partial last tiles, a problem smaller than a tile)"""
    from sduss_amd import lib
    l = lib.load()
    g = torch.Generator().manual_seed(sum(ms) + n + k)
    w = _rt(torch.randn(n, k, generator=g) * k ** -0.5); bias = torch.randn(n, generator=g)
    wg, bg = _bf(w).cuda(), bias.cuda()
    A = [_rt(torch.randn(m, k, generator=g)) for m in ms]
    R = [_rt(torch.randn(m, n, generator=g)) for m in ms]
    Ag, Rg = [_bf(a).cuda() for a in A], [_bf(r).cuda() for r in R]
    out_g = [torch.empty(m, n, dtype=torch.bfloat16, device="cuda") for m in ms]
    out_s = [torch.empty(m, n, dtype=torch.bfloat16, device="cuda") for m in ms]
    segs = _segs(lib, [dict(a=Ag[i], c=out_g[i], residual=Rg[i], M=ms[i]) for i in range(len(ms))])
    # splitk=1 on both sides: the bit-equality below is a property of the unsplit tilings (a K slicing adds a row's products in another order)
    d = _desc(lib, a=Ag[0], w=wg, c=out_g[0], bias=bg, residual=Rg[0], N=n, K=k, lda=k, ldc=n, ldr=n, n_segs=len(ms), splitk=1)
    d.segs = segs
    kg = _kinds(lambda: lib.check(l.mx_gemm(lib.current_stream(), C.byref(d)), "grouped mx_gemm"))
    assert len(kg) == 1
    ks = []
    for i, m in enumerate(ms):
        di = _desc(lib, a=Ag[i], w=wg, c=out_s[i], bias=bg, residual=Rg[i], M=m, N=n, K=k, lda=k, ldc=n, ldr=n, splitk=1)
        ks += _kinds(lambda: lib.check(l.mx_gemm(lib.current_stream(), C.byref(di)), "mx_gemm"))
    for i, m in enumerate(ms):
        _close(out_g[i], A[i] @ w.t() + bias + R[i], 2.0 ** -7, f"grouped gemm problem {i} (M {m})")
        # the accumulation order along K is the same in every tile shape (K tiles of 64 in order, two k-steps each), and bias / residual are
        # applied per element: grouped == separate bit for bit whatever tile either launch chose
        assert torch.equal(out_g[i], out_s[i]), f"problem {i}: grouped (kind {kg}) != separate (kind {ks[i]})"


def test_grouped_qkv_epilogue(cuda_device):
    """fused q|k|v projection over three resolutions: per-problem tokens per image (256 / 576 / 1024), per-problem V^T block and row length"""
    from sduss_amd import lib, ops
    l = lib.load()
    dim = 640
    spec = [(2, 256), (1, 576), (3, 1024)]                # (images, tokens per image)
    g = torch.Generator().manual_seed(77)
    w = _rt(torch.randn(3 * dim, dim, generator=g) * dim ** -0.5)
    wg = _bf(w).cuda()
    A = [_rt(torch.randn(b * L, dim, generator=g)) for b, L in spec]
    Ag = [_bf(a).cuda() for a in A]
    qk = [torch.zeros(b * L, 2 * dim, dtype=torch.bfloat16, device="cuda") for b, L in spec]
    vt = [torch.zeros(b, dim, (L + 15) // 16 * 16, dtype=torch.bfloat16, device="cuda") for b, L in spec]
    segs = _segs(lib, [dict(a=Ag[i], c=qk[i], vt=vt[i], M=spec[i][0] * spec[i][1], rows_per_batch=spec[i][1], ldvt=vt[i].shape[2]) for i in range(3)])
    d = _desc(lib, a=Ag[0], w=wg, c=qk[0], vt=vt[0], N=3 * dim, K=dim, lda=dim, ldc=2 * dim, flags=lib.EPI_QKV, seg=dim, period=3,
              out_scale=float(ops.ATTN_QSCALE), n_segs=3)
    d.segs = segs
    lib.check(l.mx_gemm(lib.current_stream(), C.byref(d)), "grouped QKV")
    for i, (b, L) in enumerate(spec):
        full = (A[i] @ w.t()).reshape(b * L, 3, dim)
        _close(qk[i], torch.cat([full[:, 0] * ops.ATTN_QSCALE, full[:, 1]], dim=1), 2.0 ** -7, f"grouped QKV q|k problem {i}")
        _close(ops.unpack_vt(vt[i], L), full[:, 2].reshape(b, L, dim), 2.0 ** -7, f"grouped QKV V^T problem {i}")
        c1, v1 = ops.gemm_qkv(Ag[i], wg, dim, 3, L, q_scale=ops.ATTN_QSCALE)
        assert torch.equal(c1, qk[i]) and torch.equal(ops.unpack_vt(v1, L), ops.unpack_vt(vt[i], L)), f"problem {i}: grouped != separate"


@pytest.mark.parametrize("cin,cout,stride,up", [(64, 128, 1, 0), (320, 320, 1, 0), (128, 128, 2, 0), (128, 64, 1, 1)])
def test_grouped_conv3x3(cuda_device, cin, cout, stride, up):
    """implicit-GEMM 3x3 conv over images of three sizes in one launch, with the per-sample row bias (time embedding) and a residual"""
    from sduss_amd import lib
    l = lib.load()
    spec = [(2, 16), (1, 24), (3, 32)]                    # (images, H = W)
    g = torch.Generator().manual_seed(cin + cout + stride + up)
    w = _rt(torch.randn(cout, cin, 3, 3, generator=g) * (9 * cin) ** -0.5); bias = torch.randn(cout, generator=g)
    wg = _bf(w.permute(0, 2, 3, 1).reshape(cout, -1).contiguous()).cuda(); bg = bias.cuda()
    nb = sum(b for b, _ in spec)
    rowbias = torch.randn(nb, cout, generator=g)
    rbg = rowbias.cuda()
    X = [_rt(torch.randn(b, cin, h, h, generator=g)) for b, h in spec]
    Xg = [_bf(x.permute(0, 2, 3, 1).contiguous()).cuda() for x in X]
    ho = [((h << up) + stride - 1) // stride for _b, h in spec]
    R = [_rt(torch.randn(spec[i][0], cout, ho[i], ho[i], generator=g)) for i in range(3)]
    Rg = [_bf(r.permute(0, 2, 3, 1).contiguous()).cuda() for r in R]
    out_g = [torch.empty(spec[i][0], ho[i], ho[i], cout, dtype=torch.bfloat16, device="cuda") for i in range(3)]
    b0 = [0, spec[0][0], spec[0][0] + spec[1][0]]
    rows = []
    for i, (b, h) in enumerate(spec):
        rows.append(dict(a=Xg[i], c=out_g[i], residual=Rg[i], rowbias=rbg[b0[i]:], M=b * ho[i] * ho[i], rows_per_batch=ho[i] * ho[i], B=b, Hin=h, Win=h,
                         Hout=ho[i], Wout=ho[i]))
    segs = _segs(lib, rows)
    d = _desc(lib, a=Xg[0], w=wg, c=out_g[0], bias=bg, residual=Rg[0], rowbias=rbg, N=cout, K=9 * cin, ldc=cout, ldr=cout, ldrb=cout, Cin=cin,
              stride=stride, up=up, n_segs=3, splitk=1)
    d.segs = segs
    lib.check(l.mx_conv3x3(lib.current_stream(), C.byref(d)), "grouped conv3x3")
    for i, (b, h) in enumerate(spec):
        xin = F.interpolate(X[i], scale_factor=2.0, mode="nearest") if up else X[i]
        want = F.conv2d(xin, w, bias, stride=stride, padding=1) + rowbias[b0[i]:b0[i] + b, :, None, None] + R[i]
        _close(out_g[i].permute(0, 3, 1, 2), want, 2.0 ** -7, f"grouped conv {cin}->{cout} s{stride} up{up} problem {i} ({h}x{h})")
        o1 = torch.empty_like(out_g[i])
        d1 = _desc(lib, a=Xg[i], w=wg, c=o1, bias=bg, residual=Rg[i], rowbias=rbg[b0[i]:], M=b * ho[i] * ho[i], N=cout, K=9 * cin, ldc=cout, ldr=cout,
                   ldrb=cout, rows_per_batch=ho[i] * ho[i], B=b, Hin=h, Win=h, Cin=cin, Hout=ho[i], Wout=ho[i], stride=stride, up=up, splitk=1)
        lib.check(l.mx_conv3x3(lib.current_stream(), C.byref(d1)), "conv3x3")
        assert torch.equal(o1, out_g[i]), f"problem {i}: grouped != separate"
        d1.splitk = 0                          # the library's own choice (these small images split over the taps): same values within the rounding
        lib.check(l.mx_conv3x3(lib.current_stream(), C.byref(d1)), "conv3x3")
        _close(o1.permute(0, 3, 1, 2), want, 2.0 ** -7, f"conv {cin}->{cout} problem {i}, automatic split")


def test_grouped_geglu_and_ln_fold_256x256(cuda_device):
    """the persistent 256 x 256 kernel over three problems (GEGLU up-projection: its cursors cross problem boundaries inside the operand
    stream), and the folded LayerNorm: row statistics written by a grouped producer, consumed by a grouped to_q-shaped launch"""
    from sduss_amd import lib, ops
    from sduss_amd.weights import _geglu_interleave
    l = lib.load()
    dim = 640
    ms = (4096, 9216 + 64, 16384)                          # the middle problem ends inside a tile
    g = torch.Generator().manual_seed(5)
    w = _rt(torch.randn(8 * dim, dim, generator=g) * dim ** -0.5); b = torch.randn(8 * dim, generator=g)
    wg, bg = _bf(_geglu_interleave(w)).cuda(), _geglu_interleave(b).cuda()
    A = [_rt(torch.randn(m, dim, generator=g)) for m in ms]
    # (one allocation: the 256 x 256 kernel addresses the problems by 32-bit offsets from the lowest base, and the caching allocator of a
    #  long test session may place separate tensors further apart than that -- the launch then takes a smaller tile, which is not this test)
    flat = _bf(torch.cat(A)).cuda()
    Ag = list(flat.split(list(ms)))
    out = [torch.empty(m, 4 * dim, dtype=torch.bfloat16, device="cuda") for m in ms]
    d = _desc(lib, a=Ag[0], w=wg, c=out[0], bias=bg, N=8 * dim, K=dim, lda=dim, ldc=4 * dim, flags=lib.EPI_GEGLU, n_segs=3)
    d.segs = _segs(lib, [dict(a=Ag[i], c=out[i], M=ms[i]) for i in range(3)])
    kinds = _kinds(lambda: lib.check(l.mx_gemm(lib.current_stream(), C.byref(d)), "grouped GEGLU"))
    assert kinds == [10], kinds                           # the 256 x 256 kernel
    for i, m in enumerate(ms):
        hid, gate = (A[i] @ w.t() + b).chunk(2, dim=-1)
        _close(out[i], hid * F.gelu(gate), 2.0 ** -7, f"grouped GEGLU problem {i}")
        assert torch.equal(out[i], ops.gemm(Ag[i], wg, bg, geglu=True)), f"GEGLU problem {i}: grouped != separate"


def test_grouped_attention(cuda_device):
    """self-attention of three resolutions in one launch (tokens 256 / 576 / 1024 per image -> the LDS-DMA kernel; with 4096 the 64-row
    kernel serves all) and the 77-key cross-attention, against softmax(q k^T / 8) v and against separate launches"""
    from sduss_amd import lib, ops
    l = lib.load()
    heads = 5
    c = heads * 64
    for spec, lk_fixed in (([(2, 256), (1, 576), (3, 1024)], None), ([(1, 1024), (1, 2304), (2, 4096)], None), ([(2, 256), (1, 576), (2, 1024)], 77),
                           ([(1, 1024), (1, 2304), (1, 4096)], 77)):
        g = torch.Generator().manual_seed(sum(b * L for b, L in spec) + (lk_fixed or 0))
        probs = (lib.AttnProblem * len(spec))()
        keep, want, outs = [], [], []
        for i, (b, lq) in enumerate(spec):
            lk = lk_fixed or lq
            q = _rt(torch.randn(b, lq, c, generator=g)); k = _rt(torch.randn(b, lk, c, generator=g)); v = _rt(torch.randn(b, lk, c, generator=g))
            qs = _rt(q * ops.ATTN_QSCALE)
            qh, kh, vh = [t.reshape(b, -1, heads, 64).transpose(1, 2) for t in (qs * (8.0 * math.log(2.0)), k, v)]
            want.append(F.scaled_dot_product_attention(qh, kh, vh).transpose(1, 2).reshape(b, lq, c))
            qg, kg, vg = _bf(qs.reshape(-1, c)).cuda(), _bf(k.reshape(-1, c)).cuda(), _bf(ops.pack_vt(v)).cuda()
            o = torch.empty(b * lq, c, dtype=torch.bfloat16, device="cuda")
            keep.append((qg, kg, vg)); outs.append(o)
            probs[i].q, probs[i].k, probs[i].vt, probs[i].o = qg.data_ptr(), kg.data_ptr(), vg.data_ptr(), o.data_ptr()
            probs[i].vt_batch_stride, probs[i].B, probs[i].Lq, probs[i].Lk, probs[i].ldvt = c * vg.shape[-1], b, lq, lk, vg.shape[-1]
        lib.check(l.mx_attention_prescaled_grouped(lib.current_stream(), probs, len(spec), c, c, c, heads), "grouped attention")
        for i, (b, lq) in enumerate(spec):
            _close(outs[i].reshape(b, lq, c), want[i], 2.0 ** -6, f"grouped attention problem {i} ({b} x {lq} x {lk_fixed or lq})")
            one = (lib.AttnProblem * 1)()
            o1 = torch.empty_like(outs[i])
            for f in ("q", "k", "vt", "vt_batch_stride", "B", "Lq", "Lk", "ldvt"):
                setattr(one[0], f, getattr(probs[i], f))
            one[0].o = o1.data_ptr()
            lib.check(l.mx_attention_prescaled_grouped(lib.current_stream(), one, 1, c, c, c, heads), "attention")
            # a separate launch may select another kernel for this problem (the launch as a whole picks it): same real arithmetic, other
            # association of the online-softmax partial sums
            assert (outs[i].float() - o1.float()).abs().max() <= 2.0 ** -6 * want[i].abs().max(), f"problem {i}: grouped vs separate"


@pytest.mark.parametrize("silu,patch,cat", [(1, 0, False), (0, 8, False), (1, 0, True)])
def test_grouped_groupnorm(cuda_device, silu, patch, cat):
    """GroupNorm(+SiLU) over images of three sizes in one stats / fold / apply launch each; exact and sliced (patch-averaged) statistics; the
    channel concatenation read in place"""
    from oracle import sdxl_unet_ref as ref
    from sduss_amd import lib
    l = lib.load()
    C_, groups, c1 = 320, 32, 192
    spec = [(2, 16), (1, 24), (3, 32)]
    g = torch.Generator().manual_seed(silu + patch)
    gamma = torch.randn(C_, generator=g); beta = torch.randn(C_, generator=g)
    gg, bg = gamma.cuda(), beta.cuda()
    probs = (lib.GnProblem * 3)()
    keep, outs, X = [], [], []
    for i, (b, h) in enumerate(spec):
        x = _rt(torch.randn(b, C_, h, h, generator=g) * 2 + 0.5)
        xg = _bf(x.permute(0, 2, 3, 1).contiguous()).cuda()
        parts = (xg[..., :c1].contiguous(), xg[..., c1:].contiguous()) if cat else (xg, None)
        y = torch.empty(b, h, h, C_, dtype=torch.bfloat16, device="cuda")
        probs[i].x, probs[i].x2, probs[i].y = parts[0].data_ptr(), (parts[1].data_ptr() if cat else None), y.data_ptr()
        probs[i].B, probs[i].H, probs[i].W = b, h, h
        keep.append(parts); outs.append(y); X.append(x)
    ws = torch.empty(l.mx_groupnorm_nhwc_grouped_workspace_bytes(probs, 3, C_), dtype=torch.uint8, device="cuda")
    lib.check(l.mx_groupnorm_nhwc_grouped(lib.current_stream(), probs, 3, c1 if cat else C_, gg.data_ptr(), bg.data_ptr(), C_, groups,
                                          1e-5, silu, patch, ws.data_ptr()), "grouped groupnorm")
    for i, (b, h) in enumerate(spec):
        want = ref.group_norm_patchavg(X[i], groups, gamma, beta, 1e-5, patch) if patch else F.group_norm(X[i], groups, gamma, beta, 1e-5)
        if silu:
            want = F.silu(want)
        _close(outs[i].permute(0, 3, 1, 2), want, 2.0 ** -7, f"grouped groupnorm problem {i} ({h}x{h})")
        y1 = torch.empty_like(outs[i])
        ws1 = torch.empty(l.mx_groupnorm_nhwc_workspace_bytes(b, h, h, C_), dtype=torch.uint8, device="cuda")
        lib.check(l.mx_groupnorm_nhwc_cat(lib.current_stream(), keep[i][0].data_ptr(), c1 if cat else C_, keep[i][1].data_ptr() if cat else None, y1.data_ptr(),
                                          gg.data_ptr(), bg.data_ptr(), b, h, h, C_, groups, 1e-5, silu, patch, ws1.data_ptr()), "groupnorm")
        assert torch.equal(y1, outs[i]), f"problem {i}: grouped != separate"      # the tile geometry of a problem depends on its own size only... see below


# ---------------------------------------------------------------------------------------------------------------------------------------
# the whole step plan on a mixed batch: ONE launch sequence (mx_unet_forward_mixed)
# ---------------------------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def tiny(cuda_device):
    from oracle import sdxl_unet_ref as ref
    from sduss_amd.config import UNetConfig
    from sduss_amd.unet import MxUNet
    ocfg = ref.UNetConfig.tiny()
    P = ref.init_params(ocfg)
    return ocfg, P, MxUNet(UNetConfig.tiny(), P, device="cuda:0")


def _mixed_inputs(ocfg, spec):
    from oracle import sdxl_unet_ref as ref
    ins = [ref.make_inputs(ocfg, b, hw, seed=10 + i) for i, (b, hw) in enumerate(spec)]
    cat = lambda k: torch.cat([x[k] for x in ins])
    for i, x in enumerate(ins):
        x[1].fill_(801.0 - 150.0 * i)                       # another timestep per group
    return ins, cat(1), cat(2), cat(3), cat(4)


def test_mixed_forward_one_group_is_the_ordinary_forward(tiny):
    from oracle import sdxl_unet_ref as ref
    ocfg, P, net = tiny
    s, t, e, te, ti = ref.make_inputs(ocfg, 3, 32)
    x = s.cuda().to(torch.bfloat16)
    a = net.forward_one(x, t.cuda(), e.cuda(), te.cuda(), ti.cuda(), gn_patch=8)
    b = net.forward_mixed([x], t.cuda(), e.cuda(), te.cuda(), ti.cuda(), gn_patch=8)[0]
    assert torch.equal(a, b)


@pytest.mark.parametrize("patch_px", [0, 64])
def test_mixed_forward_three_resolutions(tiny, patch_px):
    """{128, 256, 384} px in one launch sequence (sliced with 64-px patches as the mixed policies run it, and exact): every request within
    the single-forward tolerance of the oracle -- the literal patch pipeline for the sliced form -- and close to the per-resolution sequences
    (not bit-equal: a larger launch selects other tiles, which regroups the fp32 partial sums of the LayerNorm / GroupNorm statistics)."""
    from oracle import patch_ref, sdxl_unet_ref as ref
    ocfg, P, net = tiny
    spec = [(1, 16), (2, 32), (1, 48)]
    ins, t, e, te, ti = _mixed_inputs(ocfg, spec)
    xs = [x[0].cuda().to(torch.bfloat16) for x in ins]
    got = net.forward_mixed(xs, t.cuda(), e.cuda(), te.cuda(), ti.cuda(), gn_patch=patch_px // 8)
    row = 0
    for i, (b, hw) in enumerate(spec):
        s_, t_, e_, te_, ti_ = ins[i]
        if patch_px:
            want = patch_ref.unet_forward_sliced(P, ocfg, {str(hw * 8): s_}, t_, e_, te_, ti_, patch_size=patch_px)[str(hw * 8)]
        else:
            want = ref.unet_forward(P, ocfg, s_, t_, e_, te_, ti_)
        sl = slice(row, row + b)
        alone = net.forward_one(xs[i], t[sl].cuda(), e[sl].cuda(), te[sl].cuda(), ti[sl].cuda(), gn_patch=patch_px // 8)
        g_ = got[i].float().cpu()
        l2 = ((g_ - want).norm() / want.norm()).item()
        l2a = ((g_ - alone.float().cpu()).norm() / want.norm()).item()
        print(f"mixed forward patch {patch_px}: group {i} ({b} x {hw * 8} px): rel L2 to the oracle {l2:.4f}, to the per-resolution sequence {l2a:.4f}")
        assert torch.isfinite(g_).all() and l2 <= 0.02 and (g_ - want).abs().max() <= 0.04 * want.abs().max()
        # two bf16 evaluations of the same network with differently grouped fp32 sums: each is ~1 % from the oracle with independent roundings
        # (measured 1.1 % / 1.3 %), so they are ~1.4x that from each other -- the bound is the oracle bound, not bit equality
        assert l2a <= 0.02
        row += b


def test_mixed_forward_stages_follow_the_per_resolution_stages(tiny):
    """every block output of the mixed sequence (mx_unet_forward_mixed_trace) against the same stage of each group run alone"""
    from oracle import sdxl_unet_ref as ref
    ocfg, P, net = tiny
    spec = [(1, 16), (2, 32), (1, 48)]
    ins, t, e, te, ti = _mixed_inputs(ocfg, spec)
    xs = [x[0].cuda().to(torch.bfloat16) for x in ins]
    tr = {}
    ref.unet_forward(P, ocfg, *ins[0], trace=tr)
    names = [k for k, v in tr.items() if v.ndim == 4]
    worst = 0.0
    for name in names:
        c = tr[name].shape[1]
        scale = tr[name].shape[2] / spec[0][1]                 # this stage's size relative to the latent
        st = net.forward_mixed(xs, t.cuda(), e.cuda(), te.cuda(), ti.cuda(), gn_patch=8, stage=name)[0]
        row, off = 0, 0
        for i, (b, hw) in enumerate(spec):
            h = int(round(hw * scale))
            sl = slice(row, row + b)
            alone = net.forward_one(xs[i], t[sl].cuda(), e[sl].cuda(), te[sl].cuda(), ti[sl].cuda(), gn_patch=8, stage=name, stage_shape=(b, h, h, c))
            mine = st[off:off + b * h * h * c].reshape(b, h, h, c)
            l2 = ((mine.float() - alone.float()).norm() / alone.float().norm()).item()
            worst = max(worst, l2)
            assert l2 <= 0.02, f"stage {name}, group {i}: rel L2 {l2:.4f} between the mixed and the per-resolution sequence"
            off += b * h * h * c
            row += b
    print(f"mixed vs per-resolution stages: worst rel L2 {worst:.5f} over {len(names)} stages")


def test_mixed_denoising_step_equals_per_resolution_steps(tiny):
    """SDXLDenoiser.denoising_step on {128, 256, 384} px requests: the single mixed sequence against the per-resolution sequences, 3 steps"""
    from sduss_amd.config import UNetConfig
    from sduss_amd.pipeline import SDXLDenoiser, synthetic_request
    ocfg, P, net = tiny
    cfg = UNetConfig.tiny()
    outs = []
    for one in (True, False):
        net.mixed_one_sequence = one
        den = SDXLDenoiser(net, guidance_scale=5.0)
        reqs = {"128": [synthetic_request(0, 128, 6, cfg, den, "cuda:0")],
                "256": [synthetic_request(1, 256, 6, cfg, den, "cuda:0"), synthetic_request(2, 256, 8, cfg, den, "cuda:0")],
                "384": [synthetic_request(3, 384, 6, cfg, den, "cuda:0")]}
        for _ in range(3):
            den.denoising_step(reqs, is_sliced=True, patch_size=128)
        torch.cuda.synchronize()
        outs.append({k: torch.cat([r.latents for r in v]).float().cpu() for k, v in reqs.items()})
    net.mixed_one_sequence = True
    for k in outs[0]:
        l2 = ((outs[0][k] - outs[1][k]).norm() / outs[1][k].norm()).item()
        print(f"mixed step {k}: rel L2 to the per-resolution steps {l2:.5f}")
        assert l2 <= 0.03


# ---------------------------------------------------------------------------------------------------------------------------------------
# SD3 / SD3.5: the MMDiT step plan on a mixed batch (mx_mmdit_forward_mixed)
# ---------------------------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def tiny_sd3(cuda_device):
    from oracle import sd3_mmdit_ref as ref
    from sduss_amd.config import MMDiTConfig
    from sduss_amd.transformer_sd3 import MxSD3Transformer
    ocfg = ref.MMDiTConfig.tiny()
    P = ref.init_params(ocfg)
    return ocfg, P, MxSD3Transformer(MMDiTConfig.tiny(), P, device="cuda:0")


def test_mmdit_mixed_forward(tiny_sd3):
    """{128, 256, 384} px in one launch sequence: AdaLN modulation, the q|k|v projections into per-group joint sequences, joint and dual
    attention, the gated projections reading the joint sequence back and the positional tables are all per group.  Every request within the
    single-forward tolerance of the oracle and close to the per-resolution sequence; one group alone is the ordinary forward bit for bit."""
    from oracle import sd3_mmdit_ref as ref
    ocfg, P, net = tiny_sd3
    spec = [(1, 16), (2, 32), (1, 48)]
    ins = [ref.make_inputs(ocfg, b, hw, seed=20 + i, ctx_len=21) for i, (b, hw) in enumerate(spec)]
    for i, x in enumerate(ins):
        x[1].fill_(701.0 - 200.0 * i)
    cat = lambda k: torch.cat([x[k] for x in ins])
    xs = [x[0].cuda().to(torch.bfloat16) for x in ins]
    got = net.forward_mixed(xs, cat(1).cuda(), cat(2).cuda(), cat(3).cuda())
    for i, (b, hw) in enumerate(spec):
        lat, t, e, p = ins[i]
        with torch.inference_mode():
            want = ref.mmdit_forward(P, ocfg, lat, t, e, p)
        alone = net.forward_one(xs[i], t.cuda(), e.cuda(), p.cuda()).float().cpu()
        g_ = got[i].float().cpu()
        l2 = ((g_ - want).norm() / want.norm()).item()
        l2a = ((g_ - alone).norm() / want.norm()).item()
        print(f"mmdit mixed forward: group {i} ({b} x {hw * 8} px): rel L2 to the oracle {l2:.4f}, to the per-resolution sequence {l2a:.4f}")
        assert torch.isfinite(g_).all() and l2 <= 0.02 and (g_ - want).abs().max() <= 0.04 * want.abs().max() and l2a <= 0.02
    one = net.forward_mixed([xs[1]], ins[1][1].cuda(), ins[1][2].cuda(), ins[1][3].cuda())[0]
    assert torch.equal(one, net.forward_one(xs[1], ins[1][1].cuda(), ins[1][2].cuda(), ins[1][3].cuda()))


def test_mmdit_mixed_denoising_step(tiny_sd3):
    from sduss_amd.config import MMDiTConfig
    from sduss_amd.pipeline_sd3 import SD3Denoiser, synthetic_sd3_request
    ocfg, P, net = tiny_sd3
    cfg = MMDiTConfig.tiny()
    outs = []
    for one in (True, False):
        net.mixed_one_sequence = one
        den = SD3Denoiser(net, guidance_scale=7.0)
        reqs = {"128": [synthetic_sd3_request(0, 128, 6, cfg, den, "cuda:0", ctx_len=21)],
                "256": [synthetic_sd3_request(1, 256, 6, cfg, den, "cuda:0", ctx_len=21), synthetic_sd3_request(2, 256, 8, cfg, den, "cuda:0", ctx_len=21)],
                "384": [synthetic_sd3_request(3, 384, 6, cfg, den, "cuda:0", ctx_len=21)]}
        for _ in range(3):
            den.denoising_step(reqs, is_sliced=True, patch_size=128)
        torch.cuda.synchronize()
        outs.append({k: torch.cat([r.latents for r in v]).float().cpu() for k, v in reqs.items()})
    net.mixed_one_sequence = True
    for k in outs[0]:
        l2 = ((outs[0][k] - outs[1][k]).norm() / outs[1][k].norm()).item()
        print(f"sd3 mixed step {k}: rel L2 to the per-resolution steps {l2:.5f}")
        assert l2 <= 0.03
