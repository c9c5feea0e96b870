"""GPU parity of every HIP kernel, called through the C ABI, against the CPU oracle / a torch fp32 reference of the
same op on the same seeded inputs.  Floating-point path: tolerances are stated per test (bf16 storage = 8 mantissa
bits: one rounding is 2^-9 relative; accumulations are fp32)."""
import ctypes as C
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import patch_ref, scheduler_ref, sdxl_unet_ref as ref  # noqa: E402  (checker only)


def _bf(t):
    return t.to(torch.bfloat16)


def _rt(t):
    """bf16 round trip, fp32 result: what the kernel actually receives."""
    return t.to(torch.bfloat16).to(torch.float32)


def _close(got, want, rel, what):
    got = got.float().cpu()
    scale = want.abs().max().item() + 1e-6
    err = (got - want).abs().max().item()
    assert math.isfinite(err), f"{what}: non-finite"
    assert err <= rel * scale, f"{what}: max err {err:.5f} > {rel} * {scale:.4f}"


def _nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def _conv_pack(w):
    o, i, kh, kw = w.shape
    return w.permute(0, 2, 3, 1).reshape(o, -1).contiguous()


@pytest.mark.parametrize("m,n,k", [(256, 128, 64), (200, 320, 192), (77 * 2, 256, 128), (1024, 640, 640), (130, 4, 576),
                                   (512, 320, 192), (700, 1280, 1280), (300, 2560, 128), (8192, 1280, 640)])
def test_gemm_bias_residual(cuda_device, m, n, k):
    from sduss_amd import ops
    g = torch.Generator().manual_seed(m * 7 + n)
    a = _rt(torch.randn(m, k, generator=g)); w = _rt(torch.randn(n, k, generator=g) * k ** -0.5)
    b = torch.randn(n, generator=g); r = _rt(torch.randn(m, n, generator=g))
    want = a @ w.t() + b + r
    got = ops.gemm(_bf(a).cuda(), _bf(w).cuda(), b.cuda(), residual=_bf(r).cuda())
    _close(got, want, 2.0 ** -7, "gemm+bias+residual")
    got = ops.gemm(_bf(a).cuda(), _bf(w).cuda(), b.cuda(), silu=True, out_f32=True)
    _close(got, F.silu(a @ w.t() + b), 1e-4, "gemm+silu f32")


@pytest.mark.parametrize("m,n,k", [(8, 1280, 2816), (8, 1280, 320), (2, 1280, 1280), (16, 1536, 256), (1, 64, 64), (8, 13760, 1280), (5, 48, 192), (8, 1536, 2048),
                                   (8, 36864, 1536), (3, 22016, 1600)])      # the last two: > 32 M weights, the long-stream form (K % 256 == 0 and a remainder)
def test_gemm_small_m_weight_stream(cuda_device, m, n, k):
    """M <= 16 (gemm_small_m.hip: the time / condition embedding MLPs, the stacked time_emb_proj and AdaLN modulation): every epilogue the plans use on it"""
    from sduss_amd import ops
    g = torch.Generator().manual_seed(m * 131 + n + k)
    a = _rt(torch.randn(m, k, generator=g)); w = _rt(torch.randn(n, k, generator=g) * k ** -0.5)
    b = torch.randn(n, generator=g); r = _rt(torch.randn(m, n, generator=g))
    ad, wd, bd, rd = _bf(a).cuda(), _bf(w).cuda(), b.cuda(), _bf(r).cuda()
    lin = a @ w.t() + b
    _close(ops.gemm(ad, wd, bd), lin, 2.0 ** -7, "small-M bias")
    _close(ops.gemm(ad, wd, None), a @ w.t(), 2.0 ** -7, "small-M no bias")
    _close(ops.gemm(ad, wd, bd, silu=True), F.silu(lin), 2.0 ** -7, "small-M silu")
    _close(ops.gemm(ad, wd, bd, residual=rd, silu=True), F.silu(lin + r), 2.0 ** -7, "small-M residual + silu")
    got = ops.gemm(ad, wd, bd, out_f32=True)
    assert got.dtype == torch.float32
    _close(got, lin, 1e-4, "small-M fp32 out")
    # deterministic: the four K slices of a workgroup are added in wave order
    assert torch.equal(ops.gemm(ad, wd, bd, out_f32=True), got)
    # the row just past the form's reach takes the tile kernel and agrees
    if m == 16:
        a2 = _rt(torch.randn(17, k, generator=g))
        _close(ops.gemm(_bf(a2).cuda(), wd, bd, out_f32=True), a2 @ w.t() + b, 1e-4, "M = 17 (tile kernel)")


def test_gemm_rowbias(cuda_device):
    from sduss_amd import ops
    g = torch.Generator().manual_seed(3)
    bsz, rows, n, k = 3, 64, 192, 128
    a = _rt(torch.randn(bsz * rows, k, generator=g)); w = _rt(torch.randn(n, k, generator=g) * k ** -0.5)
    rb = torch.randn(bsz, n + 64, generator=g)  # wider row stride than N, column offset 0
    want = a @ w.t() + rb[:, :n].repeat_interleave(rows, dim=0)
    got = ops.gemm(_bf(a).cuda(), _bf(w).cuda(), None, rowbias=rb.cuda(), rows_per_batch=rows)
    _close(got, want, 2.0 ** -7, "gemm+rowbias")


@pytest.mark.parametrize("bsz,rows,n,k", [(40, 17, 256, 128), (23, 31, 320, 192), (9, 100, 1280, 256), (64, 16, 640, 128), (50, 15, 256, 128),
                                          (3, 333, 1280, 640), (2, 1025, 1280, 1280)])
def test_gemm_rowbias_residual_row_walk(cuda_device, bsz, rows, n, k):
    """the register epilogue's row bookkeeping (one division per lane, then a 16-token compare-and-wrap walk): batches that are odd, not a
    multiple of 16, shorter than a 128-token wave tile (a wave crosses several samples) or of exactly 16 tokens; 15 tokens go to the generic
    kernel.  Per-sample row bias + residual, M not a multiple of the tile."""
    from sduss_amd import ops
    g = torch.Generator().manual_seed(bsz * 1000 + rows)
    a = _rt(torch.randn(bsz * rows, k, generator=g)); w = _rt(torch.randn(n, k, generator=g) * k ** -0.5)
    b = torch.randn(n, generator=g); r = _rt(torch.randn(bsz * rows, n, generator=g))
    rb = torch.randn(bsz, n, generator=g)
    want = a @ w.t() + b + rb.repeat_interleave(rows, dim=0) + r
    got = ops.gemm(_bf(a).cuda(), _bf(w).cuda(), b.cuda(), residual=_bf(r).cuda(), rowbias=rb.cuda(), rows_per_batch=rows)
    _close(got, want, 2.0 ** -7, f"gemm+rowbias+residual, {bsz} samples of {rows} rows")


def test_gemm_geglu(cuda_device):
    from sduss_amd import ops
    from sduss_amd.weights import _geglu_interleave
    _geglu_case(192, 64)
    _geglu_case(600, 128)      # 256-row pipelined kernel


def _geglu_case(m, dim):
    from sduss_amd import ops
    from sduss_amd.weights import _geglu_interleave
    g = torch.Generator().manual_seed(5)
    a = _rt(torch.randn(m, dim, generator=g)); w = _rt(torch.randn(8 * dim, dim, generator=g) * dim ** -0.5)
    b = torch.randn(8 * dim, generator=g)
    hid, gate = (a @ w.t() + b).chunk(2, dim=-1)
    want = hid * F.gelu(gate)
    got = ops.gemm(_bf(a).cuda(), _bf(_geglu_interleave(w)).cuda(), _geglu_interleave(b).cuda(), geglu=True)
    assert got.shape == (m, 4 * dim)
    _close(got, want, 2.0 ** -7, "gemm+geglu")


@pytest.mark.parametrize("period,rows,dim", [(3, 64, 128), (2, 77, 64), (3, 256, 64), (3, 256, 640), (2, 77, 320), (3, 512, 128), (3, 333, 128), (3, 17, 128)])
def test_gemm_qkv_split(cuda_device, period, rows, dim):
    from sduss_amd import ops
    g = torch.Generator().manual_seed(period * 11 + rows)
    nb, k, groups = (2 if rows > 77 else 5 if rows > 17 else 24), 128, 2 if period == 2 else 1
    n = groups * period * dim
    a = _rt(torch.randn(nb * rows, k, generator=g)); w = _rt(torch.randn(n, k, generator=g) * k ** -0.5)
    full = a @ w.t()
    c, vt = ops.gemm_qkv(_bf(a).cuda(), _bf(w).cuda(), dim, period, rows)
    segs = full.reshape(nb * rows, groups, period, dim)
    want_c = segs[:, :, :period - 1].reshape(nb * rows, -1)
    want_v = segs[:, :, period - 1].reshape(nb, rows, groups * dim)                    # [nb, key, V cols]
    _close(c, want_c, 2.0 ** -7, "qkv row-major part")
    assert vt.shape[2] == ops.vt_ld(rows)
    _close(ops.unpack_vt(vt, rows), want_v, 2.0 ** -7, "qkv transposed V (MX_VT_POS key order)")


@pytest.mark.parametrize("stride,up,corner,hw,cin,cout", [(1, 0, 0, 16, 64, 128), (2, 0, 0, 16, 128, 64), (1, 1, 0, 8, 64, 64),
                                                          (1, 0, 4, 16, 64, 64), (2, 0, 8, 16, 64, 64), (1, 1, 8, 8, 64, 64),
                                                          (1, 0, 0, 12, 192, 320), (1, 0, 0, 16, 64, 640), (2, 0, 8, 32, 64, 320),
                                                          (1, 1, 8, 16, 128, 160), (1, 0, 4, 16, 64, 256)])
def test_conv3x3(cuda_device, stride, up, corner, hw, cin, cout):
    from sduss_amd import ops
    g = torch.Generator().manual_seed(stride * 100 + up * 10 + corner + cin)
    b = 2
    x = _rt(torch.randn(b, cin, hw, hw, generator=g)); w = _rt(torch.randn(cout, cin, 3, 3, generator=g) * (9 * cin) ** -0.5)
    bias = torch.randn(cout, generator=g)
    xin = F.interpolate(x, scale_factor=2.0, mode="nearest") if up else x
    want = ref.conv3x3(xin, w, bias, stride, corner if corner else None)
    got = ops.conv3x3(_bf(_nhwc(x)).cuda(), _bf(_conv_pack(w)).cuda(), bias.cuda(), stride=stride, up=up, corner_patch=corner)
    _close(got.permute(0, 3, 1, 2), want, 2.0 ** -7, f"conv3x3 s{stride} up{up} corner{corner}")


@pytest.mark.parametrize("corner,h,w,cin,cout,b", [(0, 16, 16, 64, 4, 2), (0, 12, 20, 320, 4, 3), (4, 16, 16, 64, 4, 2), (8, 32, 32, 128, 8, 1), (0, 7, 9, 64, 16, 2), (0, 128, 128, 320, 4, 2)])
def test_conv3x3_small_n(cuda_device, corner, h, w, cin, cout, b):
    """a handful of output channels (conv_small_n.hip: the UNet's conv_out): ragged image edges, the sliced path's corner rule, several channel counts"""
    from sduss_amd import ops
    g = torch.Generator().manual_seed(corner * 7 + h + cin + cout)
    x = _rt(torch.randn(b, cin, h, w, generator=g)); wt = _rt(torch.randn(cout, cin, 3, 3, generator=g) * (9 * cin) ** -0.5)
    bias = torch.randn(cout, generator=g)
    want = ref.conv3x3(x, wt, bias, 1, corner if corner else None)
    got = ops.conv3x3(_bf(_nhwc(x)).cuda(), _bf(_conv_pack(wt)).cuda(), bias.cuda(), corner_patch=corner)
    _close(got.permute(0, 3, 1, 2), want, 2.0 ** -7, f"conv3x3 small N corner{corner}")
    if corner == 0:
        got2 = ops.conv3x3(_bf(_nhwc(x)).cuda(), _bf(_conv_pack(wt)).cuda(), None)
        _close(got2.permute(0, 3, 1, 2), F.conv2d(x, wt, None, padding=1), 2.0 ** -7, "conv3x3 small N, no bias")


@pytest.mark.parametrize("corner,h,w,cout,b,cv", [(0, 16, 16, 80, 2, 4), (0, 12, 20, 320, 3, 4), (4, 16, 16, 160, 2, 8), (0, 7, 9, 80, 2, 3), (0, 128, 128, 320, 2, 4)])
def test_conv3x3_small_cin(cuda_device, corner, h, w, cout, b, cv):
    """a handful of non-zero input channels inside rows padded to 64 (conv_small_n.hip, conv3x3_small_cin_kernel: the UNet's conv_in): same result as the tile
    kernel over the padded input, which the oracle checks"""
    from sduss_amd import lib, ops
    g = torch.Generator().manual_seed(corner * 7 + h + cout + cv)
    x = torch.zeros(b, 64, h, w)
    x[:, :cv] = _rt(torch.randn(b, cv, h, w, generator=g))
    wt = _rt(torch.randn(cout, 64, 3, 3, generator=g) * (9 * cv) ** -0.5)     # (weights of the padded channels: anything -- they meet zeros)
    bias = torch.randn(cout, generator=g)
    want = ref.conv3x3(x, wt, bias, 1, corner if corner else None)
    xd, wd = _bf(_nhwc(x)).cuda(), _bf(_conv_pack(wt)).cuda()
    got = ops.conv3x3(xd, wd, bias.cuda(), corner_patch=corner, cin_valid=8)
    _close(got.permute(0, 3, 1, 2), want, 2.0 ** -7, f"conv3x3 small Cin corner{corner}")
    plain = ops.conv3x3(xd, wd, bias.cuda(), corner_patch=corner)                  # the tile kernel over all 64 channels
    _close(got.permute(0, 3, 1, 2), plain.float().cpu().permute(0, 3, 1, 2), 2.0 ** -7, "small Cin vs the tile kernel")


def test_conv3x3_rowbias_residual(cuda_device):
    from sduss_amd import ops
    g = torch.Generator().manual_seed(9)
    b, c, hw = 2, 64, 8
    x = _rt(torch.randn(b, c, hw, hw, generator=g)); w = _rt(torch.randn(c, c, 3, 3, generator=g) * (9 * c) ** -0.5)
    bias = torch.randn(c, generator=g); temb = torch.randn(b, c, generator=g); r = _rt(torch.randn(b, c, hw, hw, generator=g))
    want = F.conv2d(x, w, bias, padding=1) + temb[:, :, None, None] + r
    got = ops.conv3x3(_bf(_nhwc(x)).cuda(), _bf(_conv_pack(w)).cuda(), bias.cuda(), rowbias=temb.cuda(),
                      residual=_bf(_nhwc(r)).cuda().reshape(-1, c))
    _close(got.permute(0, 3, 1, 2), want, 2.0 ** -7, "conv3x3+temb+residual")


@pytest.mark.parametrize("lq,lk,heads", [(256, 256, 2), (1024, 1024, 1), (64, 77, 4), (200, 77, 1), (4096, 4096, 1)])
def test_attention(cuda_device, lq, lk, heads):
    from sduss_amd import ops
    g = torch.Generator().manual_seed(lq + lk + heads)
    b, c = 2, heads * 64
    q = _rt(torch.randn(b, lq, c, generator=g)); k = _rt(torch.randn(b, lk, c, generator=g)); v = _rt(torch.randn(b, lk, c, generator=g))
    want = ref.attention(q, k, v, heads)
    vt = ops.pack_vt(v, pad=float("nan"))             # the pad must never leak
    got = ops.attention(_bf(q.reshape(-1, c)).cuda(), _bf(k.reshape(-1, c)).cuda(), _bf(vt).cuda(), heads, lq, lk)
    _close(got.reshape(b, lq, c), want, 2.0 ** -6, f"attention {lq}x{lk}")


def test_attention_spiked_max(cuda_device):
    """forces the running-max rescale branch at a late KV tile (cdna guide rule 26)."""
    from sduss_amd import ops
    g = torch.Generator().manual_seed(1)
    b, l, c = 1, 512, 64
    q = _rt(torch.randn(b, l, c, generator=g)); k = _rt(torch.randn(b, l, c, generator=g)); v = _rt(torch.randn(b, l, c, generator=g))
    k[0, 300] = q[0, 5] * 4.0   # one key aligned with one query: the max jumps in tile 4
    k[0, 450] = q[0, 77] * 6.0
    want = ref.attention(q, k, v, 1)
    got = ops.attention(_bf(q.reshape(-1, c)).cuda(), _bf(k.reshape(-1, c)).cuda(), _bf(ops.pack_vt(v)).cuda(), 1, l, l)
    _close(got.reshape(b, l, c), want, 2.0 ** -6, "attention spiked")


@pytest.mark.parametrize("c", [64, 640, 1280, 256])
def test_layernorm(cuda_device, c):
    from sduss_amd import ops
    g = torch.Generator().manual_seed(c)
    x = _rt(torch.randn(300, c, generator=g) * 2 + 0.5); ga = torch.randn(c, generator=g); be = torch.randn(c, generator=g)
    want = F.layer_norm(x, (c,), ga, be, 1e-5)
    got = ops.layernorm(_bf(x).cuda(), ga.cuda(), be.cuda(), 1e-5)
    _close(got, want, 2.0 ** -7, "layernorm")


@pytest.mark.parametrize("c,hw,patch,silu", [(64, 16, 0, True), (320, 32, 0, True), (192, 16, 4, False), (640, 16, 8, True),
                                            (2560, 8, 0, True), (128, 32, 16, True)])
def test_groupnorm_nhwc(cuda_device, c, hw, patch, silu):
    from sduss_amd import ops
    g = torch.Generator().manual_seed(c + hw + patch)
    x = _rt(torch.randn(2, c, hw, hw, generator=g) * 1.5 + 0.7); ga = torch.randn(c, generator=g); be = torch.randn(c, generator=g)
    want = ref._gn(x, 32, ga, be, 1e-5, patch if patch else None)
    if silu:
        want = F.silu(want)
    got = ops.groupnorm_nhwc(_bf(_nhwc(x)).cuda(), ga.cuda(), be.cuda(), 32, 1e-5, silu, patch)
    _close(got.permute(0, 3, 1, 2), want, 2.0 ** -7, f"groupnorm nhwc C{c} p{patch}")


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.float16, 2.0 ** -9), (torch.bfloat16, 2.0 ** -7)])
@pytest.mark.parametrize("padding", [True, False])
def test_esymred_groupnorm(cuda_device, dtype, tol, padding):
    """The inner boundary: same call as groupnorm.py:50,58 on a split_sample() patch batch of two resolutions."""
    from sduss_amd import esymred_mp
    g = torch.Generator().manual_seed(21)
    c, cpg = 64, 2
    samples = {"256": torch.randn(1, c, 32, 32, generator=g), "384": torch.randn(2, c, 48, 48, generator=g)}
    pidx, lat_off, _res_off, patches, pmap = patch_ref.split_sample(samples, 128)
    x = patches[:, :, 1:-1, 1:-1].contiguous().to(dtype).to(torch.float32)   # interior: what the op receives mid-network
    ga = torch.randn(c, generator=g).to(dtype).float(); be = torch.randn(c, generator=g).to(dtype).float()
    want = patch_ref.groupnorm(x, ga, be, cpg, 1e-5, padding, lat_off, pmap, pidx)
    n, _, h, w = x.shape
    got = esymred_mp.groupnorm(x.to(dtype).cuda(), ga.to(dtype).cuda(), be.to(dtype).cuda(), n, c, h, w, cpg, 1e-5, padding,
                               torch.tensor(lat_off, dtype=torch.int32).cuda(), pmap.cuda(), pidx.cuda())
    assert got.shape == want.shape and got.dtype == dtype
    _close(got, want, tol, f"esymred_mp.groupnorm {dtype} pad={padding}")


def _asymmetric_table(n):
    """an adjacency table split_sample would never produce: one-directional links, a link between two latents, a self link --
    every halo side still has at most one writer (with two the reference's scatter races)"""
    pidx = torch.full((n, 4), -1, dtype=torch.int32)
    pidx[0, 3] = 1          # 0 writes into 1's left halo, 1 does NOT name 0 as its left neighbour
    pidx[1, 2] = 4          # 1 -> 4 downwards, across a latent boundary
    pidx[4, 0] = 2          # 4 -> 2 upwards (not the inverse of the link above)
    pidx[3, 1] = 3          # a patch that is its own left neighbour
    pidx[5, 3] = 0
    pidx[2, 0] = 5
    return pidx.reshape(-1)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
def test_esymred_halo_on_an_asymmetric_table_equals_the_senders_scatter(cuda_device, dtype):
    """The library gathers halos by the receiver while the reference scatters them from the sender (norm_silu_concat.cu:164-241).  The two agree
    for any table once the receiver looks its writer up in the INVERSE table and applies the SENDER's statistics (csrc/gn_halo_nchw.hip); here on a
    table with one-directional, cross-latent and self links, against the oracle's literal sender-driven scatter: halo-only bit for bit, GroupNorm +
    halo within the dtype's tolerance (the statistics differ between the two latents, so a receiver-statistics shortcut would fail)."""
    from sduss_amd import esymred_mp
    g = torch.Generator().manual_seed(23)
    n, c, h, w, cpg = 6, 16, 8, 8, 4
    x = (torch.randn(n, c, h, w, generator=g) * torch.tensor([1., 1., 1., 5., 5., 5.]).view(n, 1, 1, 1) + torch.arange(n).view(n, 1, 1, 1).float()).to(dtype)
    pidx = _asymmetric_table(n)
    got = esymred_mp.mock_groupnorm(x.cuda(), n, c, h, w, 1, pidx.cuda()).cpu()
    want = patch_ref.mock_groupnorm(x, pidx)
    assert torch.equal(got.view(torch.uint8), want.view(torch.uint8))
    lat_off, pmap = [0, 3, 6], torch.tensor([1, 1, 1, 2, 2, 2], dtype=torch.int32)
    ga = torch.randn(c, generator=g).to(dtype).float(); be = torch.randn(c, generator=g).to(dtype).float()
    want = patch_ref.groupnorm(x.float(), ga, be, cpg, 1e-5, True, lat_off, pmap, pidx)
    got = esymred_mp.groupnorm(x.cuda(), ga.to(dtype).cuda(), be.to(dtype).cuda(), n, c, h, w, cpg, 1e-5, True,
                               torch.tensor(lat_off, dtype=torch.int32).cuda(), pmap.cuda(), pidx.cuda())
    tol = {torch.float32: 1e-5, torch.float16: 2.0 ** -9, torch.bfloat16: 2.0 ** -7}[dtype]
    _close(got, want, tol, f"asymmetric table {dtype}")


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
def test_esymred_mock_groupnorm_bit_exact(cuda_device, dtype):
    """Halo exchange is pure data movement: bit-exact."""
    from sduss_amd import esymred_mp
    g = torch.Generator().manual_seed(22)
    c = 32
    samples = {"512": torch.randn(2, c, 64, 64, generator=g)}
    pidx, _lo, _ro, patches, _pm = patch_ref.split_sample(samples, 128)
    x = patches[:, :, 1:-1, 1:-1].contiguous().to(dtype)
    want = patch_ref.mock_groupnorm(x, pidx)
    n, _, h, w = x.shape
    got = esymred_mp.mock_groupnorm(x.cuda(), n, c, h, w, 1, pidx.cuda()).cpu()
    assert torch.equal(got.view(torch.uint8), want.view(torch.uint8))


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
def test_scheduler_steps(cuda_device, dtype):
    from sduss_amd import ops
    g = torch.Generator().manual_seed(31)
    _ts, sigmas, _ = scheduler_ref.sdxl_euler_tables(50)
    n = 3
    lat = (torch.randn(n, 4, 16, 16, generator=g) * 10).to(dtype)
    sig = sigmas[[0, 7, 30]]; sig_next = sigmas[[1, 8, 31]]
    want = scheduler_ref.scale_model_input(torch.cat([lat, lat]), torch.cat([sig, sig]))
    got = ops.euler_scale_input(lat.cuda(), sig, 2 * n).cpu()
    assert torch.equal(got.view(torch.uint8), want.view(torch.uint8)), "scale_model_input must be bit-exact"
    noise = torch.randn(2 * n, 4, 16, 16, generator=g).to(dtype)
    want = scheduler_ref.euler_step(scheduler_ref.cfg_combine(noise, 5.0), lat, sig, sig_next)
    got = ops.cfg_euler_step_(noise.cuda(), lat.cuda().clone(), sig, sig_next, 5.0).cpu()
    # same IEEE op sequence as torch (no FMA contraction in the kernel): bit-exact
    assert torch.equal(got.view(torch.uint8), want.view(torch.uint8)), "CFG combine + Euler step must be bit-exact"


# ----------------------------------------------------------------------------------------------------------------------
# 256 x 256 tile kernel (gemm_bf16_v3.hip): shapes for which the tile chooser picks it, checked through the profiler
# ----------------------------------------------------------------------------------------------------------------------
class _expect_v3:
    """asserts that every GEMM launched inside the block ran the 256x256 kernel (profiler kind 10)"""

    def __enter__(self):
        from sduss_amd import lib
        self.l = lib.load()
        self.l.mx_profile_enable(1)
        return self

    def __exit__(self, *exc):
        from sduss_amd import lib
        torch.cuda.synchronize()
        buf = (C.c_double * 64)()
        lib.check(self.l.mx_profile_collect(buf), "mx_profile_collect")
        self.l.mx_profile_enable(0)
        if exc[0] is None:
            launches = {k: int(buf[4 * k]) for k in range(12) if buf[4 * k] > 0}
            assert set(launches) == {10}, f"expected only the 256x256 GEMM kernel, profiler saw kinds {launches}"
        return False


@pytest.mark.parametrize("m,n,k", [(4096, 4096, 256), (3900, 4096, 192), (8192, 2048, 128),
                                   (8000, 8192, 128), (4900, 4096, 192)])   # the last two: more tiles than CUs -> persistent stream
def test_gemm_tile256_bias_residual(cuda_device, m, n, k):
    from sduss_amd import ops
    g = torch.Generator().manual_seed(m + n + k)
    a = _rt(torch.randn(m, k, generator=g)); w = _rt(torch.randn(n, k, generator=g) * k ** -0.5)
    bias = torch.randn(n, generator=g); r = _rt(torch.randn(m, n, generator=g))
    want = a @ w.t() + bias + r
    with _expect_v3():
        got = ops.gemm(_bf(a).cuda(), _bf(w).cuda(), bias.cuda(), residual=_bf(r).cuda())
    _close(got, want, 2.0 ** -7, f"gemm tile256 {m}x{n}x{k}")


def test_gemm_tile256_geglu(cuda_device):
    from sduss_amd import ops
    from sduss_amd.weights import _geglu_interleave
    g = torch.Generator().manual_seed(77)
    m, dim = 4096, 512
    a = _rt(torch.randn(m, dim, generator=g)); w = _rt(torch.randn(8 * dim, dim, generator=g) * dim ** -0.5)
    b = torch.randn(8 * dim, generator=g)
    hid, gate = (a @ w.t() + b).chunk(2, dim=-1)
    want = hid * F.gelu(gate)
    with _expect_v3():
        got = ops.gemm(_bf(a).cuda(), _bf(_geglu_interleave(w)).cuda(), _geglu_interleave(b).cuda(), geglu=True)
    _close(got, want, 2.0 ** -7, "gemm tile256 geglu")


def test_gemm_tile256_qkv(cuda_device):
    from sduss_amd import ops
    g = torch.Generator().manual_seed(78)
    nb, rows, dim, k = 4, 1024, 1024, 128
    a = _rt(torch.randn(nb * rows, k, generator=g)); w = _rt(torch.randn(3 * dim, k, generator=g) * k ** -0.5)
    full = (a @ w.t()).reshape(nb * rows, 3, dim)
    with _expect_v3():
        c, vt = ops.gemm_qkv(_bf(a).cuda(), _bf(w).cuda(), dim, 3, rows)
    _close(c, full[:, :2].reshape(nb * rows, -1), 2.0 ** -7, "tile256 qkv row-major part")
    _close(ops.unpack_vt(vt, rows), full[:, 2].reshape(nb, rows, dim), 2.0 ** -7, "tile256 qkv V^T")


# ----------------------------------------------------------------------------------------------------------------------
# mx_attention_prescaled: q carries scale*log2(e); the softmax reference is subtracted inside the matrix core
# ----------------------------------------------------------------------------------------------------------------------
def _prescaled_case(q, k, v, heads, what, tol=2.0 ** -6):
    """q, k, v fp32 [b, l, c].  The kernel sees qs = bf16(q * QSCALE); the reference is computed from that same qs
    (softmax_j exp2(qs . k_j) == attention with q' = qs * 8 ln2 at scale 1/8)."""
    from sduss_amd import ops
    b, lq, c = q.shape
    lk = k.shape[1]
    qs = _rt(q * ops.ATTN_QSCALE)
    want = ref.attention(qs * (8.0 * math.log(2.0)), k, v, heads)
    got = ops.attention(_bf(qs.reshape(-1, c)).cuda(), _bf(k.reshape(-1, c)).cuda(), _bf(ops.pack_vt(v, pad=float("nan"))).cuda(),
                        heads, lq, lk, prescaled=True)
    _close(got.reshape(b, lq, c), want, tol, what)


@pytest.mark.parametrize("lq,lk,heads", [(256, 256, 2), (1024, 1024, 1), (64, 77, 4), (200, 77, 1), (4096, 4096, 1), (333, 4429, 1), (600, 320, 2), (777, 192, 8), (2125, 1101, 1), (2048, 192, 2),
                                           (2300, 141, 1)])
def test_attention_prescaled(cuda_device, lq, lk, heads):
    g = torch.Generator().manual_seed(lq * 3 + lk + heads)
    b, c = 2, heads * 64
    q = _rt(torch.randn(b, lq, c, generator=g)); k = _rt(torch.randn(b, lk, c, generator=g)); v = _rt(torch.randn(b, lk, c, generator=g))
    _prescaled_case(q, k, v, heads, f"attention prescaled {lq}x{lk}")


@pytest.mark.parametrize("l", [512, 2112])     # 32-row-per-wave kernels / the 64-row-per-wave kernel (Lq >= 2048)
@pytest.mark.parametrize("case", ["late_spikes", "first_tile_max", "very_negative_start", "huge_logits"])
def test_attention_prescaled_reference_moves(cuda_device, case, l):
    """the lazy softmax reference: raised at late tiles, never needed again, started far below zero, and logits whose
    exp2 would overflow fp32 without a reference."""
    g = torch.Generator().manual_seed(17)
    b, c = 1, 64
    q = _rt(torch.randn(b, l, c, generator=g)); k = _rt(torch.randn(b, l, c, generator=g)); v = _rt(torch.randn(b, l, c, generator=g))
    if case == "late_spikes":
        k[0, 300] = q[0, 5] * 4.0; k[0, 450] = q[0, 77] * 6.0; k[0, 451] = q[0, 77] * 9.0
    elif case == "first_tile_max":
        k[0, 3] = q[0, 9] * 8.0
    elif case == "very_negative_start":
        k[0, :64] = -q[0, 11] * 6.0            # row 11 sees strongly negative logits in the whole first tile
    elif case == "huge_logits":
        q = q * 6.0; k = k * 6.0               # |logit| * log2(e) well beyond 128
    _prescaled_case(_rt(q), _rt(k), v, 1, f"attention prescaled {case}")


def test_gemm_out_scale(cuda_device):
    from sduss_amd import ops
    g = torch.Generator().manual_seed(91)
    for m, n, k in ((300, 192, 128), (4096, 4096, 128)):        # generic kernel / 256x256 kernel
        a = _rt(torch.randn(m, k, generator=g)); w = _rt(torch.randn(n, k, generator=g) * k ** -0.5); bias = torch.randn(n, generator=g)
        r = _rt(torch.randn(m, n, generator=g))
        got = ops.gemm(_bf(a).cuda(), _bf(w).cuda(), bias.cuda(), residual=_bf(r).cuda(), out_scale=0.18)
        _close(got, (a @ w.t() + bias) * 0.18 + r, 2.0 ** -7, f"gemm out_scale {m}x{n}")
    # QKV: only the q segment is scaled
    for nb, rows, dim, k in ((2, 64, 128, 128), (4, 1024, 1024, 128)):
        a = _rt(torch.randn(nb * rows, k, generator=g)); w = _rt(torch.randn(3 * dim, k, generator=g) * k ** -0.5)
        full = (a @ w.t()).reshape(nb * rows, 3, dim)
        c, vt = ops.gemm_qkv(_bf(a).cuda(), _bf(w).cuda(), dim, 3, rows, q_scale=0.18)
        want = torch.cat([full[:, 0] * 0.18, full[:, 1]], dim=1)
        _close(c, want, 2.0 ** -7, f"qkv q_scale {dim}")
        _close(ops.unpack_vt(vt, rows), full[:, 2].reshape(nb, rows, dim), 2.0 ** -7, "qkv q_scale leaves V alone")


# ----------------------------------------------------------------------------------------------------------------------
# channel concatenation read in place (up blocks: torch.cat([hidden, skip], dim=1) feeding norm1 and the 1x1 shortcut)
# ----------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("c1,c2,hw,patch", [(64, 64, 16, 0), (1280, 640, 8, 0), (320, 640, 16, 8), (640, 320, 32, 0)])
def test_groupnorm_cat(cuda_device, c1, c2, hw, patch):
    """groups that straddle the boundary between the two sources included (1920 / 32 = 60 channels per group, split at 1280)"""
    from sduss_amd import lib
    l = lib.load()
    g = torch.Generator().manual_seed(c1 + c2 + hw)
    c = c1 + c2
    x = _rt(torch.randn(2, c, hw, hw, generator=g) * 1.5 + 0.7); ga = torch.randn(c, generator=g); be = torch.randn(c, generator=g)
    want = F.silu(ref._gn(x, 32, ga, be, 1e-5, patch if patch else None))
    xa = _bf(_nhwc(x[:, :c1])).cuda(); xb = _bf(_nhwc(x[:, c1:])).cuda()
    y = torch.empty(2, hw, hw, c, dtype=torch.bfloat16, device="cuda")
    ws = torch.empty(l.mx_groupnorm_nhwc_workspace_bytes(2, hw, hw, c), dtype=torch.uint8, device="cuda")
    gg, bb = ga.cuda(), be.cuda()
    lib.check(l.mx_groupnorm_nhwc_cat(lib.current_stream(), xa.data_ptr(), c1, xb.data_ptr(), y.data_ptr(), gg.data_ptr(), bb.data_ptr(),
                                      2, hw, hw, c, 32, 1e-5, 1, patch, ws.data_ptr()), "mx_groupnorm_nhwc_cat")
    _close(y.permute(0, 3, 1, 2), want, 2.0 ** -7, f"groupnorm cat {c1}+{c2}")


@pytest.mark.parametrize("m,n,k1,k2", [(300, 320, 128, 64), (8192, 1280, 1280, 640), (2048, 640, 640, 640), (1000, 192, 64, 192)])
def test_gemm_split_a_operand(cuda_device, m, n, k1, k2):
    """A = [a | a2] along K read in place (generic kernel and the 256 / 128-row pipelined kernels)"""
    from sduss_amd import lib
    l = lib.load()
    g = torch.Generator().manual_seed(m + n + k1)
    a1 = _rt(torch.randn(m, k1, generator=g)); a2 = _rt(torch.randn(m, k2, generator=g))
    w = _rt(torch.randn(n, k1 + k2, generator=g) * (k1 + k2) ** -0.5); bias = torch.randn(n, generator=g)
    a1g, a2g, wg, bg = _bf(a1).cuda(), _bf(a2).cuda(), _bf(w).cuda(), bias.cuda()
    out = torch.empty(m, n, dtype=torch.bfloat16, device="cuda")
    d = lib.GemmDesc()
    d.a, d.w, d.c, d.bias, d.a2 = a1g.data_ptr(), wg.data_ptr(), out.data_ptr(), bg.data_ptr(), a2g.data_ptr()
    d.M, d.N, d.K, d.lda, d.ldc, d.lda2, d.k_split = m, n, k1 + k2, k1, n, k2, k1
    lib.check(l.mx_gemm(lib.current_stream(), C.byref(d)), "mx_gemm split A")
    _close(out, torch.cat([a1, a2], dim=1) @ w.t() + bias, 2.0 ** -7, f"gemm split A {m}x{n}x({k1}+{k2})")


# ---------------------------------------------------------------------------------------------------------------------------------------
# split-K (small launches: 128-row tiles in several K slices, combined by the last arriver in slice order; gemm_bf16_v2.hip)
# ---------------------------------------------------------------------------------------------------------------------------------------
def _splitk_of(d, conv=0):
    from sduss_amd import lib
    return lib.load().mx_gemm_splitk(C.byref(d), conv)


@pytest.mark.parametrize("m,n,k", [(2048, 1280, 1280), (2048, 1280, 5120), (1024, 640, 2560), (1856, 1280, 1280), (512, 1280, 1280)])
def test_gemm_splitk_parity_and_run_to_run_stability(cuda_device, m, n, k):
    """one-request shapes of the SDXL step (M = 2048 tokens x N 1280, K 1280 / 5120; a mixed 512 + 768 px level: M 1856): the launch is SPLIT
    in two (forced through mx_gemm_desc.splitk), equals the fp32 reference within one bf16 rounding, and is bit-identical over repeated launches although the
    workgroup that combines the slices changes from run to run (the partial tiles are always added in slice order)"""
    from sduss_amd import lib, ops
    g = torch.Generator().manual_seed(m + n + k)
    a = _rt(torch.randn(m, k, generator=g)); w = _rt(torch.randn(n, k, generator=g) * k ** -0.5)
    b = torch.randn(n, generator=g); r = _rt(torch.randn(m, n, generator=g))
    d = lib.GemmDesc()
    d.M, d.N, d.K, d.lda, d.ldc, d.ldr = m, n, k, k, n, n
    d.a = d.w = d.c = d.residual = 4096
    d.splitk = 2                                # forced: the library's own choice splits only where the partial tiles' traffic pays (long K)
    assert _splitk_of(d) == 2, "this shape is eligible for split-K"
    want = a @ w.t() + b + r
    ad, wd, bd, rd = _bf(a).cuda(), _bf(w).cuda(), b.cuda(), _bf(r).cuda()
    first = ops.gemm(ad, wd, bd, residual=rd, splitk=2)
    _close(first, want, 2.0 ** -7, f"split-K gemm {m}x{n}x{k}")
    _close(ops.gemm(ad, wd, bd, residual=rd, splitk=1), want, 2.0 ** -7, f"unsplit gemm {m}x{n}x{k}")
    for _ in range(6):
        again = ops.gemm(ad, wd, bd, residual=rd, splitk=2)
        assert torch.equal(again.view(torch.int16), first.view(torch.int16)), "split-K result changed from one launch to the next"


def test_gemm_splitk_automatic_choice_is_shape_based(cuda_device):
    """the library's own choice: long K at small M x N splits (ff.net.2 of one request), K 1280 does not (measured slower: the fp32 partial
    tiles' round trip through memory costs more than the shorter K loop saves, profiles/r04_g_splitk_bench.txt)"""
    from sduss_amd import lib
    d = lib.GemmDesc()
    d.a = d.w = d.c = 4096
    d.M, d.N, d.K, d.lda, d.ldc = 512, 1280, 5120, 5120, 1280
    assert _splitk_of(d) >= 2                   # measured 39.7 -> 20.4 us
    d.M = 2048
    assert _splitk_of(d) == 1                   # measured 41.3 -> 39.7 us: below the 25 % margin
    d.K = d.lda = 1280
    assert _splitk_of(d) == 1                   # measured SLOWER split (16.4 -> 20.8 us)
    d.M, d.K, d.lda = 8192, 5120, 5120
    assert _splitk_of(d) == 1


def test_gemm_splitk_with_folded_layernorm_and_stats(cuda_device):
    """to_q of a single request: LayerNorm folded into a split launch (the -mean * colsum term enters through slice 0 only) that also produces
    the row statistics of its output"""
    from sduss_amd import ops
    from sduss_amd.weights import fold_layernorm
    g = torch.Generator().manual_seed(77)
    m, c, n = 2048, 1280, 1280
    x = _rt(torch.randn(m, c, generator=g) * (0.5 + 1.5 * torch.rand(m, 1, generator=g)) + 3.0 * torch.randn(m, 1, generator=g))
    gamma, beta = 1.0 + 0.2 * torch.randn(c, generator=g), 0.1 * torch.randn(c, generator=g)
    w = _rt(torch.randn(n, c, generator=g) * c ** -0.5); bias = 0.1 * torch.randn(n, generator=g)
    want = F.layer_norm(x, (c,), gamma, beta, 1e-5) @ w.t() + bias
    wf, colsum, bf_ = fold_layernorm(w, bias, gamma, beta)
    st = ops.row_stats(_bf(x).cuda())
    got, (stats, slabs) = ops.gemm(_bf(x).cuda(), wf.cuda(), bf_.cuda(), ln_stats=st, ln_colsum=colsum.cuda(), ln_eps=1e-5, want_stats=True, splitk=2)
    _close(got, want, 2.0 ** -7, "split-K ln-folded linear")
    s = stats[:, :slabs].sum(dim=1).cpu()
    _close(s[:, 0], want.sum(dim=1), 2e-3, "row sums from the split launch's epilogue")


def test_conv3x3_splitk(cuda_device):
    """the 1280-channel conv of one request at the 32 x 32 level (M 2048, K 11520): split over the taps"""
    from sduss_amd import lib, ops
    g = torch.Generator().manual_seed(5)
    b, hw, cin, cout = 2, 32, 1280, 1280
    x = _rt(torch.randn(b, cin, hw, hw, generator=g)); w = _rt(torch.randn(cout, cin, 3, 3, generator=g) * (9 * cin) ** -0.5)
    bias = torch.randn(cout, generator=g)
    d = lib.GemmDesc()
    d.M, d.N, d.K, d.ldc = b * hw * hw, cout, 9 * cin, cout
    d.a = d.w = d.c = 4096
    d.B, d.Hin, d.Win, d.Cin, d.Hout, d.Wout, d.stride = b, hw, hw, cin, hw, hw, 1
    d.splitk = 3
    assert _splitk_of(d, 1) == 3
    want = F.conv2d(x, w, bias, padding=1)
    got = ops.conv3x3(_bf(_nhwc(x)).cuda(), _bf(_conv_pack(w)).cuda(), bias.cuda(), splitk=3)
    _close(got.permute(0, 3, 1, 2), want, 2.0 ** -7, "split-K conv3x3")
    again = ops.conv3x3(_bf(_nhwc(x)).cuda(), _bf(_conv_pack(w)).cuda(), bias.cuda(), splitk=3)
    assert torch.equal(again.view(torch.int16), got.view(torch.int16))


def test_gemm_tail_split_of_the_256x256_kernel(cuda_device):
    """One 1024 px request's GEGLU projection (M 2048, N 10240, K 1280 = 320 tiles of 256 x 256 = 1.25 rounds): mx_gemm cuts it along N into the whole
    round on the 256 x 256 kernel and ONE round of a smaller tile (mx_gemm_launches == 2).  The result is that of LayerNorm-free GEGLU in fp32 and is bit
    for bit what the same rows give inside the headline batch's launch (M 8192: whole rounds, one launch): no tiling changes a row's summation order."""
    import ctypes as C
    from sduss_amd import lib, ops
    from sduss_amd.weights import _geglu_interleave
    from test_headline_shapes_gpu import _records
    g = torch.Generator().manual_seed(2048)
    m, dim = 2048, 1280
    a = _rt(torch.randn(m, dim, generator=g)); w = _rt(torch.randn(8 * dim, dim, generator=g) * dim ** -0.5); b = torch.randn(8 * dim, generator=g)
    hid, gate = (a @ w.t() + b).chunk(2, dim=-1)
    want = hid * F.gelu(gate)
    ag, wg, bg = _bf(a).cuda(), _bf(_geglu_interleave(w)).cuda(), _geglu_interleave(b).cuda()
    out = {}
    rec = _records(lambda: out.__setitem__("o", ops.gemm(ag, wg, bg, geglu=True)))
    assert [r[0] for r in rec].count("gemm_256x256") == 1 and len(rec) == 2, rec
    assert sum(r[2] for r in rec) == 8 * dim and rec[0][2] == 8192, rec          # columns [0, 8192) in whole rounds, the rest on the smaller tile
    _close(out["o"], want, 2.0 ** -7, "tail-split geglu")
    big = ops.gemm(ag.repeat(4, 1), wg, bg, geglu=True)                          # M 8192: 1280 tiles = 5 whole rounds, one launch
    assert torch.equal(big[:m], out["o"]) and torch.equal(big[3 * m:], out["o"])
    # plain epilogue with a residual, M 4096 x N 10240: 640 tiles = 2.5 rounds -> 2 rounds + one round of 256 x 128
    r = _rt(torch.randn(2 * m, 8 * dim, generator=g))
    a2 = _rt(torch.randn(2 * m, dim, generator=g))
    d = lib.GemmDesc(); d.M, d.N, d.K, d.lda, d.ldc, d.ldr = 2 * m, 8 * dim, dim, dim, 8 * dim, 8 * dim
    d.a = d.w = d.c = d.residual = 256
    n_launch = lib.load().mx_gemm_launches(C.byref(d))
    got = ops.gemm(_bf(a2).cuda(), _bf(w).cuda(), b.cuda(), residual=_bf(r).cuda())
    _close(got, a2 @ w.t() + b + r, 2.0 ** -7, f"tail-split plain + residual ({n_launch} launches)")
    assert n_launch == 2


@pytest.mark.parametrize("b,hw,cin,cout", [(8, 32, 1280, 1280), (8, 32, 640, 1280), (2, 64, 640, 640), (4, 64, 320, 640), (3, 24, 1280, 1280),
                                           (9, 24, 640, 1280)])     # 5184 rows: the last 256-row tile is a quarter full
def test_groupnorm_from_the_producing_convs_partial_sums(cuda_device, b, hw, cin, cout):
    """resnet conv1 (+ bias + time-embedding row bias) leaves, per 64 output rows and channel, the sum and sum of squares of what it stores
    (mx_gemm_desc.gn_part_out: sums of its accumulators, the constants bias + time embedding are added back by the fold); norm2 + SiLU from those partials equals
    GroupNorm + SiLU of the stored tensor (resnet.py:414-429) and the statistics pass is gone;
    where the launch cannot (a small batch on 128-row tiles) the caller is told (None)."""
    from sduss_amd import ops
    g = torch.Generator().manual_seed(b + hw + cin + cout)
    x = _rt(torch.randn(b, cin, hw, hw, generator=g)); w = _rt(torch.randn(cout, cin, 3, 3, generator=g) * (9 * cin) ** -0.5)
    bias = torch.randn(cout, generator=g); temb = torch.randn(b, cout, generator=g)
    ga = 1.0 + 0.2 * torch.randn(cout, generator=g); be = 0.1 * torch.randn(cout, generator=g)
    y, part = ops.conv3x3(_bf(_nhwc(x)).cuda(), _bf(_conv_pack(w)).cuda(), bias.cuda(), rowbias=temb.cuda(), want_gn_partials=True)
    if part is None:                                       # a small launch takes 128-row tiles: no partial sums, the plan keeps the statistics pass there
        assert b * hw * hw <= 8192, "a chip-filling conv must be able to leave its partial sums"
        return
    yf = y.float().cpu()                                   # [b, hw, hw, cout]
    rows = (yf - bias - temb[:, None, None, :]).reshape(-1, 64, cout)      # what the accumulators held (to the output's bf16 rounding)
    pc = part.cpu()
    assert torch.isfinite(pc).all()
    s_ref, q_ref = rows.sum(dim=1), (rows * rows).sum(dim=1)
    assert (pc[..., 0] - s_ref).abs().max().item() <= 2.0 ** -7 * yf.abs().reshape(-1, 64, cout).sum(dim=1).max().item()      # fp32 sums before the bf16 rounding vs sums of the rounded values
    assert ((pc[..., 1] - q_ref).abs() / (q_ref + 1.0)).max().item() <= 2.0 ** -5
    want = F.silu(F.group_norm(yf.permute(0, 3, 1, 2), 32, ga, be, 1e-5))
    got = ops.groupnorm_nhwc_from_partials(y, ga.cuda(), be.cuda(), 32, 1e-5, True, part, add_bias=bias.cuda(), add_rowbias=temb.cuda())
    _close(got.permute(0, 3, 1, 2), want, 2.0 ** -7, f"groupnorm from the conv's partial sums b{b} {hw}x{hw} {cin}->{cout}")
    again = ops.groupnorm_nhwc_from_partials(y, ga.cuda(), be.cuda(), 32, 1e-5, True, ops.conv3x3(_bf(_nhwc(x)).cuda(), _bf(_conv_pack(w)).cuda(), bias.cuda(), rowbias=temb.cuda(), want_gn_partials=True)[1],
                                             add_bias=bias.cuda(), add_rowbias=temb.cuda())
    assert torch.equal(again, got)                         # fixed summation order: bit-stable run to run
