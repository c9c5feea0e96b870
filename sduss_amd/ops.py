"""Thin torch-tensor wrappers over the per-kernel entry points of the C ABI (mx_gemm, mx_conv3x3, mx_attention,
mx_layernorm, mx_groupnorm_nhwc, scheduler steps).  They allocate outputs with torch and pass raw pointers + the current
stream; nothing is computed in Python.  Used by the parity tests and by ``pipeline.py``."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import lib as _lib


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _bf16(t):
    assert t.is_cuda and t.dtype == torch.bfloat16 and t.is_contiguous(), "expected a contiguous CUDA bf16 tensor"
    return t


def gemm(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None,
         rowbias: Optional[torch.Tensor] = None, rows_per_batch: int = 0, silu: bool = False, geglu: bool = False,
         out_f32: bool = False, out_scale: float = 0.0, ln_stats: Optional[torch.Tensor] = None, ln_colsum: Optional[torch.Tensor] = None,
         ln_eps: float = 1e-5, want_stats: bool = False, splitk: int = 0, ln_final: Optional[torch.Tensor] = None, want_final: bool = False,
         final_buffers=None):
    """C[M,N] = (A[M,K] W[N,K]^T + bias) * out_scale (+rowbias +residual, silu | geglu).  With ``geglu`` the weight/bias rows must be
    interleaved as weights._geglu_interleave does; the output is [M, N/2].
    ``ln_stats`` = (stats [M, pitch, 2], slabs) + ``ln_colsum`` [N]: LayerNorm folded into this GEMM (w, bias folded by
    weights.fold_layernorm).  ``want_stats``: also return the row statistics of C as such a pair (from the epilogue when the kernel can,
    else mx_row_stats); entries past ``slabs`` of a row are not initialised.
    ``want_final`` (with ``want_stats``): also return the FINALISED statistics [M, 2] = (mean, rstd with ``ln_eps``) the launch's last workgroup per
    256-row panel leaves (mx_gemm_desc.ln_final_out), or None where the launch cannot (mx_gemm_ln_final_supported); ``ln_final`` + ``ln_colsum``:
    the consumer side (the 256 x 256 kernel's form of the folded LayerNorm).  ``final_buffers`` = (final [M, 2] fp32, tickets [ceil(M / 256)] int32,
    zero): reuse these instead of allocating (and skip the host-side check of the tickets, which synchronises)."""
    l = _lib.load()
    _bf16(a); _bf16(w)
    m, k = a.shape
    n = w.shape[0]
    d = _lib.GemmDesc()
    flags = (_lib.EPI_SILU if silu else 0) | (_lib.EPI_GEGLU if geglu else 0) | (_lib.EPI_OUT_F32 if out_f32 else 0)
    nout = n // 2 if geglu else n
    c = torch.empty((m, nout), dtype=torch.float32 if out_f32 else torch.bfloat16, device=a.device)
    d.a, d.w, d.c = a.data_ptr(), w.data_ptr(), c.data_ptr()
    d.bias, d.rowbias, d.residual = _p(bias), _p(rowbias), _p(residual)
    d.M, d.N, d.K, d.lda, d.ldc = m, n, k, k, nout
    d.ldr = residual.shape[1] if residual is not None else 0
    d.ldrb = rowbias.shape[1] if rowbias is not None else 0
    d.rows_per_batch, d.flags, d.out_scale = rows_per_batch, flags, out_scale
    d.splitk = splitk                           # 0 auto, 1 never, 2..4 forced (mx_gemm_desc.splitk)
    if ln_stats is not None:
        st_, slabs_ = ln_stats
        assert st_.dtype == torch.float32 and ln_colsum.dtype == torch.float32 and st_.shape == (m, stats_pitch(slabs_), 2)
        d.ln_stats, d.ln_colsum, d.ln_slabs, d.ln_eps = st_.data_ptr(), ln_colsum.data_ptr(), slabs_, ln_eps
    if ln_final is not None:
        assert ln_final.dtype == torch.float32 and ln_final.shape == (m, 2) and ln_colsum.dtype == torch.float32
        d.ln_final, d.ln_colsum, d.ln_eps = ln_final.data_ptr(), ln_colsum.data_ptr(), ln_eps
    stats = None
    final = None
    if want_stats:
        slabs = l.mx_gemm_stats_slabs(C.byref(d))
        stats = torch.full((m, stats_pitch(max(slabs, 1)), 2), float("nan"), dtype=torch.float32, device=a.device)
        if slabs > 0:
            d.stats_out = stats.data_ptr()
            if want_final and l.mx_gemm_ln_final_supported(C.byref(d)):
                if final_buffers is not None:
                    final, cnt = final_buffers
                else:
                    final = torch.full((m, 2), float("nan"), dtype=torch.float32, device=a.device)
                    cnt = torch.zeros((m + 255) // 256, dtype=torch.int32, device=a.device)
                d.ln_final_out, d.ln_final_cnt, d.ln_eps = final.data_ptr(), cnt.data_ptr(), ln_eps
    _lib.check(l.mx_gemm(_lib.current_stream(), C.byref(d)), "mx_gemm")
    if want_stats:
        if not d.stats_out:
            _lib.check(l.mx_row_stats(_lib.current_stream(), c.data_ptr(), nout, stats.data_ptr(), m, nout), "mx_row_stats")
        if want_final:
            if final is not None and final_buffers is None:
                assert int(cnt.abs().sum()) == 0, "ln_final_cnt must be zero again after the launch"
            return c, (stats, max(slabs, 1)), final
        return c, (stats, max(slabs, 1))
    return c


def stats_pitch(slabs: int) -> int:
    """MX_STATS_PITCH: slabs per row of a statistics buffer."""
    return (slabs + 3) & ~3


def row_stats(x: torch.Tensor) -> torch.Tensor:
    """(sum, sum of squares) of every row of a bf16 [M, C] matrix -> (fp32 [M, 4, 2] with slab 0 filled, 1): the one-slab ``ln_stats``."""
    l = _lib.load()
    _bf16(x)
    m, c = x.shape
    st = torch.full((m, 4, 2), float("nan"), dtype=torch.float32, device=x.device)
    _lib.check(l.mx_row_stats(_lib.current_stream(), x.data_ptr(), c, st.data_ptr(), m, c), "mx_row_stats")
    return st, 1


def vt_ld(lk: int) -> int:
    """MX_VT_LD: row length of a V^T image for lk keys."""
    return (lk + 15) // 16 * 16


def vt_pos(lk: int) -> torch.Tensor:
    """MX_VT_POS for keys 0..lk-1 (bits 2 and 3 of the key index swapped; its own inverse)."""
    k = torch.arange(lk)
    return (k & ~12) | ((k & 4) << 1) | ((k & 8) >> 1)


def pack_vt(v: torch.Tensor, pad: float = 0.0) -> torch.Tensor:
    """v [B, Lk, C] -> V^T image [B, C, MX_VT_LD(Lk)] in the attention kernel's key order (include/mxdenoise.h)."""
    b, lk, c = v.shape
    vt = torch.full((b, c, vt_ld(lk)), pad, dtype=v.dtype, device=v.device)
    vt[:, :, vt_pos(lk).to(v.device)] = v.permute(0, 2, 1)
    return vt


def unpack_vt(vt: torch.Tensor, lk: int) -> torch.Tensor:
    """inverse of pack_vt: [B, C, ld] -> [B, Lk, C]."""
    return vt[:, :, vt_pos(lk).to(vt.device)].permute(0, 2, 1)


def gemm_qkv(a: torch.Tensor, w: torch.Tensor, seg: int, period: int, rows_per_batch: int, q_scale: float = 0.0,
             ln_stats: Optional[torch.Tensor] = None, ln_colsum: Optional[torch.Tensor] = None, ln_eps: float = 1e-5,
             bias: Optional[torch.Tensor] = None, ln_final: Optional[torch.Tensor] = None):
    """Fused projection with the V segments written transposed.  Returns (c [M, N/period*(period-1)],
    vt [M/rows_per_batch, N/period, ldvt] in MX_VT_POS key order; see unpack_vt)."""
    l = _lib.load()
    _bf16(a); _bf16(w)
    m, k = a.shape
    n = w.shape[0]
    nb = m // rows_per_batch
    ldvt = vt_ld(rows_per_batch)
    c = torch.empty((m, n // period * (period - 1)), dtype=torch.bfloat16, device=a.device)
    vt = torch.zeros((nb, n // period, ldvt), dtype=torch.bfloat16, device=a.device)
    d = _lib.GemmDesc()
    d.a, d.w, d.c, d.vt = a.data_ptr(), w.data_ptr(), c.data_ptr(), vt.data_ptr()
    d.M, d.N, d.K, d.lda, d.ldc = m, n, k, k, c.shape[1]
    d.rows_per_batch, d.flags, d.seg, d.period, d.ldvt = rows_per_batch, _lib.EPI_QKV, seg, period, ldvt
    d.out_scale = q_scale                       # scales the q segment only (mx_attention_prescaled)
    d.bias = _p(bias)
    if ln_stats is not None:
        d.ln_stats, d.ln_colsum, d.ln_slabs, d.ln_eps = ln_stats[0].data_ptr(), ln_colsum.data_ptr(), ln_stats[1], ln_eps
    if ln_final is not None:
        d.ln_final, d.ln_colsum, d.ln_eps = ln_final.data_ptr(), ln_colsum.data_ptr(), ln_eps
    _lib.check(l.mx_gemm(_lib.current_stream(), C.byref(d)), "mx_gemm(qkv)")
    return c, vt


def conv3x3(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], stride: int = 1, up: int = 0,
            corner_patch: int = 0, rowbias: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None, splitk: int = 0, want_gn_partials: bool = False,
            cin_valid: int = 0):
    """x NHWC bf16 [B,H,W,Cin]; w bf16 [Cout, 9*Cin] tap-major; returns NHWC bf16 [B,Ho,Wo,Cout].  ``cin_valid`` > 0: the caller's promise that only the first
    cin_valid (<= 8) channels of x are non-zero (mx_gemm_desc.cin_valid: conv_in's zero-padded latent).  ``want_gn_partials``: also return the per-64-row,
    per-channel (sum, sum of squares) of its ACCUMULATORS (the output minus bias and row bias) [M / 64, Cout, 2] (mx_gemm_desc.gn_part_out), or None where it cannot."""
    l = _lib.load()
    _bf16(x); _bf16(w)
    b, h, wd, cin = x.shape
    cout = w.shape[0]
    hv, wv = h << up, wd << up
    ho, wo = (hv + stride - 1) // stride, (wv + stride - 1) // stride
    c = torch.empty((b, ho, wo, cout), dtype=torch.bfloat16, device=x.device)
    d = _lib.GemmDesc()
    d.a, d.w, d.c, d.bias, d.rowbias, d.residual = x.data_ptr(), w.data_ptr(), c.data_ptr(), _p(bias), _p(rowbias), _p(residual)
    d.M, d.N, d.K, d.ldc, d.ldr = b * ho * wo, cout, 9 * cin, cout, cout
    d.ldrb = rowbias.shape[1] if rowbias is not None else 0
    d.rows_per_batch = ho * wo
    d.B, d.Hin, d.Win, d.Cin, d.Hout, d.Wout, d.stride, d.up, d.corner_patch = b, h, wd, cin, ho, wo, stride, up, corner_patch
    d.splitk = splitk
    d.cin_valid = cin_valid
    part = None
    if want_gn_partials and l.mx_gemm_gn_partials_supported(C.byref(d), 1):
        part = torch.full((b * ho * wo // 64, cout, 2), float("nan"), dtype=torch.float32, device=x.device)
        d.gn_part_out = part.data_ptr()
    _lib.check(l.mx_conv3x3(_lib.current_stream(), C.byref(d)), "mx_conv3x3")
    if want_gn_partials:
        return c, part
    return c


ATTN_QSCALE = 0.125 * 1.4426950408889634     # MX_ATTN_QSCALE(1/sqrt(64))


def attention(q: torch.Tensor, k: torch.Tensor, vt: torch.Tensor, heads: int, lq: int, lk: int, prescaled: bool = False) -> torch.Tensor:
    """q [B*Lq, H*64], k [B*Lk, H*64], vt [B, H*64, ldvt] (all bf16) -> o [B*Lq, H*64].  ``prescaled``: q already carries
    the factor ATTN_QSCALE (mx_attention_prescaled)."""
    l = _lib.load()
    _bf16(q); _bf16(k); _bf16(vt)
    b = vt.shape[0]
    o = torch.empty_like(q)
    if prescaled:
        _lib.check(l.mx_attention_prescaled(_lib.current_stream(), q.data_ptr(), q.shape[1], k.data_ptr(), k.shape[1], vt.data_ptr(),
                                            vt.shape[2], vt.shape[1] * vt.shape[2], o.data_ptr(), o.shape[1], b, heads, lq, lk),
                   "mx_attention_prescaled")
        return o
    _lib.check(l.mx_attention(_lib.current_stream(), q.data_ptr(), q.shape[1], k.data_ptr(), k.shape[1], vt.data_ptr(),
                              vt.shape[2], vt.shape[1] * vt.shape[2], o.data_ptr(), o.shape[1], b, heads, lq, lk,
                              0.125), "mx_attention")
    return o


def attn_tail(ao: torch.Tensor, y: torch.Tensor, w1: torch.Tensor, b1: torch.Tensor, wq: torch.Tensor, bq: torch.Tensor, colsum_q: torch.Tensor,
              k: torch.Tensor, vt: torch.Tensor, w2: torch.Tensor, b2: torch.Tensor, heads: int, L: int, ctx_len: int, ln_eps: float = 1e-5,
              chained: bool = True, finalise: bool = True, sync: Optional[torch.Tensor] = None):
    """The attention tail of a BasicTransformerBlock on the hidden state ``y`` [M, C] (a copy is updated and returned):
    y1 = ao w1^T + b1 + y;  q2 = norm2(y1) wq^T + bq (wq / bq / colsum_q LayerNorm-folded, prescaled);  ao2 = cross-attention(q2, k, vt);
    y2 = ao2 w2^T + b2 + y1.  ``chained`` = ONE launch (mx_attn_tail); else the four launches with the SAME descriptors (the results must be equal bit
    for bit).  k [B * ctx_len, ldk >= C], vt [B, C, MX_VT_LD(ctx_len)].  Returns (y2, (slab statistics, slabs) of y2, finalised statistics [M, 2] or None,
    q2, ao2)."""
    l = _lib.load()
    for t_ in (ao, y, w1, wq, w2, k, vt):
        _bf16(t_)
    m, c = y.shape
    b = m // L
    dev = y.device
    y = y.clone()
    q2 = torch.empty_like(y)
    ao2 = torch.empty_like(y)
    td = _lib.AttnTailDesc()

    def lin(d, a_, w_, bias, c_, res):
        d.a, d.w, d.c, d.bias = a_.data_ptr(), w_.data_ptr(), c_.data_ptr(), bias.data_ptr()
        d.M, d.N, d.K, d.lda, d.ldc = m, c, c, c, c
        if res is not None:
            d.residual, d.ldr = res.data_ptr(), c
    lin(td.out1, ao, w1, b1, y, y)
    lin(td.to_q, y, wq, bq, q2, None)
    lin(td.out2, ao2, w2, b2, y, y)
    td.out1.stats_out = 16                       # (shape query below: which operands exist)
    slabs = l.mx_gemm_stats_slabs(C.byref(td.out1))
    assert slabs > 0, "out1 cannot write slab statistics at this shape"
    st1 = torch.full((m, stats_pitch(slabs), 2), float("nan"), dtype=torch.float32, device=dev)
    st2 = torch.full((m, stats_pitch(slabs), 2), float("nan"), dtype=torch.float32, device=dev)
    td.out1.stats_out = st1.data_ptr()
    td.to_q.ln_stats, td.to_q.ln_colsum, td.to_q.ln_slabs, td.to_q.ln_eps = st1.data_ptr(), colsum_q.data_ptr(), slabs, ln_eps
    td.to_q.out_scale = ATTN_QSCALE
    td.out2.stats_out = st2.data_ptr()
    final = cnt = None
    if finalise:
        final = torch.full((m, 2), float("nan"), dtype=torch.float32, device=dev)
        cnt = torch.zeros((m + 255) // 256, dtype=torch.int32, device=dev)
        td.out2.ln_final_out, td.out2.ln_final_cnt, td.out2.ln_eps = final.data_ptr(), cnt.data_ptr(), ln_eps
    td.k, td.ldk, td.vt, td.ldvt, td.vt_batch_stride = k.data_ptr(), k.shape[1], vt.data_ptr(), vt.shape[2], vt.shape[1] * vt.shape[2]
    td.B, td.heads, td.L, td.ctx_len = b, heads, L, ctx_len
    stream = _lib.current_stream()
    if chained:
        if sync is None:
            sync = torch.zeros(l.mx_attn_tail_sync_bytes(m) // 4, dtype=torch.int32, device=dev)
        td.sync = sync.data_ptr()
        assert l.mx_attn_tail_supported(C.byref(td)), "mx_attn_tail cannot serve this shape"
        _lib.check(l.mx_attn_tail(stream, C.byref(td)), "mx_attn_tail")
    else:
        _lib.check(l.mx_gemm(stream, C.byref(td.out1)), "mx_gemm")
        _lib.check(l.mx_gemm(stream, C.byref(td.to_q)), "mx_gemm")
        _lib.check(l.mx_attention_cross_prescaled(stream, q2.data_ptr(), c, k.data_ptr(), k.shape[1], vt.data_ptr(), vt.shape[2], vt.shape[1] * vt.shape[2],
                                                  ao2.data_ptr(), c, b, heads, L, ctx_len), "mx_attention_cross_prescaled")
        _lib.check(l.mx_gemm(stream, C.byref(td.out2)), "mx_gemm")
    return y, (st2, slabs), final, q2, ao2, sync


def attn_tail_status(sync: torch.Tensor) -> int:
    """the error word of an mx_attn_tail sync buffer (0 = every wait ended); synchronises"""
    l = _lib.load()
    w = C.c_uint(0)
    _lib.check(l.mx_attn_tail_status(_lib.current_stream(), sync.data_ptr(), C.byref(w)), "mx_attn_tail_status")
    return w.value


def layernorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    l = _lib.load()
    _bf16(x)
    y = torch.empty_like(x)
    _lib.check(l.mx_layernorm(_lib.current_stream(), x.data_ptr(), y.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                              x.shape[0], x.shape[1], eps), "mx_layernorm")
    return y


def groupnorm_nhwc_from_partials(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, groups: int, eps: float, silu: bool, part: torch.Tensor,
                                 chunk: int = 64, add_bias: Optional[torch.Tensor] = None, add_rowbias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """GroupNorm (+SiLU) of x NHWC from the partial sums the producing launch left (conv3x3(..., want_gn_partials=True): sums of its accumulators, i.e. of
    x minus ``add_bias`` [C] + ``add_rowbias`` [B, ld] -- pass the conv's bias and row bias): no statistics pass."""
    l = _lib.load()
    _bf16(x)
    b, h, w, c = x.shape
    y = torch.empty_like(x)
    ws = torch.empty(l.mx_groupnorm_nhwc_workspace_bytes(b, h, w, c), dtype=torch.uint8, device=x.device)
    _lib.check(l.mx_groupnorm_nhwc_from_partials(_lib.current_stream(), x.data_ptr(), y.data_ptr(), gamma.data_ptr(), beta.data_ptr(), b, h, w, c, groups, eps,
                                                 1 if silu else 0, part.data_ptr(), chunk, _p(add_bias), _p(add_rowbias),
                                                 add_rowbias.shape[1] if add_rowbias is not None else 0, ws.data_ptr()), "mx_groupnorm_nhwc_from_partials")
    return y


def groupnorm_nhwc(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, groups: int, eps: float, silu: bool,
                   patch: int = 0) -> torch.Tensor:
    l = _lib.load()
    _bf16(x)
    b, h, w, c = x.shape
    y = torch.empty_like(x)
    ws = torch.empty(l.mx_groupnorm_nhwc_workspace_bytes(b, h, w, c), dtype=torch.uint8, device=x.device)
    _lib.check(l.mx_groupnorm_nhwc(_lib.current_stream(), x.data_ptr(), y.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                   b, h, w, c, groups, eps, int(silu), patch, ws.data_ptr()), "mx_groupnorm_nhwc")
    return y


def euler_scale_input(latents: torch.Tensor, sigma: torch.Tensor, n_rows: int) -> torch.Tensor:
    """[n_lat, ...] -> [n_rows, ...] = latents[r % n_lat] / sqrt(sigma^2+1)  (CFG duplication fused)."""
    l = _lib.load()
    latents = latents.contiguous()
    n_lat = latents.shape[0]
    elems = latents[0].numel()
    out = torch.empty((n_rows, *latents.shape[1:]), dtype=latents.dtype, device=latents.device)
    sg = sigma.to(device=latents.device, dtype=torch.float32).contiguous()
    _lib.check(l.mx_euler_scale_input(_lib.current_stream(), latents.data_ptr(), out.data_ptr(), sg.data_ptr(), n_lat, n_rows,
                                      elems, _lib.torch_dtype_code(latents.dtype)), "mx_euler_scale_input")
    return out


def cfg_euler_step_(noise: torch.Tensor, latents: torch.Tensor, sigma: torch.Tensor, sigma_next: torch.Tensor,
                    guidance_scale: float) -> torch.Tensor:
    """In place: latents <- Euler step with eps = u + g (t - u); noise = [uncond rows ; cond rows] (g > 0)
    or the plain prediction (g <= 0)."""
    l = _lib.load()
    assert latents.is_contiguous() and noise.is_contiguous() and noise.dtype == latents.dtype
    n_lat = latents.shape[0]
    sg = sigma.to(device=latents.device, dtype=torch.float32).contiguous()
    sn = sigma_next.to(device=latents.device, dtype=torch.float32).contiguous()
    _lib.check(l.mx_cfg_euler_step(_lib.current_stream(), noise.data_ptr(), latents.data_ptr(), sg.data_ptr(), sn.data_ptr(),
                                   float(guidance_scale), n_lat, latents[0].numel(), _lib.torch_dtype_code(latents.dtype)),
               "mx_cfg_euler_step")
    return latents


def layernorm_mod(x: torch.Tensor, scale: torch.Tensor, shift: torch.Tensor, rows_per_batch: int, eps: float = 1e-6,
                  scale2: Optional[torch.Tensor] = None, shift2: Optional[torch.Tensor] = None):
    """x bf16 [M, C]; scale/shift fp32 [B, C] (contiguous rows) -> y (and y2 when scale2/shift2 are given)."""
    l = _lib.load()
    _bf16(x)
    m, c = x.shape
    y = torch.empty_like(x)
    y2 = torch.empty_like(x) if scale2 is not None else None
    for t in (scale, shift, scale2, shift2):
        assert t is None or (t.dtype == torch.float32 and t.is_contiguous() and t.shape[1] == c)
    _lib.check(l.mx_layernorm_mod(_lib.current_stream(), x.data_ptr(), y.data_ptr(), _p(y2), scale.data_ptr(), shift.data_ptr(),
                                  _p(scale2), _p(shift2), c, m, c, rows_per_batch, eps), "mx_layernorm_mod")
    return (y, y2) if y2 is not None else y


def rmsnorm_heads_(x: torch.Tensor, nbatch: int, rows_per_batch: int, batch_rows: int, row_off: int, heads_total: int,
                   heads_q: int, wq: torch.Tensor, wk: torch.Tensor, eps: float = 1e-6, q_scale: float = 1.0) -> torch.Tensor:
    l = _lib.load()
    _bf16(x)
    _lib.check(l.mx_rmsnorm_heads(_lib.current_stream(), x.data_ptr(), x.shape[1], nbatch, rows_per_batch, batch_rows, row_off,
                                  heads_total, heads_q, wq.data_ptr(), wk.data_ptr(), eps, q_scale), "mx_rmsnorm_heads")
    return x


def cfg_flow_step_(noise: torch.Tensor, latents: torch.Tensor, sigma: torch.Tensor, sigma_next: torch.Tensor,
                   guidance_scale: float) -> torch.Tensor:
    l = _lib.load()
    assert latents.is_contiguous() and noise.is_contiguous() and noise.dtype == latents.dtype
    sg = sigma.to(device=latents.device, dtype=torch.float32).contiguous()
    sn = sigma_next.to(device=latents.device, dtype=torch.float32).contiguous()
    _lib.check(l.mx_cfg_flow_step(_lib.current_stream(), noise.data_ptr(), latents.data_ptr(), sg.data_ptr(), sn.data_ptr(),
                                  float(guidance_scale), latents.shape[0], latents[0].numel(), _lib.torch_dtype_code(latents.dtype)),
               "mx_cfg_flow_step")
    return latents
