// Shared argument block and fused epilogue of the gfx950 GEMM / implicit-GEMM kernels (gemm_bf16.hip: generic
// 128-row tile; gemm_bf16_v2.hip: 256-row pipelined tile).  See gemm_bf16.hip for the orientation: the accumulator
// block acc[i][j] of v_mfma_f32_16x16x32_bf16 holds, per lane, features n = n_wave0 + 16 i + 4 (lane>>4) + {0..3}
// of token m = m_wave0 + 16 j + (lane & 15).
#pragma once
#include "common.h"
#include "../../include/mxdenoise.h"

namespace mx {

struct GemmArgs {
  const bf16_t* a;
  const bf16_t* w;
  void* c;
  const float* bias;
  const float* rowbias;
  const bf16_t* residual;
  bf16_t* vt;
  int M, N, K;
  int lda, ldc, ldr, ldrb;
  int rows_per_batch;
  int flags;
  int seg, period, ldvt;
  int B, Hin, Win, Cin, Hout, Wout, stride, up, corner_patch;
  int a_batch_rows, a_row_off, c_batch_rows, c_row_off;
  const float* gate;
  int ldg;
};

// row of the A operand / of the output for logical row m (joint-sequence remap, see mxdenoise.h)
__device__ __forceinline__ long gemm_in_row(const GemmArgs& p, int m) {
  if (p.a_batch_rows <= 0) return m;
  const int b = m / p.rows_per_batch;
  return (long)b * p.a_batch_rows + p.a_row_off + (m - b * p.rows_per_batch);
}
__device__ __forceinline__ long gemm_out_row(const GemmArgs& p, int m, int bidx) {
  if (p.c_batch_rows <= 0) return m;
  return (long)bidx * p.c_batch_rows + p.c_row_off + (m - bidx * p.rows_per_batch);
}

// Residual values of the plain epilogue path, fetched by the caller BEFORE its main loop (older than every LDS-DMA, so the
// kernel's counted vmcnt waits cover them and the HBM read hides under the K loop).
template <int NI, int MI>
__device__ __forceinline__ void gemm_prefetch_residual(const GemmArgs& p, u32x2 (&pre)[NI][MI], const int m_wave0,
                                                       const int wave_n0, const int fr, const int fq) {
#pragma unroll
  for (int j = 0; j < MI; ++j) {
    const int m = m_wave0 + j * 16 + fr;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int n = wave_n0 + i * 16 + fq * 4;
      long rr = m;
      if (m < p.M) {
        const int bidx = (p.rows_per_batch > 0) ? (m / p.rows_per_batch) : 0;
        rr = (p.flags & MX_EPI_RES_BCAST) ? (long)(m - bidx * p.rows_per_batch) : gemm_out_row(p, m, bidx);
      }
      pre[i][j] = (m < p.M && n < p.N) ? *reinterpret_cast<const u32x2*>(p.residual + rr * p.ldr + n) : u32x2{0u, 0u};
    }
  }
}

template <int NI, int MI, int BN, bool PRE = false>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& p, f32x4 (&acc)[NI][MI], const int m_wave0,
                                              const int wave_n0, const int fr, const int fq,
                                              const u32x2 (*pre)[MI] = nullptr) {
  const int flags = p.flags;
  const bool geglu = (flags & MX_EPI_GEGLU) != 0;
  const bool qkv = (flags & MX_EPI_QKV) != 0;
  // QKV: the wave's feature range lies inside one segment (seg % 64 == 0)
  int seg_idx = 0, seg_grp = 0, seg_pos = 0;
  bool to_vt = false;
  if (qkv) {
    seg_idx = wave_n0 / p.seg;
    seg_grp = seg_idx / p.period;
    seg_pos = seg_idx - seg_grp * p.period;
    to_vt = (seg_pos == p.period - 1);
  }

#pragma unroll
  for (int j = 0; j < MI; ++j) {
    const int m = m_wave0 + j * 16 + fr;
    if (m >= p.M) continue;
    const int bidx = (p.rows_per_batch > 0) ? (m / p.rows_per_batch) : 0;
    const long orow = gemm_out_row(p, m, bidx);
    const long rrow = (flags & MX_EPI_RES_BCAST) ? (long)(m - bidx * p.rows_per_batch) : orow;
#pragma unroll
    for (int i = 0; i < (NI); ++i) {
      if (geglu && i >= NI / 2) continue;
      const int n = wave_n0 + i * 16 + fq * 4;  // packed feature index of v[0]
      if (n >= p.N) continue;
      float v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = acc[i][j][q];
      if (p.bias) {
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(p.bias + n);
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] += b4[q];
      }
      if (geglu) {
        float g[4];
        const int ng = n + (NI / 2) * 16;  // gate blocks follow the hidden blocks inside the wave tile
#pragma unroll
        for (int q = 0; q < 4; ++q) g[q] = acc[i + NI / 2][j][q];
        if (p.bias) {
          const f32x4 b4 = *reinterpret_cast<const f32x4*>(p.bias + ng);
#pragma unroll
          for (int q = 0; q < 4; ++q) g[q] += b4[q];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = v[q] * gelu_fast(g[q]);
        const int nout = wave_n0 / 2 + i * 16 + fq * 4;
        u32x2 o = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
        *reinterpret_cast<u32x2*>(reinterpret_cast<bf16_t*>(p.c) + (long)m * p.ldc + nout) = o;
        continue;
      }
      if (p.rowbias) {
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(p.rowbias + (long)bidx * p.ldrb + n);
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] += b4[q];
      }
      if (p.gate) {
        const f32x4 g4 = *reinterpret_cast<const f32x4*>(p.gate + (long)bidx * p.ldg + n);
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] *= g4[q];
      }
      if (p.residual) {
        u32x2 r;
        if constexpr (PRE) r = pre[i][j];
        else r = *reinterpret_cast<const u32x2*>(p.residual + rrow * p.ldr + n);
        v[0] += bf16lo_to_f32(r[0]); v[1] += bf16hi_to_f32(r[0]);
        v[2] += bf16lo_to_f32(r[1]); v[3] += bf16hi_to_f32(r[1]);
      }
      if (flags & MX_EPI_SILU) {
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = silu_f(v[q]);
      }
      if (flags & MX_EPI_GELU_TANH) {
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = gelu_tanh_f(v[q]);
      }
      if (qkv) {
        const int nin = n - seg_idx * p.seg;  // position inside the segment
        if (to_vt) {
          const int key0 = (p.c_batch_rows > 0 ? p.c_row_off : 0) + m - bidx * p.rows_per_batch;
          const int key = MX_VT_POS(key0);      // attention's V^T key order (mxdenoise.h)
          const int nv = p.N / p.period;
          bf16_t* dst = p.vt + ((long)bidx * nv + (long)seg_grp * p.seg + nin) * p.ldvt + key;
#pragma unroll
          for (int q = 0; q < 4; ++q) dst[(long)q * p.ldvt] = f32_to_bf16(v[q]);
        } else {
          const int ccol = seg_grp * (p.period - 1) * p.seg + seg_pos * p.seg + nin;
          u32x2 o = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
          *reinterpret_cast<u32x2*>(reinterpret_cast<bf16_t*>(p.c) + orow * p.ldc + ccol) = o;
        }
        continue;
      }
      if (flags & MX_EPI_OUT_F32) {
        f32x4 o = {v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.c) + orow * p.ldc + n) = o;
      } else {
        u32x2 o = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
        *reinterpret_cast<u32x2*>(reinterpret_cast<bf16_t*>(p.c) + orow * p.ldc + n) = o;
      }
    }
  }
}

}  // namespace mx
