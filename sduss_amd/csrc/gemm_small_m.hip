// C = X W^T + epilogue for M <= 16 rows (round 5): the time / added-condition embedding MLPs and the stacked time_emb_proj of the UNet
// (unet.py:314-341, resnet.py:421: B = 8 rows at the headline batch), the timestep / pooled-text MLPs and the stacked AdaLN modulation of the MMDiT
// (SD3Transformer.py: 1.4 GB of weights for 8 rows).  These launches are weight STREAMS: every weight is read once and meets M <= 16 activations.
// The generic 128-row tile kernel walked them as ordinary tiles -- N / 128 workgroups, each a serial K loop of LDS-staged tiles (33 us for
// M8 N1280 K2816, 10 workgroups on 256 CUs).  Here one workgroup owns 16 output features: its four waves split K, every lane reads its 16 bytes of W
// and of X straight from memory into MFMA fragments (v_mfma_f32_16x16x32_bf16, A = 16 weight rows, B = the <= 16 activation rows), the four partial
// tiles meet in LDS in wave order (deterministic), and wave 0 applies bias -> residual -> SiLU and stores.  HBM-bound: N K 2 bytes per launch.
#include <algorithm>
#include <cstdlib>

#include "common.h"
#include "../../include/mxdenoise.h"
#include "gemm_args.h"

namespace mx {

constexpr int kSmallMRows = 16;

// WAVES waves split K; a workgroup walks `groups` consecutive 16-feature groups (long N: fewer, longer-lived workgroups keep the weight stream going)
template <bool F32OUT, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void gemm_small_m_kernel(const GemmArgs p, const int groups) {
  __shared__ __attribute__((aligned(16))) float red[WAVES][64][4];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, kq = lane >> 4;
  const int steps = p.K >> 5;                          // 32-deep k-steps (K % 64 == 0: an even count)
  const int per = (steps + WAVES - 1) / WAVES;
  const int s0 = min(wave * per, steps), s1 = min(s0 + per, steps);
  const int xr = r < p.M ? r : p.M - 1;                // rows >= M repeat the last row; their results are never stored
  const bf16_t* xp = p.a + (long)xr * p.lda + kq * 8;
  const int ngroups = p.N >> 4;
  for (int gi = 0; gi < groups; ++gi) {
    const int grp = blockIdx.x * groups + gi;
    if (grp >= ngroups) break;                         // (uniform)
    const int n0 = grp * 16;
    const bf16_t* wp = p.w + (long)(n0 + r) * p.K + kq * 8;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    int s = s0;
    for (; s + 4 <= s1; s += 4) {                      // four k-steps in flight: 8 x 16 bytes per lane
      bf16x8 wf[4], xf[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        wf[u] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(wp + (long)(s + u) * 32));
        xf[u] = *reinterpret_cast<const bf16x8*>(xp + (long)(s + u) * 32);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[u], xf[u], acc, 0, 0, 0);
    }
    for (; s < s1; ++s) {
      const bf16x8 wf = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(wp + (long)s * 32));
      const bf16x8 xf = *reinterpret_cast<const bf16x8*>(xp + (long)s * 32);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xf, acc, 0, 0, 0);
    }
    if (gi > 0) __syncthreads();                       // wave 0 has read the previous group's partial tiles
    *reinterpret_cast<f32x4*>(&red[wave][lane][0]) = acc;
    __syncthreads();
    if (wave != 0) continue;
    f32x4 v = *reinterpret_cast<const f32x4*>(&red[0][lane][0]);
#pragma unroll
    for (int w = 1; w < WAVES; ++w) v += *reinterpret_cast<const f32x4*>(&red[w][lane][0]);     // wave order: the same bits every launch
    // lane (r, kq) holds row m = r, features n0 + 4 kq + {0..3}
    const int m = r, n = n0 + 4 * kq;
    if (m < p.M) {
      if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + n);
      if (p.residual) {
        const u32x2 rr = *reinterpret_cast<const u32x2*>(p.residual + (long)m * p.ldr + n);
        v[0] += bf16lo_to_f32(rr[0]); v[1] += bf16hi_to_f32(rr[0]); v[2] += bf16lo_to_f32(rr[1]); v[3] += bf16hi_to_f32(rr[1]);
      }
      if (p.flags & MX_EPI_SILU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = silu_f(v[e]);
      }
      if constexpr (F32OUT) {
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.c) + (long)m * p.ldc + n) = v;
      } else {
        *reinterpret_cast<u32x2*>(reinterpret_cast<bf16_t*>(p.c) + (long)m * p.ldc + n) = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
      }
    }
  }
}

// The LONG stream (N K > 32 M weights: the MMDiT's stacked AdaLN modulation, 1.4 GB): no K split, no barrier in the loop.  The <= 16 activation rows are staged once per
// workgroup in LDS (row stride K + 8 elements: the 16 rows' pieces fall on distinct banks); every WAVE then owns 16 output features at a time over the whole K, eight
// 16-byte weight loads in flight per lane, each row read front to back as whole lines, and walks on to its next group.  (The K-split form above reached 3.5-3.9 TB/s
// on this size -- two barriers per group drain its loads -- the tile kernel 4.45.)
template <bool F32OUT>
__global__ __launch_bounds__(256) void gemm_small_m_stream_kernel(const GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char xs_raw[];      // X [M][K + 8] bf16 (lanes of rows >= M read row M - 1: their results are never stored)
  bf16_t* xs = reinterpret_cast<bf16_t*>(xs_raw);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int r = lane & 15, kq = lane >> 4;
  const int K = p.K, ldx = K + 8;
  for (int c = tid; c < p.M * (K >> 3); c += 256) {
    const int row = c / (K >> 3), piece = c - row * (K >> 3);
    *reinterpret_cast<u32x4*>(xs + (long)row * ldx + piece * 8) = *reinterpret_cast<const u32x4*>(p.a + (long)row * p.lda + piece * 8);
  }
  __syncthreads();
  const bf16_t* xl = xs + (long)(r < p.M ? r : p.M - 1) * ldx + kq * 8;
  const int steps = K >> 5;
  const int ngroups = p.N >> 4;
  for (int grp = blockIdx.x * 4 + wave; grp < ngroups; grp += gridDim.x * 4) {
    const int n0 = grp * 16;
    const bf16_t* wp = p.w + (long)(n0 + r) * K + kq * 8;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    int s = 0;
    for (; s + 16 <= steps; s += 16) {                  // sixteen 16-byte loads in flight per lane (1 KB of every one of the wave's 16 rows)
      bf16x8 wf[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) wf[u] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(wp + (long)(s + u) * 32));
#pragma unroll
      for (int u = 0; u < 16; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[u], *reinterpret_cast<const bf16x8*>(xl + (s + u) * 32), acc, 0, 0, 0);
    }
    for (; s + 8 <= steps; s += 8) {
      bf16x8 wf[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) wf[u] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(wp + (long)(s + u) * 32));
#pragma unroll
      for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[u], *reinterpret_cast<const bf16x8*>(xl + (s + u) * 32), acc, 0, 0, 0);
    }
    for (; s < steps; ++s)
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(wp + (long)s * 32)), *reinterpret_cast<const bf16x8*>(xl + s * 32), acc, 0, 0, 0);
    // lane (r, kq) holds row m = r, features n0 + 4 kq + {0..3}
    const int m = r, n = n0 + 4 * kq;
    if (m < p.M) {
      f32x4 v = acc;
      if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + n);
      if (p.residual) {
        const u32x2 rr = *reinterpret_cast<const u32x2*>(p.residual + (long)m * p.ldr + n);
        v[0] += bf16lo_to_f32(rr[0]); v[1] += bf16hi_to_f32(rr[0]); v[2] += bf16lo_to_f32(rr[1]); v[3] += bf16hi_to_f32(rr[1]);
      }
      if (p.flags & MX_EPI_SILU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = silu_f(v[e]);
      }
      if constexpr (F32OUT) *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.c) + (long)m * p.ldc + n) = v;
      else *reinterpret_cast<u32x2*>(reinterpret_cast<bf16_t*>(p.c) + (long)m * p.ldc + n) = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
    }
  }
}

constexpr long kSmallMLong = 32L << 20;        // weights (elements) from which the launch is a LONG stream
static bool small_m_long_fits(long M, long K) { return M * (K + 8) * 2 <= 64 * 1024; }      // the staged activations within 64 KB of LDS

// does this form serve d?  (plain linear layers only: bias, per-row residual, SiLU, bf16 or fp32 out)
bool small_m_serves(const mx_gemm_desc* d, bool conv) {
  static const bool off = [] { const char* e = getenv("MX_SMALL_M"); return e && e[0] == '0'; }();      // A/B: the generic tile kernel
  if (off || conv || d->n_segs != 0 || d->M <= 0 || d->M > kSmallMRows) return false;
  if (d->N % 16 != 0 || d->K % 64 != 0) return false;
  // a long stream (> 64 MB of weights) runs the no-split form, which stages the M x K activations in LDS (64 KB); beyond that the tile kernel
  if ((long)d->N * d->K > kSmallMLong && !small_m_long_fits(d->M, d->K)) return false;
  if (d->flags & ~(MX_EPI_SILU | MX_EPI_OUT_F32)) return false;
  if (d->rowbias || d->gate || d->vt || d->a2 || d->ln_stats || d->ln_final || d->stats_out || d->ln_final_out || d->gn_part_out) return false;
  if (d->out_scale != 0.f || d->a_batch_rows > 0 || d->c_batch_rows > 0 || d->splitk > 1) return false;
  if (d->lda % 8 != 0 || d->ldc % 4 != 0 || (d->residual && d->ldr % 4 != 0)) return false;
  return true;
}

int launch_small_m(hipStream_t s, const GemmArgs& a) {
  const int ngroups = a.N / 16;
  const int ncu = cu_count();
  // few groups (N 1280: 80): sixteen waves split K, so that a wave's share is one or two rounds of loads; many (the stacked projections): four waves per
  // workgroup and as many consecutive groups per workgroup as keep ~8 workgroups per CU busy for the launch's life
  const bool f32 = (a.flags & MX_EPI_OUT_F32) != 0;
  if ((long)a.N * a.K > kSmallMLong) {          // the long stream: persistent waves, one 16-feature group at a time over the whole K
    const size_t lds = (size_t)a.M * (a.K + 8) * 2;
    const int per_cu = std::max(1, std::min(6, (int)((160 * 1024) / (lds + 256))));
    const dim3 grid((unsigned)std::min(cdiv(ngroups, 4), per_cu * ncu));
    if (f32) hipLaunchKernelGGL((gemm_small_m_stream_kernel<true>), grid, dim3(256), lds, s, a);
    else hipLaunchKernelGGL((gemm_small_m_stream_kernel<false>), grid, dim3(256), lds, s, a);
    return 0;
  }
  const bool wide = ngroups < 2 * ncu && a.K >= 1024;
  const int groups = wide ? 1 : std::max(1, std::min(16, ngroups / (8 * ncu)));
  const dim3 grid((unsigned)cdiv(ngroups, groups));
  if (wide) {
    if (f32) hipLaunchKernelGGL((gemm_small_m_kernel<true, 16>), grid, dim3(1024), 0, s, a, groups);
    else hipLaunchKernelGGL((gemm_small_m_kernel<false, 16>), grid, dim3(1024), 0, s, a, groups);
  } else {
    if (f32) hipLaunchKernelGGL((gemm_small_m_kernel<true, 4>), grid, dim3(256), 0, s, a, groups);
    else hipLaunchKernelGGL((gemm_small_m_kernel<false, 4>), grid, dim3(256), 0, s, a, groups);
  }
  return 0;
}

}  // namespace mx
