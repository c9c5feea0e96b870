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

// does this form serve d?  (plain linear layers only: bias, per-row residual, SiLU, bf16 or fp32 out)
bool small_m_serves(const mx_gemm_desc* d, bool conv) {
  static const bool off = [] { const char* e = getenv("MX_SMALL_M"); return e && e[0] == '0'; }();      // A/B: the generic tile kernel
  if (off || conv || d->n_segs != 0 || d->M <= 0 || d->M > kSmallMRows) return false;
  if (d->N % 16 != 0 || d->K % 64 != 0) return false;
  // beyond ~64 MB of weights the tile kernel's LDS-staged stream is the faster one (the MMDiT's stacked AdaLN modulation, 1.36 GB: 305 vs 390 us; at 340 MB a tie:
  // profiles/r05_q_small_m.txt)
  if ((long)d->N * d->K > (32L << 20)) return false;
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
  const bool wide = ngroups < 2 * ncu && a.K >= 1024;
  const int groups = wide ? 1 : std::max(1, std::min(16, ngroups / (8 * ncu)));
  const dim3 grid((unsigned)cdiv(ngroups, groups));
  const bool f32 = (a.flags & MX_EPI_OUT_F32) != 0;
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
