// bf16 MFMA GEMM / implicit-GEMM conv3x3, large-tile pipelined variant for gfx950 (MI355X).
//
// Same math, orientation, swizzle and epilogues as gemm_bf16.hip (the generic fallback); what changes is the
// schedule, built for the SDXL step shapes (M = B*H*W in {8192, 32768, 131072}, N in multiples of 320):
//   * tile 256 tokens x BN features (BN = 160 or 128) x BK 64, 512 threads = 8 waves as 4(m) x 2(n): every SDXL
//     GEMM/conv at UNet batch 8 then decomposes into a multiple of 256 workgroups -- one full wave of the chip's
//     256 CUs, no tail round (with 128x128 tiles the N=1280 layers ran 640 tiles = 2.5 rounds);
//   * operands go HBM/L2 -> LDS directly (global_load_lds_dwordx4, no staging VGPRs / ds_write), XOR swizzle applied on
//     the per-lane SOURCE address (the LDS image of an LDS-DMA is lane-linear; cdna guide rule 21);
//   * 3-stage LDS ring, loads issued two K tiles ahead and left in flight across the barrier with a COUNTED
//     s_waitcnt vmcnt(N) + raw s_barrier (one barrier per K tile; cdna guide "Pipelining across barriers").
//     All LDS lives in ONE __shared__ array and the main loop contains no ordinary global load, so hipcc has no
//     reason to drain the DMA queue early.  A K tile is issued EVERY iteration (past the end the source is clamped
//     and the bytes land in a stage nobody reads any more), so one counted wait serves every iteration and the body
//     is a single basic block.
//   * schedule SCHED 1: barrier at the top of the iteration, fragment reads of k-step 1 and the DMA issue interleaved
//     with the MFMAs of k-step 0 (sched_group_barrier).
//     schedule SCHED 4: barrier BETWEEN the two MFMA blocks of an iteration, so the k-step-0 fragments of the NEXT
//     tile are read while the k-step-1 MFMAs of the current tile run: no fragment-read latency is exposed after a
//     barrier.
//   * conv3x3: per-row source pointers are recomputed only when the tap changes (every Cin/64 K tiles) and otherwise
//     just advance by one K tile; out-of-image taps and rows beyond M walk a zero page instead of branching.
#include <cstdlib>

#include "common.h"
#include "../../include/mxdenoise.h"
#include "gemm_args.h"

namespace mx {

// zero page the conv loader walks for padding taps: as long as the widest input channel count (bytes = 2*Cin)
constexpr int kZeroPageBytes = 16384;
__device__ __attribute__((aligned(64))) unsigned int g_zero_page[kZeroPageBytes / 4] = {0};

constexpr int BM2 = 256;
constexpr int BK2 = 64;
constexpr int NSTAGE = 3;

__device__ __forceinline__ int swz2(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

template <int BN, bool CONV, int SCHED>
__global__ __launch_bounds__(512, 2) void gemm_v2_kernel(const GemmArgs p) {
  constexpr int NI = BN / 32;                 // 16-wide feature blocks per wave (wave covers BN/2 features)
  constexpr int MI = 4;                       // 16-wide token blocks per wave (wave covers 64 tokens)
  constexpr int WCH = BN * 8;                 // 16-byte chunks of the W tile
  constexpr int XI = BM2 * 8 / 512;           // X load instructions per thread per tile (4)
  constexpr int WI = (WCH + 511) / 512;       // W load instructions per thread per tile (3 for BN=160, 2 for 128)
  constexpr int LOADS = XI + WI;              // per-thread DMA instructions per K tile
  constexpr int STAGE_ELEMS = (BM2 + BN) * BK2;
  __shared__ __attribute__((aligned(16))) bf16_t smem[NSTAGE * STAGE_ELEMS];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1;
  const int wn = wave & 1;
  const int mt = (p.M + BM2 - 1) / BM2;
  const int m0 = (blockIdx.x % mt) * BM2;     // m fastest: workgroups with equal id mod 8 (one XCD) share X panels
  const int n0 = (blockIdx.x / mt) * BN;
  const int nk = p.K / BK2;
  const char* zero = reinterpret_cast<const char*>(g_zero_page);

  // ---- per-thread source descriptors (thread -> LDS chunk slot q = i*512 + tid; row = q>>3, slot c' = q&7) ----
  const int cs = tid & 7;
  unsigned xoff[XI];        // GEMM: byte offset of (row, swizzled chunk) from p.a; rows beyond M clamp to M-1
  const char* xcur[XI];     // CONV: current source pointer of the row for the current tap (advances 128 B per K tile)
  int cb[XI], cy[XI], cx[XI];
  unsigned xchb[XI];        // CONV: byte offset of the thread's swizzled chunk inside a K tile
#pragma unroll
  for (int i = 0; i < XI; ++i) {
    const int row = (i * 512 + tid) >> 3;
    const int ch = swz2(row, cs);             // logical k-chunk this thread fetches for its slot
    const int m = m0 + row;
    if constexpr (!CONV) {
      const int mc = m < p.M ? m : p.M - 1;   // clamped rows are computed and discarded by the epilogue mask
      xoff[i] = (unsigned)((gemm_in_row(p, mc) * p.lda + ch * 8) * 2);
    } else {
      xchb[i] = ch * 16;
      if (m < p.M) {
        const int hw = p.Hout * p.Wout;
        const int b = m / hw;
        const int r = m - b * hw;
        const int oy = r / p.Wout;
        cb[i] = b; cy[i] = oy * p.stride; cx[i] = (r - oy * p.Wout) * p.stride;
      } else {
        cb[i] = -1; cy[i] = 0; cx[i] = 0;
      }
      xcur[i] = zero;
    }
  }
  unsigned woff[WI];
#pragma unroll
  for (int i = 0; i < WI; ++i) {
    int q = i * 512 + tid;
    if (q >= WCH) q -= WCH;                   // BN=160: the last instruction re-fetches rows 0..31 (same bytes, same slot)
    const int row = q >> 3;
    woff[i] = (unsigned)(((long)(n0 + row) * p.K + swz2(row, cs) * 8) * 2);
  }

  // CONV: (re)compute the row pointers for tap `tap` at channel offset 0
  auto conv_set_tap = [&](int tap) {
    const int dy = tap / 3 - 1;
    const int dx = tap - (tap / 3) * 3 - 1;
    const int Hv = p.Hin << p.up, Wv = p.Win << p.up;
    const int P = p.corner_patch;
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      int iy = cy[i] + dy;
      const int ix = cx[i] + dx;
      if (P > 0 && dy != 0 && dx != 0) {
        // halo-corner rule of the reference's sliced path (norm_silu_concat.cu:210-221, 228-239)
        const bool cross_r = ((iy + P) / P) != ((cy[i] + P) / P);
        const bool cross_c = ((ix + P) / P) != ((cx[i] + P) / P);
        if (cross_r && cross_c) iy = cy[i];
      }
      const bool ok = (cb[i] >= 0) && (iy >= 0) && (iy < Hv) && (ix >= 0) && (ix < Wv);
      const long off = ((((long)cb[i] * p.Hin + (iy >> p.up)) * p.Win + (ix >> p.up)) * p.Cin) * 2;
      xcur[i] = (ok ? reinterpret_cast<const char*>(p.a) + off : zero) + xchb[i];
    }
  };

  const int tiles_per_tap = CONV ? p.Cin / BK2 : 1;
  int tap_next = 0, in_tap = 0;   // CONV: issue cursor (tiles are issued strictly in order 0,1,2,...)

  // issue K tile `kt` (clamped to the last one) into ring stage `stage`
  auto issue_tile = [&](int kt, int stage) {
    bf16_t* st = smem + stage * STAGE_ELEMS;
    const int ktc = kt < nk ? kt : nk - 1;
    if constexpr (!CONV) {
      const char* xb = reinterpret_cast<const char*>(p.a) + (long)ktc * (BK2 * 2);
#pragma unroll
      for (int i = 0; i < XI; ++i) glds16(xb + xoff[i], st + (i * 512 + wave * 64) * 8);
    } else {
      if (kt < nk) {                           // wave-uniform; past the end the pointers simply stay where they are
        if (in_tap == 0) conv_set_tap(tap_next);
#pragma unroll
        for (int i = 0; i < XI; ++i) {
          glds16(xcur[i], st + (i * 512 + wave * 64) * 8);
          xcur[i] += BK2 * 2;
        }
        if (++in_tap == tiles_per_tap) { in_tap = 0; ++tap_next; }
      } else {
#pragma unroll
        for (int i = 0; i < XI; ++i) glds16(zero + xchb[i], st + (i * 512 + wave * 64) * 8);
      }
    }
    bf16_t* sw = st + BM2 * BK2;
    const char* wb = reinterpret_cast<const char*>(p.w) + (long)ktc * (BK2 * 2);
#pragma unroll
    for (int i = 0; i < WI; ++i) {
      const int qb = (i * 512 + wave * 64 >= WCH) ? i * 512 + wave * 64 - WCH : i * 512 + wave * 64;  // wave-uniform slot base
      glds16(wb + woff[i], sw + qb * 8);
    }
  };

  f32x4 acc[NI][MI];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < MI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15;
  const int fq = lane >> 4;

  auto load_frags = [&](int stage, int ks, bf16x8 (&wf)[NI], bf16x8 (&xf)[MI]) {
    const bf16_t* sx = smem + stage * STAGE_ELEMS;
    const bf16_t* sw = sx + BM2 * BK2;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int row = wn * (BN / 2) + i * 16 + fr;
      wf[i] = *reinterpret_cast<const bf16x8*>(&sw[row * BK2 + swz2(row, ks * 4 + fq) * 8]);
    }
#pragma unroll
    for (int j = 0; j < MI; ++j) {
      const int row = wm * 64 + j * 16 + fr;
      xf[j] = *reinterpret_cast<const bf16x8*>(&sx[row * BK2 + swz2(row, ks * 4 + fq) * 8]);
    }
  };
  auto mfma_block = [&](const bf16x8 (&wf)[NI], const bf16x8 (&xf)[MI]) {
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < MI; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf[j], acc[i][j], 0, 0, 0);
  };
  auto wait_tile = [&]() {   // all but the youngest tile's DMA instructions of this thread have completed
    if constexpr (LOADS == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  };
  constexpr int NM = NI * MI;
  constexpr int NF = NI + MI;

  // residual of the plain epilogue: issue its loads first (oldest in the vmcnt queue), consume them after the K loop
  u32x2 rpre[NI][MI];
  const bool use_pre = p.residual != nullptr && !(p.flags & (MX_EPI_GEGLU | MX_EPI_QKV));
  if (use_pre) gemm_prefetch_residual<NI, MI>(p, rpre, m0 + wm * 64, n0 + wn * (BN / 2), fr, fq);

  issue_tile(0, 0);
  issue_tile(1, 1);

  if constexpr (SCHED == 1) {
    for (int kt = 0; kt < nk; ++kt) {
      wait_tile();
      __builtin_amdgcn_s_barrier();
      const int stage = kt % NSTAGE;
      bf16x8 wf0[NI], xf0[MI], wf1[NI], xf1[MI];
      load_frags(stage, 0, wf0, xf0);
      issue_tile(kt + 2, (kt + 2) % NSTAGE);   // that stage was last read in iteration kt-1, which every wave has left
      load_frags(stage, 1, wf1, xf1);
      mfma_block(wf0, xf0);
      mfma_block(wf1, xf1);
      __builtin_amdgcn_sched_group_barrier(0x100, NF, 0);                 // fragment reads of k-step 0
#pragma unroll
      for (int g = 0; g < LOADS; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                // MFMA
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                // one LDS-DMA (VMEM read)
      }
#pragma unroll
      for (int g = 0; g < (NF + 1) / 2; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                // fragment reads of k-step 1
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 2 * NM - LOADS - (NF + 1) / 2, 0);
    }
  } else {
    // SCHED 4.  Invariant at the top of iteration kt: the k-step-0 fragments of tile kt are in (wf0, xf0), tile kt+1
    // is in flight or landed in stage (kt+1)%3, stage (kt+2)%3 is free.
    bf16x8 wf0[NI], xf0[MI], wf1[NI], xf1[MI];
    wait_tile();
    __builtin_amdgcn_s_barrier();
    load_frags(0, 0, wf0, xf0);
    for (int kt = 0; kt < nk; ++kt) {
      const int stage = kt % NSTAGE;
      load_frags(stage, 1, wf1, xf1);
      mfma_block(wf0, xf0);                     // k-step 0 of tile kt, with the k-step-1 fragment reads in its shadow
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
      for (int g = 0; g < (NF - 1) / 2; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, NM - (NF - 1) / 2, 0);
      issue_tile(kt + 2, (kt + 2) % NSTAGE);    // stage (kt+2)%3 == (kt-1)%3: every wave passed a barrier after reading it
      wait_tile();                              // all but tile kt+2's DMAs of this thread done => tile kt+1 has landed
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's reads of the stage others will restage are retired
      __builtin_amdgcn_s_barrier();
      load_frags((kt + 1) % NSTAGE, 0, wf0, xf0);   // next tile's k-step-0 fragments (stale bytes past the end, never used)
      mfma_block(wf1, xf1);                     // k-step 1 of tile kt covers those reads
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
      for (int g = 0; g < (NF - 1) / 2; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, NM - (NF - 1) / 2, 0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the clamped tail DMAs before the kernel moves on

  if (use_pre) gemm_epilogue<NI, MI, BN, true>(p, acc, m0 + wm * 64, n0 + wn * (BN / 2), fr, fq, rpre);
  else gemm_epilogue<NI, MI, BN, false>(p, acc, m0 + wm * 64, n0 + wn * (BN / 2), fr, fq);
}

template <int SCHED>
static void launch_sched(hipStream_t s, const GemmArgs& a, bool conv, int bn) {
  const int mt = cdiv(a.M, BM2);
  dim3 grid(mt * (a.N / bn)), block(512);
  if (bn == 160) {
    if (conv) hipLaunchKernelGGL((gemm_v2_kernel<160, true, SCHED>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((gemm_v2_kernel<160, false, SCHED>), grid, block, 0, s, a);
  } else {
    if (conv) hipLaunchKernelGGL((gemm_v2_kernel<128, true, SCHED>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((gemm_v2_kernel<128, false, SCHED>), grid, block, 0, s, a);
  }
}

int launch_v2(hipStream_t s, const GemmArgs& a, bool conv, int bn) {
  static const int sched = [] { const char* e = getenv("MX_V2_SCHED"); return e ? atoi(e) : 1; }();
  if (sched == 4) launch_sched<4>(s, a, conv, bn);
  else launch_sched<1>(s, a, conv, bn);
  return 0;
}

}  // namespace mx
