// The tile body of the 256-row ping-pong GEMM / implicit-GEMM conv (gemm_bf16_v5.hip holds the description and the stand-alone kernel): one
// 256 x {160, 128} output tile per call, every thread of a 512-thread workgroup calls.  Shared with the chained launch of attn_tail.hip, which walks
// several dependent GEMM stages in ONE launch on exactly this code -- same tiles, same order of summation, same epilogue: same bits.
#pragma once
#include "common.h"
#include "../../include/mxdenoise.h"
#include "gemm_args.h"

namespace mx {

// zero page the loaders read for padding taps / past-the-end DMAs: as long as the widest input channel count (2 * Cin bytes); one per
// translation unit that includes this header (device symbols are not linked across them)
static __device__ __attribute__((aligned(64))) unsigned int g_zero_page5[16384 / 4] = {0};

constexpr int BK5 = 64;
constexpr int NSTAGE5 = 3;

__device__ __forceinline__ int swz5(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }
__device__ __forceinline__ void glds16_5(const void* gsrc, void* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}
// raw barrier that neither the compiler's memory motion nor its instruction scheduler crosses
#define MX5_BAR()                                 \
  do {                                            \
    asm volatile("" ::: "memory");                \
    __builtin_amdgcn_s_barrier();                 \
    asm volatile("" ::: "memory");                \
    __builtin_amdgcn_sched_barrier(0);            \
  } while (0)

#if defined(MX_EXP) && MX_EXP == 8   // diagnostic build: wall-clock stamps (100 MHz s_memrealtime) per workgroup, read back by tools/exp/timeline_v4.py
static __device__ unsigned long long g_v5_stamps[1024 * 2 * 4];
#define MX5_STAMP(slot) do { if (lane == 0 && (wave == 0 || wave == 7) && blockIdx.x < 1024) \
    g_v5_stamps[(blockIdx.x * 2 + (wave == 7)) * 4 + (slot)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define MX5_STAMP(slot) do {} while (0)
#endif

// MI: 16-wide token blocks per wave; tile rows BM5 = 64 * MI (256, or 128 for small M); FEAT / GEGLU: the epilogue features compiled in
// (gemm_args.h EPI_F_*; the launcher picks the smallest instantiation that serves the launch)
// VEC: the per-sample vectors (row bias, gate) are compiled in -- 40 registers of the epilogue; without them the QKV form does not spill
// WT: the tile's output (and its row statistics) leave as WRITE-THROUGH (sc1) stores -- the chained launch (attn_tail.hip) hands them to other
// workgroups inside the launch; the arithmetic is the same instruction sequence, so a tile's values do not depend on WT.
// tm / tn: the tile (the stand-alone kernel derives them from blockIdx, a chained launch from its work ticket); smem: NSTAGE5 stages.
template <int BN, int MI, bool CONV, int FEAT, bool GEGLU, bool VEC, bool WT = false>
__device__ __forceinline__ void gemm_v5_tile(const GemmArgs& pk, int tm, const int tn, bf16_t* const smem) {
  constexpr int BM5 = 64 * MI;
  constexpr int NI = BN / 32;                 // 16-wide feature blocks per wave (BN / 2 features)
  constexpr int WCH = BN * 8;                 // 16-byte chunks of the W tile
  constexpr int XI = BM5 * 8 / 512;           // X load instructions per thread per tile (4, or 2 for the 128-row tile)
  constexpr int WI = (WCH + 511) / 512;       // W load instructions per thread per tile (3 for BN 160, 2 for 128)
  constexpr int LOADS = XI + WI;
  constexpr int STAGE_ELEMS = (BM5 + BN) * BK5;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1;                   // 0..3: token quarter of the tile; groups: wm 0-1 = A, wm 2-3 = B
  const int wn = wave & 1;
  const bool group_b = wave >= 4;
  MX5_STAMP(0);
  GemmArgs p = pk;
  gemm_select_seg(p, pk, tm);
  const int nk = p.K / BK5;
  const char* zero = reinterpret_cast<const char*>(g_zero_page5);
  const int cs = tid & 7;
  const int tiles_per_tap = CONV ? p.Cin / BK5 : 1;

  // ---- issue side (as gemm_v2): the K tile the NEXT DMA group belongs to and ready-made per-thread source pointers for it ----
  bool parked = false;
  int is_kt = 0;
  const char* xsrc[XI];
  long xjump[XI];
  const char* wsrc[WI];
  int cb[XI], cy[XI], cx[XI];
  unsigned xchb[XI];
  int tap_next = 0, in_tap = 0;

  auto conv_set_tap = [&](int tap) __attribute__((always_inline)) {
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
      const bool ok = (cb[i] >= 0) && (iy >= -p.vhalo) && (iy < Hv + p.vhalo) && (ix >= 0) && (ix < Wv);
      const long off = ((((long)cb[i] * (p.Hin + 2 * p.vhalo) + (iy >> p.up) + p.vhalo) * p.Win + (ix >> p.up)) * p.Cin) * 2;
      xsrc[i] = (ok ? reinterpret_cast<const char*>(p.a) + off : zero) + xchb[i];
    }
  };
  auto setup_tile = [&]() __attribute__((always_inline)) {
    const int m0 = tm * BM5;
    const int n0 = tn * BN;
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      const int row = (i * 512 + tid) >> 3;
      const int ch = swz5(row, cs);
      const int m = m0 + row;
      if constexpr (!CONV) {
        const int mc = m < p.M ? m : p.M - 1;
        xsrc[i] = reinterpret_cast<const char*>(p.a) + (gemm_in_row(p, mc) * p.lda + ch * 8) * 2;
        xjump[i] = p.a2 != nullptr ? (reinterpret_cast<const char*>(p.a2) + ((long)mc * p.lda2 + ch * 8) * 2) - (xsrc[i] + (long)p.k_split * 2) : 0;
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
      }
    }
    if constexpr (CONV) conv_set_tap(0);
#pragma unroll
    for (int i = 0; i < WI; ++i) {
      int q = i * 512 + tid;
      if (q >= WCH) q -= WCH;                 // BN=160: the last instruction re-fetches rows 0..31 (same bytes, same slot)
      const int row = q >> 3;
      wsrc[i] = reinterpret_cast<const char*>(p.w) + ((long)(n0 + row) * p.K + swz5(row, cs) * 8) * 2;
    }
    tap_next = 0; in_tap = 0;
  };
  auto park_on_zero_page = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < XI; ++i) xsrc[i] = zero + lane * 16;
#pragma unroll
    for (int i = 0; i < WI; ++i) wsrc[i] = zero + lane * 16;
  };
  auto issue_group = [&](int stage) __attribute__((always_inline)) {
    bf16_t* st = smem + stage * STAGE_ELEMS;
    bf16_t* sw = st + BM5 * BK5;
#pragma unroll
    for (int i = 0; i < XI; ++i) glds16_5(xsrc[i], st + (i * 512 + wave * 64) * 8);
#pragma unroll
    for (int i = 0; i < WI; ++i) {
      const int qb = (i * 512 + wave * 64 >= WCH) ? i * 512 + wave * 64 - WCH : i * 512 + wave * 64;
      glds16_5(wsrc[i], sw + qb * 8);
    }
  };
  auto advance_cursor = [&]() __attribute__((always_inline)) {
    if (parked) return;
    if (++is_kt == nk) { is_kt = 0; parked = true; park_on_zero_page(); return; }
#pragma unroll
    for (int i = 0; i < WI; ++i) wsrc[i] += BK5 * 2;
    if constexpr (!CONV) {
      const bool to_a2 = p.a2 != nullptr && is_kt * BK5 == p.k_split;
#pragma unroll
      for (int i = 0; i < XI; ++i) xsrc[i] += BK5 * 2 + (to_a2 ? xjump[i] : 0L);
    } else {
      if (++in_tap == tiles_per_tap) {
        in_tap = 0;
        conv_set_tap(++tap_next);
      } else {
#pragma unroll
        for (int i = 0; i < XI; ++i) xsrc[i] += BK5 * 2;
      }
    }
  };

  const int fr = lane & 15;
  const int fq = lane >> 4;
  // fragment addresses (bytes inside a stage): lane (fr, fq) reads row base + fr, chunk 4 ks + fq
  unsigned wrd[2], xrd[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const int wrow = wn * (BN / 2) + fr, xrow = wm * 16 * MI + fr;
    wrd[ks] = (unsigned)(((BM5 * BK5) + wrow * BK5 + swz5(wrow, ks * 4 + fq) * 8) * 2);   // (16 i more rows keep the swizzle: (row >> 1) & 7 of row + 16 i)
    xrd[ks] = (unsigned)((xrow * BK5 + swz5(xrow, ks * 4 + fq) * 8) * 2);
  }

  setup_tile();
  issue_group(0); advance_cursor();
  issue_group(1); advance_cursor();
  auto wait_all_but_newest = [&]() __attribute__((always_inline)) {       // all of this thread's DMA groups but the youngest have landed
    if constexpr (LOADS == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    else if constexpr (LOADS == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (LOADS == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    static_assert(LOADS >= 4 && LOADS <= 7, "counted wait");
  };
  wait_all_but_newest();                      // own part of K tile 0 landed

  f32x4 acc[NI][MI];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < MI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  float ln_rstd[MI];
#pragma unroll
  for (int j = 0; j < MI; ++j) ln_rstd[j] = 1.0f;
  if constexpr (!CONV) {
    if (p.ln_stats != nullptr) gemm_ln_init<NI, MI>(p, acc, tm * BM5 + wm * 16 * MI, tn * BN + wn * (BN / 2), fr, fq, ln_rstd);
  }
  MX5_BAR();                                  // every wave's part of K tile 0 has landed
  if (group_b) MX5_BAR();                     // group B runs one barrier behind group A
  MX5_STAMP(1);

  int stage = 0;
  for (int kt = 0; kt < nk; ++kt) {
    // ---- L: all fragments of this K tile, the DMA share of tile kt + 2, the cursor ----
    const char* sb = reinterpret_cast<const char*>(smem) + stage * (STAGE_ELEMS * 2);
    bf16x8 wf[2][NI], xf[2][MI];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int i = 0; i < NI; ++i) wf[ks][i] = *reinterpret_cast<const bf16x8*>(sb + wrd[ks] + i * (16 * BK5 * 2));
#pragma unroll
      for (int j = 0; j < MI; ++j) xf[ks][j] = *reinterpret_cast<const bf16x8*>(sb + xrd[ks] + j * (16 * BK5 * 2));
    }
    const int st2 = stage >= 1 ? stage - 1 : NSTAGE5 - 1;      // (kt + 2) % 3: the stage of K tile kt - 1
    issue_group(st2);
    advance_cursor();
    wait_all_but_newest();                                      // own part of K tile kt + 1
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the fragment reads have returned: the stage may be restaged one phase from now
    __builtin_amdgcn_sched_barrier(0);
    MX5_BAR();
    // ---- M: registers only ----
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < MI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks][i], xf[ks][j], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    MX5_BAR();
    stage = stage == NSTAGE5 - 1 ? 0 : stage + 1;
  }
  MX5_STAMP(2);
  if (!group_b) MX5_BAR();                    // re-align the two groups

  const int m0 = tm * BM5, n0 = tn * BN;
  static_assert(!GEGLU || (NI % 4 == 0 && !CONV), "the gated epilogue pairs whole 32-feature halves");
  gemm_epilogue_regs<NI, MI, GEGLU, VEC, true, true, FEAT, true, false, WT>(p, acc, m0 + wm * 16 * MI, n0 + wn * (BN / 2), fr, fq, ln_rstd);
  if constexpr (!GEGLU && FEAT == 0 && MI == 4) {      // GroupNorm partial sums of the accumulators (gemm_args.h): pure ALU + 10 stores behind the tile's own
    if (pk.gn_part != nullptr) gemm_gn_partials<NI, MI>(p, acc, m0 + wm * 16 * MI, n0 + wn * (BN / 2), fr, fq);
  }
  MX5_STAMP(3);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the past-the-end DMAs are drained before the workgroup retires
  if constexpr (!CONV && !GEGLU && BM5 == 256) {      // finalised row statistics: the last workgroup of the 256-row panel folds its slabs (gemm_args.h)
    if (pk.ln_final_out != nullptr) gemm_ln_finalize(p, tm, pk.N / BN, reinterpret_cast<volatile int*>(smem));
  }
}


}  // namespace mx
