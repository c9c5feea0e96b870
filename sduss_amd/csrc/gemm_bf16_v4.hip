// bf16 MFMA GEMM, 256 x 256 output tile, PING-PONG schedule of the two wave rows (round 2).  Same math, orientation,
// swizzle and epilogue as gemm_bf16_v3.hip; what changes is how the eight waves share a CU.
//
// Why.  Measured in round 2 (profiles/r02_a_dma_stream_and_store_microbench.txt): the L2 -> LDS operand stream alone delivers a
// 64-KB K tile in 0.73-0.79 us (83-90 GB/s per CU) and its 64 MFMAs per wave need ~1.0 us of matrix-pipe time, but the
// one-barrier-per-K-tile loop of gemm_v3 takes 1.48 us: all eight waves wait, pass the barrier together, read their 24
// fragments together (LDS saturated, matrix pipe idle), then compete for the matrix pipe together.  Here the two wave rows
// run the SAME program one barrier apart (after the cdna guide's "256^2 8-phase template"), so that on every SIMD one wave
// is in a matrix segment (MFMAs on register operands only) while its partner reads fragments and issues LDS-DMA:
//
//   per K tile, per wave:   LA | MA | LB | MB          (| = s_barrier; wave row 1 lags row 0 by one barrier)
//     LA  read W sub-tiles 0, 1 (8 x ds_read_b128) and X sub-tile 0 (8); issue 2 half-tiles of the operand stream
//     MA  32 MFMA: acc[W0, X0], acc[W1, X0]
//     LB  read X sub-tile 1 (8, into X0's registers); issue 2 half-tiles; one counted s_waitcnt vmcnt(4)
//     MB  32 MFMA: acc[W1, X1], acc[W0, X1]
//   A first form with 16-MFMA segments (four per K tile, eight barriers) was correct but only 3 % faster than gemm_v3: its
//   ablation builds (round 2, git history) showed ~150 cycles of barrier + bookkeeping and ~200 cycles of fragment
//   reads + LDS-DMA issue per 256-cycle matrix segment, i.e. the partner's L segment was the longer one.  32-MFMA segments halve
//   the barriers and give the L segments 512 cycles of cover; the loader's control flow is peeled out of the hot loop.
//
//   * tile 256 tokens x 256 features x BK 64; 512 threads = 8 waves as 2 (tokens) x 4 (features); a wave owns 128 x 64 outputs
//     (32 accumulator blocks of v_mfma_f32_16x16x32_bf16, 128 VGPRs); register sub-tiles: X 32 VGPRs, W0 / W1 16 each;
//   * half-tiles (16 KB = 128 rows x 64 k): XH[q] holds, for BOTH wave rows, token sub-range q of the wave's 128 tokens; WH[h]
//     holds feature sub-range h of all four wave columns' 64 features -- so LA needs XH0, WH0, WH1 and LB needs XH1.  This is a
//     loader-side row permutation (the per-lane DMA source address); waves keep contiguous 128-token x 64-feature output blocks,
//     so the epilogue and the GEGLU / QKV / RMSNorm pairings are unchanged;
//   * LDS = a ring of TEN half-tile slots (all 160 KB); the stream order is XH0 WH0 WH1 XH1 per K tile, half-tile s lives in
//     slot s mod 10.  LA of K tile t issues WH1, XH1 of tile t + 1 and LB issues XH0, WH0 of tile t + 2: each DMA lands in a slot
//     whose last fragment read completed at least two barriers earlier for BOTH wave rows, and the wait in LB (all but the two
//     youngest half-tiles) plus the two barriers before the next LA order landing before reading (cdna guide: "Read a staged
//     buffer one phase AFTER the wait that retires it", one barrier more for staggered wave groups);
//   * persistent: one workgroup per CU walks tiles t, t + grid, ...; the stream runs on into the next tile; the epilogue is the
//     register-exchange one (gemm_args.h: no LDS, no barrier).  The two wave rows re-align for the epilogue (row 0 takes one
//     extra barrier after the K loop, row 1 one before it).
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "../../include/mxdenoise.h"
#include "gemm_args.h"

#ifndef MX_EXP
#define MX_EXP 0
#endif

namespace mx {

constexpr int BM4 = 256;
constexpr int BN4 = 256;
constexpr int BK4 = 64;
constexpr int HT_BYTES = 128 * BK4 * 2;        // one half-tile: 128 rows x 64 k = 16 KB
constexpr int NSLOT4 = 10;

__device__ __forceinline__ int swz4(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }

__device__ __forceinline__ void glds16_4(const void* gsrc, void* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

// raw barrier that neither the compiler's memory motion nor its instruction scheduler crosses
#define MX_BAR()                                  \
  do {                                            \
    asm volatile("" ::: "memory");                \
    __builtin_amdgcn_s_barrier();                 \
    asm volatile("" ::: "memory");                \
    __builtin_amdgcn_sched_barrier(0);            \
  } while (0)

#define MX_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)

#if MX_EXP == 8   // diagnostic build: wall-clock stamps (100 MHz s_memrealtime) per workgroup and tile, read back by tools/exp/timeline_v4.py
__device__ unsigned long long g_v4_stamps[256 * 2 * 64];
#define MX_STAMP(slot) do { if (lane == 0 && (wave == 0 || wave == 7) && (slot) < 32) { \
    g_v4_stamps[(blockIdx.x * 2 + (wave == 7)) * 64 + (slot)] = __builtin_amdgcn_s_memrealtime(); \
    g_v4_stamps[(blockIdx.x * 2 + (wave == 7)) * 64 + 32 + (slot)] = __builtin_amdgcn_s_memtime(); } } while (0)   /* shader clock beside the wall clock */
#else
#define MX_STAMP(slot) do {} while (0)
#endif

// VEC: per-sample vectors (row bias, gate) compiled in; FEAT: gemm_args.h EPI_F_*; GEGLU: the gated epilogue INSTEAD of the plain one
// LN (round 4): folded LayerNorm from FINALISED row statistics (mxdenoise.h ln_final: 8 bytes per row, written by the producing launch's last
// workgroup per panel).  Round 3 measured the slab form here (16 slabs per row summed in the hand-over between two tiles): +10-20 us per
// launch, more than the normalisation pass it replaces; the finalised form needs 8 + 4 loads per lane and tile, requested BEFORE the
// previous tile's epilogue so that their latency hides behind it.
template <bool VEC, int FEAT, bool GEGLU, bool LN = false>
__global__ __launch_bounds__(512, 2) void gemm_v4_kernel(const GemmArgs pk) {
  constexpr int NI = 4;                        // 16-wide feature blocks per wave (64 features)
  constexpr int MI = 8;                        // 16-wide token blocks per wave (128 tokens)
  __shared__ __attribute__((aligned(16))) char smem[NSLOT4 * HT_BYTES];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2;                    // 0..1: wave row = ping-pong group
  const int wn = wave & 3;                     // 0..3
  const int mt = gemm_m_tiles(pk, BM4);         // grouped launch (gemm_args.h): the m-tiles of all problems
  const int nt = pk.N / BN4;
  const int total_tiles = mt * nt;
  const int nk = pk.K / BK4;
  const char* abase = reinterpret_cast<const char*>(pk.a);   // grouped: the lowest of the problems' bases (launch_v4 checks the 32-bit reach)
  const char* wbase = reinterpret_cast<const char*>(pk.w);
  const int cs = tid & 7;
  MX_STAMP(0);
  [[maybe_unused]] int stamp_i = 1;

  // (Round 5, measured and removed: walking the feature panels of the fused q | k | v projection last to first, so that the V panels' 2-byte V^T stores
  //  drain behind the q / k panels' K loops instead of at the launch's end -- 83.2 us against 81.9 us for the plain order, same box: nil.)
  auto tile_of = [&](const int t, int& tm, int& tn) __attribute__((always_inline)) { gemm_tile_of_block(t, mt, nt, pk.xcd_map, tm, tn); };
  // ---- issue side: four cursors, one per half-tile kind (0 XH0, 1 WH0, 2 WH1, 3 XH1 = stream order inside a K tile).  Cursor c
  //      points at the next (tile, K tile) of its kind, holds ready-made per-thread byte offsets (the chooser guarantees they fit
  //      32 bits) and the ring slot of its next half-tile (stream index mod 10: + 4 per issue). ----
  int c_tile[4], c_kt[4], c_slot[4];
  unsigned c_off[4][2];
  auto setup = [&](const int c, const int t) __attribute__((always_inline)) {
    int tm, tn;
    tile_of(t, tm, tn);
    // the X half-tiles belong to the tile's problem: its rows, its base (as a byte offset from abase), its joint-sequence remap
    int seg_m = pk.M, rpb = pk.rows_per_batch, abr = pk.a_batch_rows, aro = pk.a_row_off;
    unsigned abyte = 0;
    if ((c == 0 || c == 3) && pk.nseg > 0) {
      int sidx = 0;
#pragma unroll
      for (int i = 1; i < kMaxSegs; ++i) if (i < pk.nseg && tm >= pk.prob[i].tile0) sidx = i;
      const GemmSeg& g = pk.prob[sidx];
      tm -= g.tile0; seg_m = g.M; rpb = g.rows_per_batch; abr = g.a_batch_rows; aro = g.a_row_off;
      abyte = (unsigned)(reinterpret_cast<const char*>(g.a) - abase);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = (i * 512 + tid) >> 3;    // row of the half-tile this thread's chunk belongs to; slot cs holds chunk swz4(row, cs)
      if (c == 0 || c == 3) {                  // XH[q]: rows 64 w + r  <->  token 128 w + 64 q + r of the tile
        const int q = c == 3;
        const int m = tm * BM4 + 128 * (row >> 6) + 64 * q + (row & 63);
        const int mc = m < seg_m ? m : seg_m - 1;  // clamped rows are computed and discarded by the epilogue mask
        long in_row = mc;
        if (abr > 0) { const int b = mc / rpb; in_row = (long)b * abr + aro + (mc - b * rpb); }
        c_off[c][i] = abyte + (unsigned)((in_row * pk.lda + swz4(row, cs) * 8) * 2);
      } else {                                 // WH[h]: rows 32 w + r  <->  feature 64 w + 32 h + r of the tile
        const int h = c == 2;
        const int n = tn * BN4 + 64 * (row >> 5) + 32 * h + (row & 31);
        c_off[c][i] = (unsigned)(((long)n * pk.K + swz4(row, cs) * 8) * 2);
      }
    }
  };
  auto issue = [&](const int c) __attribute__((always_inline)) {     // branch-free
    char* st = smem + c_slot[c] * HT_BYTES;
    const char* base = (c == 0 || c == 3) ? abase : wbase;
#pragma unroll
    for (int i = 0; i < 2; ++i) glds16_4(base + c_off[c][i], st + (i * 512 + wave * 64) * 16);
    c_slot[c] = c_slot[c] >= NSLOT4 - 4 ? c_slot[c] - (NSLOT4 - 4) : c_slot[c] + 4;
  };
  // hot form: the cursor stays inside its tile (the caller guarantees it)
  auto advance_hot = [&](const int c) __attribute__((always_inline)) {
    ++c_kt[c];
#pragma unroll
    for (int i = 0; i < 2; ++i) c_off[c][i] += BK4 * 2;
  };
  auto advance = [&](const int c) __attribute__((always_inline)) {
    if (c_tile[c] >= total_tiles) return;      // parked
    if (++c_kt[c] == nk) {
      c_kt[c] = 0;
      c_tile[c] += (int)gridDim.x;
      if (c_tile[c] < total_tiles) { setup(c, c_tile[c]); return; }
#pragma unroll
      for (int i = 0; i < 2; ++i) c_off[c][i] = lane * 16;           // past the end: same instruction count, harmless bytes
      return;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) c_off[c][i] += BK4 * 2;
  };

  // ---- fragment addresses (bytes from the start of a half-tile): lane (fr, fq) reads row base + fr, chunk 4 ks + fq ----
  const int fr = lane & 15;
  const int fq = lane >> 4;
  unsigned xrd[2], wrd[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    xrd[ks] = (unsigned)(((64 * wm + fr) * BK4 + swz4(fr, ks * 4 + fq) * 8) * 2);
    wrd[ks] = (unsigned)(((32 * wn + fr) * BK4 + swz4(fr, ks * 4 + fq) * 8) * 2);
  }
  auto read_x = [&](bf16x8 (&xf)[4][2], const int slot) __attribute__((always_inline)) {
    const char* s = smem + slot * HT_BYTES;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) xf[jj][ks] = *reinterpret_cast<const bf16x8*>(s + xrd[ks] + jj * (16 * BK4 * 2));
  };
  auto read_w = [&](bf16x8 (&wf)[2][2], const int slot) __attribute__((always_inline)) {
    const char* s = smem + slot * HT_BYTES;
#pragma unroll
    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) wf[ii][ks] = *reinterpret_cast<const bf16x8*>(s + wrd[ks] + ii * (16 * BK4 * 2));
  };
  auto wrap = [](int s) __attribute__((always_inline)) { return s >= NSLOT4 ? s - NSLOT4 : s; };

  // ---- prologue: stream positions 0 (all four kinds) and 1 (XH0, WH0) ----
#pragma unroll
  for (int c = 0; c < 4; ++c) { c_tile[c] = blockIdx.x; c_kt[c] = 0; c_slot[c] = c; setup(c, blockIdx.x); }
  issue(0); advance(0);
  issue(1); advance(1);
  issue(2); advance(2);
  issue(3); advance(3);
  issue(0); advance(0);
  issue(1); advance(1);
  asm volatile("s_waitcnt vmcnt(4)" ::: "memory");     // position 0 has landed (this thread's part)
  MX_BAR();

  // LN: (mean, rstd) of the lane's eight tokens and the column sums of its 16 features for the NEXT tile, requested before the current tile's
  // epilogue (dead during the K loop, so they cost the loop nothing); the QKV instantiation has no registers for that and loads at the tile start
  constexpr bool LN_PF = LN && FEAT != EPI_F_QKV;
  [[maybe_unused]] f32x2 pst[MI];
  [[maybe_unused]] f32x4 pcs[NI];
  auto ln_prefetch = [&](const int t, const bool with_cs) __attribute__((always_inline)) {
    int tm_i, tn_i;
    tile_of(t, tm_i, tn_i);
    if (!with_cs) {
#pragma unroll
      for (int j = 0; j < MI; ++j) {
        const int m = tm_i * BM4 + wm * 16 * MI + j * 16 + fr;
        pst[j] = *reinterpret_cast<const f32x2*>(pk.ln_final + (long)(m < pk.M ? m : pk.M - 1) * 2);
      }
    } else {
#pragma unroll
      for (int i = 0; i < NI; ++i) pcs[i] = *reinterpret_cast<const f32x4*>(pk.ln_colsum + tn_i * BN4 + wn * 16 * NI + i * 16 + fq * 4);
    }
  };
  if constexpr (LN_PF) ln_prefetch(blockIdx.x, false);

  int rs = 0;                                  // ring slot of XH0 of the K tile being computed (stream index 4 t mod 10)
  for (int tile = blockIdx.x; tile < total_tiles; tile += (int)gridDim.x) {
    f32x4 acc[NI][MI];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < MI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float ln_rstd[MI] = {};                    // (LN == false: unused, the epilogue is built without the rstd multiply)
    if constexpr (LN_PF) {
      ln_prefetch(tile, true);                 // the column sums (64 bytes per lane, the same for every token panel: cache-hot) at the tile start
#pragma unroll
      for (int j = 0; j < MI; ++j) ln_rstd[j] = pst[j][1];
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < MI; ++j) acc[i][j] = pcs[i] * (-pst[j][0]);
    } else if constexpr (LN) {
      int tm_i, tn_i;
      tile_of(tile, tm_i, tn_i);
      gemm_ln_init_final<NI, MI, FEAT != EPI_F_QKV>(pk, acc, tm_i * BM4 + wm * 16 * MI, tn_i * BN4 + wn * 16 * NI, fr, fq, ln_rstd);
    }

    if (wm == 1) MX_BAR();                     // wave row 1 runs one barrier behind row 0
    MX_STAMP(stamp_i);

    // one K tile; HOT: no cursor leaves its tile during this iteration (plain pointer increments, no control flow)
    auto k_tile = [&](auto hot_tag) __attribute__((always_inline)) {
      constexpr bool HOT = decltype(hot_tag)::value;
      bf16x8 xf[4][2], w0[2][2], w1[2][2];
      // ---- LA ----
      read_w(w0, wrap(rs + 1));
      read_x(xf, rs);
      read_w(w1, wrap(rs + 2));
      issue(2); if constexpr (HOT) advance_hot(2); else advance(2);
      issue(3); if constexpr (HOT) advance_hot(3); else advance(3);
      MX_BAR();
      // ---- MA ----
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
          for (int ii = 0; ii < 2; ++ii) acc[ii][jj] = MX_MFMA(w0[ii][ks], xf[jj][ks], acc[ii][jj]);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
          for (int ii = 0; ii < 2; ++ii) acc[2 + ii][jj] = MX_MFMA(w1[ii][ks], xf[jj][ks], acc[2 + ii][jj]);
      __builtin_amdgcn_s_setprio(0);
      MX_BAR();
      // ---- LB ----
      read_x(xf, wrap(rs + 3));
      issue(0); if constexpr (HOT) advance_hot(0); else advance(0);
      issue(1); if constexpr (HOT) advance_hot(1); else advance(1);
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // all but the two youngest half-tiles: the next stream position has landed
      MX_BAR();
      // ---- MB ----
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
          for (int ii = 0; ii < 2; ++ii) acc[2 + ii][4 + jj] = MX_MFMA(w1[ii][ks], xf[jj][ks], acc[2 + ii][4 + jj]);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
          for (int ii = 0; ii < 2; ++ii) acc[ii][4 + jj] = MX_MFMA(w0[ii][ks], xf[jj][ks], acc[ii][4 + jj]);
      __builtin_amdgcn_s_setprio(0);
      MX_BAR();
      rs = rs >= NSLOT4 - 4 ? rs - (NSLOT4 - 4) : rs + 4;
    };
    // During K tile kt the cursors of WH1 / XH1 move from stream position kt + 1 to kt + 2 and those of XH0 / WH0 from kt + 2 to
    // kt + 3: all stay inside this tile while kt + 3 < nk.
    int kt = 0;
    for (; kt + 3 < nk; ++kt) k_tile(std::true_type{});
    for (; kt < nk; ++kt) k_tile(std::false_type{});

    MX_STAMP(stamp_i + 1);
    if (wm == 0) MX_BAR();                     // re-align the two wave rows

    int tm, tn;
    tile_of(tile, tm, tn);
    GemmArgs p = pk;
    gemm_select_seg(p, pk, tm);
    const int m0 = tm * BM4, n0 = tn * BN4;
    if constexpr (LN_PF) { if (tile + (int)gridDim.x < total_tiles) ln_prefetch(tile + (int)gridDim.x, false); }
    gemm_epilogue_regs<NI, MI, GEGLU, VEC, false, LN, FEAT, false, LN && FEAT == EPI_F_QKV>(p, acc, m0 + wm * 16 * MI, n0 + wn * 16 * NI, fr, fq, ln_rstd);
    MX_STAMP(stamp_i + 2);
    stamp_i += 3;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the past-the-end DMAs before the workgroup retires
}

#if MX_EXP == 8
extern "C" int mx_debug_v4_stamps(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_v4_stamps), sizeof(g_v4_stamps)); }
#endif

int launch_v4(hipStream_t s, const GemmArgs& a) {
  const int ncu = cu_count();
  const int tiles = (a.nseg > 0 ? a.mt_total : cdiv(a.M, BM4)) * (a.N / BN4);
  const dim3 grid(tiles > ncu && ncu > 0 ? ncu : tiles), block(512);
  // the smallest instantiation that serves the launch (each carries only its own epilogue code: gemm_args.h, EPI_F_*)
  const bool vec = a.rowbias || a.gate;
  const int feat = gemm_epi_features(a.flags);
#define MX_V4(VEC_, FEAT_, GEGLU_) hipLaunchKernelGGL((gemm_v4_kernel<VEC_, FEAT_, GEGLU_>), grid, block, 0, s, a)
  if (a.ln_final != nullptr) {                 // the folded LayerNorm's instantiations: GEGLU / QKV / plain, no per-sample vectors (the launcher checked)
    if ((a.flags & MX_EPI_GEGLU) && !(feat & EPI_F_ACT)) hipLaunchKernelGGL((gemm_v4_kernel<false, 0, true, true>), grid, block, 0, s, a);
    else if (!(a.flags & MX_EPI_GEGLU) && feat == EPI_F_QKV) hipLaunchKernelGGL((gemm_v4_kernel<false, EPI_F_QKV, false, true>), grid, block, 0, s, a);
    else if (!(a.flags & MX_EPI_GEGLU) && feat == 0) hipLaunchKernelGGL((gemm_v4_kernel<false, 0, false, true>), grid, block, 0, s, a);
    else return 1;
    return 0;
  }
  if (a.flags & MX_EPI_GEGLU) {
    if (feat & EPI_F_ACT) MX_V4(false, EPI_F_ACT, true); else MX_V4(false, 0, true);      // (the gated epilogue takes no per-sample vectors)
  } else if (!vec) {
    if (feat == 0) MX_V4(false, 0, false);
    else if (feat == EPI_F_QKV) MX_V4(false, EPI_F_QKV, false);
    else if (feat == EPI_F_TANH) MX_V4(false, EPI_F_TANH, false);
    else MX_V4(true, EPI_F_ALL, false);
  } else {
    if (feat == 0) MX_V4(true, 0, false); else MX_V4(true, EPI_F_ALL, false);
  }
#undef MX_V4
  return 0;
}

}  // namespace mx
