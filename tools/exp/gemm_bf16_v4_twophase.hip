// EXPERIMENT (round 3, not shipped): the two-segment form of sduss_amd/csrc/gemm_bf16_v4.hip.  Built by tools/exp/build_v4_twophase.sh; measured
// 4-8 % slower in wall time than the four-segment form on the same box (profiles/r03_h_gemm_bench_ab_twophase.txt) although it spends fewer
// shader cycles on barriers -- see the note in DESIGN.md section 4 on the power-limited clock.
//
// bf16 MFMA GEMM, 256 x 256 output tile, PING-PONG schedule of the two wave rows.  Same math, orientation, swizzle and epilogue as the
// lock-step 256 x 256 kernel of round 1 (tools/exp/gemm_bf16_v3.hip); what changes is how the eight waves share a CU.
//
// Why ping-pong (round 2, profiles/r02_a_dma_stream_and_store_microbench.txt): the L2 -> LDS operand stream alone delivers a 64-KB K tile in
// 0.73-0.79 us and its 64 MFMAs per wave need ~1.0 us of matrix-pipe time, but a one-barrier-per-K-tile loop takes 1.48 us: all eight waves
// wait, pass the barrier together, read their 24 fragments together (LDS saturated, matrix pipe idle), then compete for the matrix pipe
// together.  Here the two wave rows run the SAME program one barrier apart (after the cdna guide's "256^2 8-phase template"), so that on
// every SIMD one wave is in a matrix segment while its partner reads fragments and issues LDS-DMA.
//
// TWO segments per K tile (round 3; the round-2 form had four -- LA | MA | LB | MB, the shipped sduss_amd/csrc/gemm_bf16_v4.hip):
//
//   per K tile, per wave:   L | M          (| = s_barrier; wave row 1 lags row 0 by one barrier)
//     L  read W sub-tiles 0, 1 (8 x ds_read_b128) and X sub-tile 0 (8); issue 2 half-tiles of the operand stream
//     M  read X sub-tile 1 (8, into its OWN registers, issued first and hidden under the MFMAs that follow); issue 2 half-tiles;
//        32 MFMA acc[W0 W1, X0]; 32 MFMA acc[W1 W0, X1]
//   The four-segment form spent ~150 cycles per barrier on barrier + bookkeeping against 512-cycle matrix segments (tools/exp/stamps_v4.py);
//   1024-cycle segments halve that.  It became possible when the epilogue's registers were trimmed (round 3): the second X sub-tile
//   needs 32 more VGPRs in the loop.
//
//   * tile 256 tokens x 256 features x BK 64; 512 threads = 8 waves as 2 (tokens) x 4 (features); a wave owns 128 x 64 outputs
//     (32 accumulator blocks of v_mfma_f32_16x16x32_bf16, 128 VGPRs); register sub-tiles: X0, X1 32 VGPRs each, W0 / W1 16 each;
//   * half-tiles (16 KB = 128 rows x 64 k): XH[q] holds, for BOTH wave rows, token sub-range q of the wave's 128 tokens; WH[h]
//     holds feature sub-range h of all four wave columns' 64 features -- so L needs XH0, WH0, WH1 and M needs XH1.  This is a
//     loader-side row permutation (the per-lane DMA source address); waves keep contiguous 128-token x 64-feature output blocks,
//     so the epilogue and the GEGLU / QKV / RMSNorm pairings are unchanged;
//   * LDS = a ring of TEN half-tile slots (all 160 KB); the stream order is XH0 WH0 WH1 XH1 per K tile, half-tile s lives in
//     slot s mod 10 (2.5 K tiles resident).  With phases numbered globally (row 0: L(t) in phase 2t, M(t) in 2t + 1; row 1 one later)
//     XH0 / WH0 / WH1 of tile t are last read in phase 2t + 1 and XH1 in 2t + 2; a half-tile must be issued by ALL eight waves
//     at least a phase and a half before its first read.  Both hold when every wave issues, in global phase 2u - 1, XH0 / WH0 of tile
//     u + 1 and, in phase 2u, WH1 / XH1 of tile u + 1 -- i.e. the two rows issue DIFFERENT half-tiles in their L and M segments:
//         row 0:  L(t): WH1, XH1 of t + 1      M(t): XH0, WH0 of t + 2      wait (all but the last 4 loads) at the end of M
//         row 1:  L(t): XH0, WH0 of t + 2      M(t): WH1, XH1 of t + 2      wait at the end of L
//     Each wait is followed by a barrier and a full segment of the other row before the first read (cdna guide: "Read a staged buffer
//     one phase AFTER the wait that retires it", one barrier more for staggered wave groups).  The cursors are indexed by ROLE (two
//     issued in L, two in M); which half-tile kind a role streams depends on the wave row;
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
template <bool VEC, int FEAT, bool GEGLU>
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

  // ---- issue side: four cursors, indexed by ROLE: roles 0, 1 are issued in the L segment, roles 2, 3 in the M segment.  The half-tile
  //      KIND (0 XH0, 1 WH0, 2 WH1, 3 XH1 = stream order inside a K tile) a role streams depends on the wave row (see the header):
  //      row 0: roles {WH1, XH1, XH0, WH0}, row 1: roles {XH0, WH0, WH1, XH1}.  A cursor points at the next (tile, K tile) of its kind,
  //      holds ready-made per-thread byte offsets (the chooser guarantees they fit 32 bits) and the ring slot of its next half-tile
  //      (stream index mod 10: + 4 per issue). ----
  int c_tile[4], c_kt[4], c_slot[4];
  unsigned c_off[4][2];
  auto kind_of = [&](const int role) __attribute__((always_inline)) { return wm == 0 ? ((role + 2) & 3) : role; };   // (wave-uniform)
  auto setup = [&](const int c, const int t) __attribute__((always_inline)) {
    const int kind = kind_of(c);
    const bool is_x = kind == 0 || kind == 3;
    int tm, tn;
    gemm_tile_of_block(t, mt, nt, pk.xcd_map, tm, tn);
    // the X half-tiles belong to the tile's problem: its rows, its base (as a byte offset from abase), its joint-sequence remap
    int seg_m = pk.M, rpb = pk.rows_per_batch, abr = pk.a_batch_rows, aro = pk.a_row_off;
    unsigned abyte = 0;
    if (is_x && pk.nseg > 0) {
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
      if (is_x) {                              // XH[q]: rows 64 w + r  <->  token 128 w + 64 q + r of the tile
        const int q = kind == 3;
        const int m = tm * BM4 + 128 * (row >> 6) + 64 * q + (row & 63);
        const int mc = m < seg_m ? m : seg_m - 1;  // clamped rows are computed and discarded by the epilogue mask
        long in_row = mc;
        if (abr > 0) { const int b = mc / rpb; in_row = (long)b * abr + aro + (mc - b * rpb); }
        c_off[c][i] = abyte + (unsigned)((in_row * pk.lda + swz4(row, cs) * 8) * 2);
      } else {                                 // WH[h]: rows 32 w + r  <->  feature 64 w + 32 h + r of the tile
        const int h = kind == 2;
        const int n = tn * BN4 + 64 * (row >> 5) + 32 * h + (row & 31);
        c_off[c][i] = (unsigned)(((long)n * pk.K + swz4(row, cs) * 8) * 2);
      }
    }
  };
  const char* c_base[4];                       // operand base of every role (wave-uniform)
#pragma unroll
  for (int c = 0; c < 4; ++c) { const int kind = kind_of(c); c_base[c] = (kind == 0 || kind == 3) ? abase : wbase; }
  auto issue = [&](const int c) __attribute__((always_inline)) {     // branch-free
    char* st = smem + c_slot[c] * HT_BYTES;
#pragma unroll
    for (int i = 0; i < 2; ++i) glds16_4(c_base[c] + c_off[c][i], st + (i * 512 + wave * 64) * 16);
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

  // ---- prologue: stream position 0 (all four kinds), then what the row must have in flight before its first segment: row 0 position 1
  //      of XH0 / WH0 (its M-segment roles), row 1 position 1 of all four kinds ----
#pragma unroll
  for (int c = 0; c < 4; ++c) { c_tile[c] = blockIdx.x; c_kt[c] = 0; c_slot[c] = kind_of(c); setup(c, blockIdx.x); }
  issue(0); advance(0);
  issue(1); advance(1);
  issue(2); advance(2);
  issue(3); advance(3);
  issue(2); advance(2);
  issue(3); advance(3);
  if (wm == 1) {
    issue(0); advance(0);
    issue(1); advance(1);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // position 0 has landed (this thread's part)
  } else {
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  }
  MX_BAR();

  int rs = 0;                                  // ring slot of XH0 of the K tile being computed (stream index 4 t mod 10)
  for (int tile = blockIdx.x; tile < total_tiles; tile += (int)gridDim.x) {
    f32x4 acc[NI][MI];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < MI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float ln_rstd[MI] = {};              // (no folded LayerNorm in this kernel: pick_tile, gemm_bf16.hip)

    if (wm == 1) MX_BAR();                     // wave row 1 runs one barrier behind row 0
    MX_STAMP(stamp_i);

    // one K tile; HOT: no cursor leaves its tile during this iteration (plain pointer increments, no control flow)
    auto k_tile = [&](auto hot_tag) __attribute__((always_inline)) {
      constexpr bool HOT = decltype(hot_tag)::value;
      bf16x8 x0[4][2], x1[4][2], w0[2][2], w1[2][2];
      // ---- L: fragments of W (both sub-tiles) and X sub-tile 0; two half-tiles of the stream ----
      read_w(w0, wrap(rs + 1));
      read_x(x0, rs);
      read_w(w1, wrap(rs + 2));
      issue(0); if constexpr (HOT) advance_hot(0); else advance(0);
      issue(1); if constexpr (HOT) advance_hot(1); else advance(1);
      if (wm == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // row 1: everything but this segment's loads has landed
      // the fragments have returned before the barrier (the builtin, so that the compiler's counter model knows it and does not wait for
      // ALL LDS reads -- X sub-tile 1 included -- in front of the first MFMA of M)
      __builtin_amdgcn_s_waitcnt(0xC07F);                              // lgkmcnt(0)
      MX_BAR();
      // ---- M: X sub-tile 1 is requested first and arrives under the first 32 MFMAs; the segment's two half-tiles go out before them ----
      read_x(x1, wrap(rs + 3));
      issue(2); if constexpr (HOT) advance_hot(2); else advance(2);
      issue(3); if constexpr (HOT) advance_hot(3); else advance(3);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
          for (int ii = 0; ii < 2; ++ii) acc[ii][jj] = MX_MFMA(w0[ii][ks], x0[jj][ks], acc[ii][jj]);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
          for (int ii = 0; ii < 2; ++ii) acc[2 + ii][jj] = MX_MFMA(w1[ii][ks], x0[jj][ks], acc[2 + ii][jj]);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
          for (int ii = 0; ii < 2; ++ii) acc[2 + ii][4 + jj] = MX_MFMA(w1[ii][ks], x1[jj][ks], acc[2 + ii][4 + jj]);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
          for (int ii = 0; ii < 2; ++ii) acc[ii][4 + jj] = MX_MFMA(w0[ii][ks], x1[jj][ks], acc[ii][4 + jj]);
      __builtin_amdgcn_s_setprio(0);
      if (wm == 0) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // row 0: everything but this segment's loads has landed
      MX_BAR();
      rs = rs >= NSLOT4 - 4 ? rs - (NSLOT4 - 4) : rs + 4;
    };
    // During K tile kt the cursors move to stream position kt + 2 (row 0's L roles) or kt + 3 (all others): all stay inside this tile
    // while kt + 3 < nk.
    int kt = 0;
    for (; kt + 3 < nk; ++kt) k_tile(std::true_type{});
    for (; kt < nk; ++kt) k_tile(std::false_type{});

    MX_STAMP(stamp_i + 1);
    if (wm == 0) MX_BAR();                     // re-align the two wave rows

    int tm, tn;
    gemm_tile_of_block(tile, mt, nt, pk.xcd_map, tm, tn);
    GemmArgs p = pk;
    gemm_select_seg(p, pk, tm);
    const int m0 = tm * BM4, n0 = tn * BN4;
    gemm_epilogue_regs<NI, MI, GEGLU, VEC, false, false, FEAT>(p, acc, m0 + wm * 16 * MI, n0 + wn * 16 * NI, fr, fq, ln_rstd);
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
