// bf16 MFMA GEMM, 256 x 256 output tile, for the launches that fill the chip with such tiles (SD3.5 MMDiT projections and
// feed-forwards, SDXL GEGLU up-projections and QKV): same math, orientation and epilogue as gemm_bf16_v2.hip.
//
// Why a second large-tile kernel.  Removing one pipeline component at a time from the kernels (tools/exp_build.sh,
// profiles/r01_e_gemm_component_removal.txt) shows the K loop of the 256 x 160 kernel is held by the CU's L2 -> LDS fetch
// path, not by the matrix cores: its LDS-DMA stream alone takes 0.86 us per K tile (62 GB/s per CU, the per-CU L2 gather
// rate of the microarchitecture guide) against 0.56 us of MFMA work.  Bytes fetched per FLOP fall with the tile's harmonic
// size: (256+160)/(256*160) -> (256+256)/(256*256) is 23 % less DMA per MFMA, which brings the two within 15 % of each
// other (1.03 us of DMA, 0.89 us of MFMA per K tile), and the 128 x 64 wave tile reads 0.375 KB of LDS per MFMA instead
// of 0.45.
//
//   * tile 256 tokens x 256 features x BK 64.  The X tile and the W tile of a K tile (32 KB each, 128-byte rows) are
//     separate slots of a FIVE-slot ring = the CU's whole 160 KB: two slots are being read, three are in flight
//     (W of K tile g+1 and X of K tile g+2 are issued during K tile g), so the DMA queue never runs dry at the barrier.
//     Earlier forms: five 32-KB stages of BK 32 (64-byte LDS rows made every DMA row piece half a cache line: the stream
//     ran at half rate, 34 GB/s per CU) and two 64-KB stages of BK 64 (one K tile in flight: 46 GB/s per CU);
//   * 512 threads = 8 waves as 2 (tokens) x 4 (features); a wave owns 128 tokens x 64 features = 32 accumulator blocks of
//     v_mfma_f32_16x16x32_bf16 (128 VGPRs);
//   * same XOR swizzle and source-side application as v2 (16-byte chunk ^= (row >> 1) & 7 on 128-byte rows);
//   * one counted s_waitcnt vmcnt(4) + one raw s_barrier per K tile; the DMA issue is branch-free and interleaved with the MFMAs of
//     k-step 0 (sched_group_barrier), the fragment reads of k-step 1 with its second half;
//   * launches of more tiles than CUs whose epilogue needs no per-sample vectors are PERSISTENT: one workgroup per CU walks
//     its tiles (t, t + grid, ...) and their half-tiles form one DMA stream, so the first three half-tiles of the next tile
//     land during the epilogue, which transposes through the two ring slots the last K iteration just released
//     (round 1; gemm_args.h: gemm_epilogue_regs since round 2).  The variant with row-bias / gate support is at the 256-VGPR wall and keeps one
//     tile per workgroup.
#include <cstdlib>

#include "common.h"
#include "../../include/mxdenoise.h"
#include "gemm_args.h"

#ifndef MX_EXP
#define MX_EXP 0   // tools/exp_build.sh: 1 = no MFMA, 2 = no LDS-DMA inside the K loop, 4 = no epilogue (diagnostics only)
#endif

namespace mx {

#if MX_EXP == 7   // diagnostic build: wall-clock stamps (100 MHz s_memrealtime) per workgroup and tile, read back by tools/exp/stamps_v3.py
__device__ unsigned long long g_v3_stamps[256 * 2 * 64];
#define MX_STAMP(slot) do { if (lane == 0 && (wave == 0 || wave == 7) && (slot) < 64) \
    g_v3_stamps[(blockIdx.x * 2 + (wave == 7)) * 64 + (slot)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define MX_STAMP(slot) do {} while (0)
#endif

constexpr int BM3 = 256;
constexpr int BN3 = 256;
constexpr int BK3 = 64;

#if MX_EXP == 5 || MX_EXP == 6   // diagnostic: LDS-DMA source without the in-line chunk permutation (results wrong; timing only)
__device__ __forceinline__ int swz3(int row, int chunk) { return chunk; }
#else
__device__ __forceinline__ int swz3(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }
#endif

__device__ __forceinline__ void glds16_3(const void* gsrc, void* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

// VEC: the epilogue supports per-sample vectors (row bias / gate); the lean variant (!VEC) has the registers to run as a
// persistent tile stream.
template <bool VEC>
__global__ __launch_bounds__(512, 2) void gemm_v3_kernel(const GemmArgs p) {
  constexpr bool PERSIST = true;               // one workgroup per CU walks its tiles (round 2: also the per-sample-vector variant)
  constexpr int NI = 4;                        // 16-wide feature blocks per wave (64 features)
  constexpr int MI = 8;                        // 16-wide token blocks per wave (128 tokens)
  constexpr int XI = BM3 * 8 / 512;            // X DMA instructions per thread per K tile (4)
  constexpr int WI = BN3 * 8 / 512;            // W DMA instructions per thread per K tile (4)
  constexpr int LOADS = XI + WI;
  constexpr int SLOT_ELEMS = 256 * BK3;        // one X tile or one W tile: 32 KB
  constexpr int NSLOT = 5;
  __shared__ __attribute__((aligned(16))) bf16_t smem[NSLOT * SLOT_ELEMS];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2;                    // 0..1
  const int wn = wave & 3;                     // 0..3
  const int mt = (p.M + BM3 - 1) / BM3;
  const int nt = p.N / BN3;
  const int total_tiles = mt * nt;
  const int nk = p.K / BK3;

  // ---- issue side.  The X and W half-tiles of the workgroup's tiles (t = blockIdx.x, + gridDim.x, ...) form ONE stream of
  //      LDS-DMA groups: X(0) W(0) X(1) W(1) ...; half-tile h of the stream lives in slot h % 5.  Each cursor holds ready-made
  //      per-thread source offsets for its next group.  issue_x / issue_w are branch-free (they share a basic block with the
  //      MFMAs so each LDS-DMA can sit in an MFMA shadow); advance_x / advance_w hold the control flow and run after the MFMAs.
  //      Because the stream runs on into the next tile, that tile's X(0), W(0), X(1) are in flight during the epilogue. ----
  int x_tile = blockIdx.x, w_tile = blockIdx.x;   // tile of the next X / W group (>= total_tiles: parked)
  int x_kt = 0, w_kt = 0;                         // its K tile
  unsigned xoff[XI], woff[WI];                    // byte offsets from p.a / p.w (the chooser guarantees they fit 32 bits)
  const char* abase = reinterpret_cast<const char*>(p.a);
  const char* wbase = reinterpret_cast<const char*>(p.w);
  const int cs = tid & 7;
  auto setup_x = [&](int t) __attribute__((always_inline)) {
    int tm, tn;
    gemm_tile_of_block(t, mt, nt, p.xcd_map, tm, tn);
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      const int row = (i * 512 + tid) >> 3;    // LDS slot: row, slot cs holds logical chunk swz3(row, cs)
      const int m = tm * BM3 + row;
      const int mc = m < p.M ? m : p.M - 1;    // clamped rows are computed and discarded by the epilogue mask
      xoff[i] = (unsigned)((gemm_in_row(p, mc) * p.lda + swz3(row, cs) * 8) * 2);
    }
  };
  auto setup_w = [&](int t) __attribute__((always_inline)) {
    int tm, tn;
    gemm_tile_of_block(t, mt, nt, p.xcd_map, tm, tn);
#pragma unroll
    for (int i = 0; i < WI; ++i) {
      const int row = (i * 512 + tid) >> 3;
      woff[i] = (unsigned)(((long)(tn * BN3 + row) * p.K + swz3(row, cs) * 8) * 2);
    }
  };
  auto issue_x = [&](int slot) __attribute__((always_inline)) {
    bf16_t* st = smem + slot * SLOT_ELEMS;
#pragma unroll
    for (int i = 0; i < XI; ++i) glds16_3(abase + xoff[i], st + (i * 512 + wave * 64) * 8);
  };
  auto issue_w = [&](int slot) __attribute__((always_inline)) {
    bf16_t* st = smem + slot * SLOT_ELEMS;
#pragma unroll
    for (int i = 0; i < WI; ++i) glds16_3(wbase + woff[i], st + (i * 512 + wave * 64) * 8);
  };
  auto advance_x = [&]() __attribute__((always_inline)) {
    if (x_tile >= total_tiles) return;         // parked
    if (++x_kt == nk) {                        // on to the workgroup's next tile, or past the end of the stream
      x_kt = 0;
      x_tile = !PERSIST ? total_tiles : x_tile + (int)gridDim.x;
      if constexpr (PERSIST) {
        if (x_tile < total_tiles) { setup_x(x_tile); return; }
      }
#pragma unroll
      for (int i = 0; i < XI; ++i) xoff[i] = lane * 16;          // same instruction count, harmless bytes (the head of A)
      return;
    }
#pragma unroll
    for (int i = 0; i < XI; ++i) xoff[i] += BK3 * 2;
  };
  auto advance_w = [&]() __attribute__((always_inline)) {
    if (w_tile >= total_tiles) return;
    if (++w_kt == nk) {
      w_kt = 0;
      w_tile = !PERSIST ? total_tiles : w_tile + (int)gridDim.x;
      if constexpr (PERSIST) {
        if (w_tile < total_tiles) { setup_w(w_tile); return; }
      }
#pragma unroll
      for (int i = 0; i < WI; ++i) woff[i] = lane * 16;
      return;
    }
#pragma unroll
    for (int i = 0; i < WI; ++i) woff[i] += BK3 * 2;
  };

  // ---- fragment addresses: lane (fr, fq) reads row base + fr, chunk 4*ks + fq of a 16-row block ----
  const int fr = lane & 15;
  const int fq = lane >> 4;
  const int x_row = wm * 128 + fr;             // + 16 j
  const int w_row = wn * 64 + fr;              // + 16 i
  int koff[2];                                 // element offset of the lane's chunk for k-step 0 / 1; the same for every
#pragma unroll                                 // block of 16 rows: the swizzle depends on (row >> 1) & 7 = (fr >> 1) & 7
  for (int ks = 0; ks < 2; ++ks) koff[ks] = swz3(fr, ks * 4 + fq) * 8;

  if (p.stagger_ticks > 0) {                   // EXPERIMENT: XCD x (= blockIdx % 8) starts x/8 of a tile period late
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long want = (unsigned long long)p.stagger_ticks * (blockIdx.x & 7) / 8;
    while (__builtin_amdgcn_s_memrealtime() - t0 < want) __builtin_amdgcn_s_sleep(32);
  }
  setup_x(x_tile);
  setup_w(w_tile);
  issue_x(0); advance_x();                     // X(0) -> slot 0, W(0) -> slot 1, X(1) -> slot 2
  issue_w(1); advance_w();
  issue_x(2); advance_x();

  MX_STAMP(0);
  [[maybe_unused]] int stamp_i = 1;
  int xs = 0;                                  // slot of the X half-tile being read = (2 g) % 5 for stream position g; W follows
  for (int tile = blockIdx.x; tile < total_tiles; tile += (!PERSIST ? total_tiles : (int)gridDim.x)) {
    f32x4 acc[NI][MI];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < MI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float ln_rstd[MI] = {};              // (no folded LayerNorm in this kernel: pick_tile, gemm_bf16.hip)

    for (int kt = 0; kt < nk; ++kt) {
      // all but the youngest group (the X half-tile of the next stream position) has landed => X and W of this position are in
      // LDS (vmcnt retires in order: epilogue stores of the previous tile only make the wait more conservative)
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      __builtin_amdgcn_s_barrier();
#if MX_EXP == 7
      if (kt == 0) { MX_STAMP(stamp_i); }
#endif
      const int ws = xs == NSLOT - 1 ? 0 : xs + 1;
      const int f0 = ws == NSLOT - 1 ? 0 : ws + 1;           // slot of the next X half-tile, in flight
      const int f1 = f0 == NSLOT - 1 ? 0 : f0 + 1;           // the two slots read in the previous iteration (or used by the
      const int f2 = f1 == NSLOT - 1 ? 0 : f1 + 1;           // previous tile's epilogue), which every wave has left
      const bf16_t* sx = smem + xs * SLOT_ELEMS;
      const bf16_t* sw = smem + ws * SLOT_ELEMS;
      bf16x8 wf0[NI], xf0[MI], wf1[NI], xf1[MI];
#pragma unroll
      for (int i = 0; i < NI; ++i) wf0[i] = *reinterpret_cast<const bf16x8*>(sw + (w_row + 16 * i) * BK3 + koff[0]);
#pragma unroll
      for (int j = 0; j < MI; ++j) xf0[j] = *reinterpret_cast<const bf16x8*>(sx + (x_row + 16 * j) * BK3 + koff[0]);
#if MX_EXP != 2
      issue_w(f1);
      issue_x(f2);
#endif
#pragma unroll
      for (int i = 0; i < NI; ++i) wf1[i] = *reinterpret_cast<const bf16x8*>(sw + (w_row + 16 * i) * BK3 + koff[1]);
#pragma unroll
      for (int j = 0; j < MI; ++j) xf1[j] = *reinterpret_cast<const bf16x8*>(sx + (x_row + 16 * j) * BK3 + koff[1]);
#if MX_EXP != 1 && MX_EXP != 5
#pragma unroll
      for (int j = 0; j < MI; ++j)
#pragma unroll
        for (int i = 0; i < NI; ++i)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf0[i], xf0[j], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < MI; ++j)
#pragma unroll
        for (int i = 0; i < NI; ++i)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf1[i], xf1[j], acc[i][j], 0, 0, 0);
#endif
      // schedule: the 12 fragment reads of k-step 0; then its 32 MFMAs with the 8 LDS-DMAs (first half) and the 12 reads of
      // k-step 1 (second half) in their shadows; then the 32 MFMAs of k-step 1
      __builtin_amdgcn_sched_group_barrier(0x100, NI + MI, 0);
#pragma unroll
      for (int s = 0; s < LOADS; ++s) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      }
#pragma unroll
      for (int s = 0; s < NI + MI; ++s) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 2 * NI * MI - 2 * LOADS - (NI + MI), 0);
#if MX_EXP != 2
      advance_w();
      advance_x();
#endif
      xs = f0;
    }

    MX_STAMP(stamp_i + 1);
    int tm, tn;
    gemm_tile_of_block(tile, mt, nt, p.xcd_map, tm, tn);
    const int m0 = tm * BM3, n0 = tn * BN3;
#if MX_EXP == 4   // no epilogue: keep the accumulators alive with a store that never executes on real data
    {
      float t = 0.f;
      for (int i = 0; i < NI; ++i) for (int j = 0; j < MI; ++j) for (int q = 0; q < 4; ++q) t += acc[i][j][q];
      if (t == 12345.678f) reinterpret_cast<bf16_t*>(p.c)[m0 + n0] = f32_to_bf16(t);
    }
#else
    // Register-exchange epilogue (gemm_args.h): no LDS, no barrier.  The ring keeps receiving the first half-tiles of the
    // workgroup's next tile meanwhile; a wave that finishes early waits at the next K loop's first barrier, behind which
    // the slots its slower siblings are still reading get re-issued.
    if (p.flags & MX_EPI_GEGLU) gemm_epilogue_regs<NI, MI, true, VEC, false, false>(p, acc, m0 + wm * 16 * MI, n0 + wn * 16 * NI, fr, fq, ln_rstd);
    else gemm_epilogue_regs<NI, MI, false, VEC, false, false>(p, acc, m0 + wm * 16 * MI, n0 + wn * 16 * NI, fr, fq, ln_rstd);
    MX_STAMP(stamp_i + 2);
    stamp_i += 3;
    // the cursors' per-thread offsets are recomputed from (tile, K tile) rather than kept in registers across the epilogue
    if constexpr (PERSIST) {
    if (x_tile < total_tiles) {
      setup_x(x_tile);
#pragma unroll
      for (int i = 0; i < XI; ++i) xoff[i] += x_kt * (BK3 * 2);
    } else {
#pragma unroll
      for (int i = 0; i < XI; ++i) xoff[i] = lane * 16;
    }
    if (w_tile < total_tiles) {
      setup_w(w_tile);
#pragma unroll
      for (int i = 0; i < WI; ++i) woff[i] += w_kt * (BK3 * 2);
    } else {
#pragma unroll
      for (int i = 0; i < WI; ++i) woff[i] = lane * 16;
    }
    }
#endif
    if constexpr (!PERSIST) break;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the past-the-end DMAs before the workgroup retires
}

#if MX_EXP == 7
extern "C" int mx_debug_v3_stamps(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_v3_stamps), sizeof(g_v3_stamps));
}
#endif

int launch_v3(hipStream_t s, const GemmArgs& a) {
  const int ncu = cu_count();
  static const bool persist = [] { const char* e = getenv("MX_V3_PERSIST"); return !(e && e[0] == '0'); }();
  const int tiles = cdiv(a.M, BM3) * (a.N / BN3);
  dim3 block(512);
  // more tiles than CUs: one workgroup per CU walks its tiles as one DMA stream (the next tile's operands fly during the epilogue)
  const dim3 grid(persist && tiles > ncu && ncu > 0 ? ncu : tiles);
  static const int stagger = [] { const char* e = getenv("MX_V3_STAGGER_US"); return e ? (int)(atof(e) * 100.0) : 0; }();   // 100 MHz ticks
  GemmArgs a2 = a; a2.stagger_ticks = stagger;
  if (a.rowbias || a.gate) hipLaunchKernelGGL(gemm_v3_kernel<true>, grid, block, 0, s, a2);    // per-sample vectors compiled in
  else hipLaunchKernelGGL(gemm_v3_kernel<false>, grid, block, 0, s, a2);
  return 0;
}

}  // namespace mx
