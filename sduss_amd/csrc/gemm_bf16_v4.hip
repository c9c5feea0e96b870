// bf16 MFMA GEMM, 256 x 256 output tile, EIGHT-PHASE PING-PONG schedule (round 2).  Same math, orientation, swizzle and
// epilogue as gemm_bf16_v3.hip; what changes is how the eight waves share a CU.
//
// Why.  Measured in round 2 (profiles/r02_a_dma_stream_and_store_microbench.txt): the L2 -> LDS operand stream alone delivers a
// 64-KB K tile in 0.73-0.79 us (83-90 GB/s per CU) and its 64 MFMAs per wave need ~1.0 us of matrix-pipe time, but the
// one-barrier-per-K-tile loop of gemm_v3 takes 1.48 us: all eight waves wait, pass the barrier together, read their 24
// fragments together (LDS saturated, matrix pipe idle), then compete for the matrix pipe together.  Here the two wave rows
// run the SAME program one barrier apart (cdna guide "The 256^2 8-phase template"), so that on every SIMD one wave is in a
// matrix segment (16 MFMAs on register operands) while its partner reads fragments from LDS and issues LDS-DMA:
//
//   per K tile, per wave:   L1 | M1 | L2 | M2 | L3 | M3 | L4 | M4        (| = s_barrier; rows wm = 1 lag by one barrier)
//     L1  read W sub-tile 0 (4 x ds_read_b128) and X sub-tile 0 (8)       M1  acc[W0, X0] += ...   (16 MFMA = one C quadrant x K 64)
//     L2  read W sub-tile 1 (4)                                           M2  acc[W1, X0]
//     L3  read X sub-tile 1 (8, into X0's registers)                      M3  acc[W1, X1]
//     L4  counted s_waitcnt vmcnt(4)                                      M4  acc[W0, X1]
//   each L segment also issues one 16-KB half-tile of the operand stream (2 LDS-DMA instructions per thread).
//
//   * tile 256 tokens x 256 features x BK 64; 512 threads = 8 waves as 2 (tokens) x 4 (features); a wave owns 128 x 64 outputs
//     (32 accumulator blocks of v_mfma_f32_16x16x32_bf16, 128 VGPRs), walked as four 64 x 32 quadrants in snake order so every
//     L segment loads at most one new register sub-tile (X 32 VGPRs, W0 / W1 16 each);
//   * LDS: two K-tile buffers x four half-tiles of 16 KB = 128 KB.  Half-tile XH[q] holds, for BOTH wave rows, token sub-range q
//     of the wave's 128 tokens; WH[h] holds feature sub-range h of all four wave columns' 64 features -- so segment L1 needs
//     only XH[0] + WH[0], L2 WH[1], L3 XH[1].  This is a loader-side row permutation (the per-lane DMA source address); waves keep
//     contiguous 128-token x 64-feature output blocks, so the epilogue and the GEGLU / QKV / RMSNorm pairings are unchanged;
//   * operand stream: half-tiles of stream position t + 2 are issued during K tiles t and t + 1 (XH0 in L3, WH0 in L4, WH1 in the
//     next L1, XH1 in the next L2): each lands in a buffer region whose last fragment read completed at least two barriers
//     earlier for BOTH wave rows, and one counted wait per K tile (vmcnt(4) in L4: everything but the two youngest half-tiles)
//     followed by two barriers orders the landing before the first read (cdna guide "Read a staged buffer one phase AFTER
//     the wait that retires it" with the extra barrier for staggered wave groups);
//   * persistent: one workgroup per CU walks tiles t, t + grid, ...; the stream runs on into the next tile; the epilogue is the
//     register-exchange one (gemm_args.h: no LDS, no barrier).  The two wave rows re-align for the epilogue (row 0 takes one
//     extra barrier after the K loop, row 1 one before it).
#include <cstdlib>

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
constexpr int HT_ELEMS = 128 * BK4;            // one half-tile: 128 rows x 64 k = 16 KB

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

#if MX_EXP == 1 || MX_EXP == 12 || MX_EXP == 13 || MX_EXP == 123   // ablation: no MFMA (operands kept alive)
__device__ __forceinline__ f32x4 mx_mfma_stub(bf16x8 a, bf16x8 b, f32x4 c) { asm volatile("" :: "v"(a), "v"(b)); return c; }
#define MX_MFMA(a, b, c) mx_mfma_stub(a, b, c)
#else
#define MX_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)
#endif

template <bool VEC>
__global__ __launch_bounds__(512, 2) void gemm_v4_kernel(const GemmArgs p) {
  constexpr int NI = 4;                        // 16-wide feature blocks per wave (64 features)
  constexpr int MI = 8;                        // 16-wide token blocks per wave (128 tokens)
  __shared__ __attribute__((aligned(16))) bf16_t smem[2 * 4 * HT_ELEMS];   // [buffer][XH0, XH1, WH0, WH1]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2;                    // 0..1: wave row = ping-pong group
  const int wn = wave & 3;                     // 0..3
  const int mt = (p.M + BM4 - 1) / BM4;
  const int nt = p.N / BN4;
  const int total_tiles = mt * nt;
  const int nk = p.K / BK4;
  const char* abase = reinterpret_cast<const char*>(p.a);
  const char* wbase = reinterpret_cast<const char*>(p.w);
  const int cs = tid & 7;

  // ---- issue side: four cursors, one per half-tile kind, in stream order XH0, WH0, WH1, XH1.  Cursor c points at the next
  //      (tile, K tile) of its kind and holds ready-made per-thread byte offsets (the chooser guarantees they fit 32 bits). ----
  int c_tile[4], c_kt[4];
  unsigned c_off[4][2];
  unsigned c_buf = 0;                          // bit c: LDS buffer the cursor's next issue goes to
  auto setup = [&](const int c, const int t) __attribute__((always_inline)) {
    int tm, tn;
    gemm_tile_of_block(t, mt, nt, p.xcd_map, tm, tn);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = (i * 512 + tid) >> 3;    // row of the half-tile this thread's chunk belongs to; slot cs holds chunk swz4(row, cs)
      if (c == 0 || c == 3) {                  // XH[q]: rows 64 w + r  <->  token 128 w + 64 q + r of the tile
        const int q = c == 3;
        const int m = tm * BM4 + 128 * (row >> 6) + 64 * q + (row & 63);
        const int mc = m < p.M ? m : p.M - 1;  // clamped rows are computed and discarded by the epilogue mask
        c_off[c][i] = (unsigned)((gemm_in_row(p, mc) * p.lda + swz4(row, cs) * 8) * 2);
      } else {                                 // WH[h]: rows 32 w + r  <->  feature 64 w + 32 h + r of the tile
        const int h = c == 2;
        const int n = tn * BN4 + 64 * (row >> 5) + 32 * h + (row & 31);
        c_off[c][i] = (unsigned)(((long)n * p.K + swz4(row, cs) * 8) * 2);
      }
    }
  };
  auto issue = [&](const int c) __attribute__((always_inline)) {     // branch-free
#if MX_EXP == 2 || MX_EXP == 12 || MX_EXP == 23 || MX_EXP == 123
    if (c_kt[c] >= 0) { c_buf ^= 1u << c; return; }   // ablation: no LDS-DMA (results garbage)
#endif
    const int slot = c == 0 ? 0 : c == 3 ? 1 : c == 1 ? 2 : 3;
    bf16_t* st = smem + (((c_buf >> c) & 1) * 4 + slot) * HT_ELEMS;
    const char* base = (c == 0 || c == 3) ? abase : wbase;
#pragma unroll
    for (int i = 0; i < 2; ++i) glds16_4(base + c_off[c][i], st + (i * 512 + wave * 64) * 8);
    c_buf ^= 1u << c;
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
  const char* sbase = reinterpret_cast<const char*>(smem);
  auto read_x = [&](bf16x8 (&xf)[4][2], const int buf, const int q) __attribute__((always_inline)) {
#if MX_EXP == 3 || MX_EXP == 13 || MX_EXP == 23 || MX_EXP == 123
    asm volatile("" : "+v"(xf[0][0]), "+v"(xf[1][0]), "+v"(xf[2][0]), "+v"(xf[3][0])); return;   // ablation: no fragment reads
#endif
    const char* s = sbase + (buf * 4 + q) * (HT_ELEMS * 2);
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) xf[jj][ks] = *reinterpret_cast<const bf16x8*>(s + xrd[ks] + jj * (16 * BK4 * 2));
  };
  auto read_w = [&](bf16x8 (&wf)[2][2], const int buf, const int h) __attribute__((always_inline)) {
#if MX_EXP == 3 || MX_EXP == 13 || MX_EXP == 23 || MX_EXP == 123
    asm volatile("" : "+v"(wf[0][0]), "+v"(wf[1][0])); return;
#endif
    const char* s = sbase + (buf * 4 + 2 + h) * (HT_ELEMS * 2);
#pragma unroll
    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) wf[ii][ks] = *reinterpret_cast<const bf16x8*>(s + wrd[ks] + ii * (16 * BK4 * 2));
  };

  // ---- prologue: stream positions 0 (all four kinds) and 1 (XH0, WH0) ----
#pragma unroll
  for (int c = 0; c < 4; ++c) { c_tile[c] = blockIdx.x; c_kt[c] = 0; setup(c, blockIdx.x); }
  issue(0); advance(0);
  issue(1); advance(1);
  issue(2); advance(2);
  issue(3); advance(3);
  issue(0); advance(0);
  issue(1); advance(1);
  asm volatile("s_waitcnt vmcnt(4)" ::: "memory");     // position 0 has landed (this thread's part)
  MX_BAR();

  int buf = 0;                                 // LDS buffer of the K tile being computed = stream position & 1
  for (int tile = blockIdx.x; tile < total_tiles; tile += (int)gridDim.x) {
    f32x4 acc[NI][MI];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < MI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (wm == 1) MX_BAR();                     // wave row 1 runs one barrier behind row 0

    for (int kt = 0; kt < nk; ++kt) {
      bf16x8 xf[4][2], w0[2][2], w1[2][2];
      // ---- L1 / M1 ----
      read_w(w0, buf, 0);
      read_x(xf, buf, 0);
      issue(2); advance(2);
      MX_BAR();
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
          for (int ii = 0; ii < 2; ++ii) acc[ii][jj] = MX_MFMA(w0[ii][ks], xf[jj][ks], acc[ii][jj]);
      __builtin_amdgcn_s_setprio(0);
      MX_BAR();
      // ---- L2 / M2 ----
      read_w(w1, buf, 1);
      issue(3); advance(3);
      MX_BAR();
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
          for (int ii = 0; ii < 2; ++ii) acc[2 + ii][jj] = MX_MFMA(w1[ii][ks], xf[jj][ks], acc[2 + ii][jj]);
      __builtin_amdgcn_s_setprio(0);
      MX_BAR();
      // ---- L3 / M3 ----
      read_x(xf, buf, 1);
      issue(0); advance(0);
      MX_BAR();
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
          for (int ii = 0; ii < 2; ++ii) acc[2 + ii][4 + jj] = MX_MFMA(w1[ii][ks], xf[jj][ks], acc[2 + ii][4 + jj]);
      __builtin_amdgcn_s_setprio(0);
      MX_BAR();
      // ---- L4 / M4 ----
      issue(1); advance(1);
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // all but the two youngest half-tiles: the next stream position has landed
      MX_BAR();
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
          for (int ii = 0; ii < 2; ++ii) acc[ii][4 + jj] = MX_MFMA(w0[ii][ks], xf[jj][ks], acc[ii][4 + jj]);
      __builtin_amdgcn_s_setprio(0);
      MX_BAR();
      buf ^= 1;
    }
    if (wm == 0) MX_BAR();                     // re-align the two wave rows

    int tm, tn;
    gemm_tile_of_block(tile, mt, nt, p.xcd_map, tm, tn);
    const int m0 = tm * BM4, n0 = tn * BN4;
    if (p.flags & MX_EPI_GEGLU) gemm_epilogue_regs<NI, MI, true, VEC, false>(p, acc, m0 + wm * 16 * MI, n0 + wn * 16 * NI, fr, fq);
    else gemm_epilogue_regs<NI, MI, false, VEC, false>(p, acc, m0 + wm * 16 * MI, n0 + wn * 16 * NI, fr, fq);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the past-the-end DMAs before the workgroup retires
}

int launch_v4(hipStream_t s, const GemmArgs& a) {
  static const int ncu = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
      hipDeviceProp_t prop;
      if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) n = prop.multiProcessorCount;
    }
    return n & ~7;                              // whole XCD groups, so tile % 8 stays the workgroup's XCD (gemm_tile_of_block)
  }();
  const int tiles = cdiv(a.M, BM4) * (a.N / BN4);
  const dim3 grid(tiles > ncu && ncu > 0 ? ncu : tiles), block(512);
  if (a.rowbias || a.gate) hipLaunchKernelGGL(gemm_v4_kernel<true>, grid, block, 0, s, a);
  else hipLaunchKernelGGL(gemm_v4_kernel<false>, grid, block, 0, s, a);
  return 0;
}

}  // namespace mx
