// bf16 MFMA GEMM, 256 x 256 output tile, for the launches that fill the chip with such tiles (SD3.5 MMDiT projections and
// feed-forwards, SDXL GEGLU up-projections and QKV): same math, orientation and epilogues as gemm_bf16_v2.hip.
//
// Why a second large-tile kernel.  Removing one pipeline component at a time from the 256 x 160 kernel (tools/exp_build.sh,
// profiles/r01_d_gemm_component_removal.txt) shows its K loop is held by the CU's L2 -> LDS fetch path, not by the matrix
// cores: MFMA + fragment reads alone run a K tile in ~1.0 us, the LDS-DMA stream alone in ~0.86 us (62 GB/s per CU, the
// per-CU L2 gather rate of the microarchitecture guide), both together 1.3x the slower one.  Bytes fetched per FLOP fall with
// the tile's harmonic size: (256+160)/(256*160) -> (256+256)/(256*256) is 23 % less DMA per MFMA, and the wave tile
// 128 x 64 reads 0.375 KB of LDS per MFMA instead of 0.45.
//
//   * tile 256 tokens x 256 features, K in stages of 32 (64-byte LDS rows): 32 KB per stage, ring of NSTG = 5 stages
//     = the CU's whole 160 KB, four stages (128 KB) in flight;
//   * 512 threads = 8 waves as 2 (tokens) x 4 (features); a wave owns 128 tokens x 64 features = 32 accumulator blocks of
//     v_mfma_f32_16x16x32_bf16 (128 VGPRs), one MFMA k-step per stage;
//   * LDS image of 64-byte rows: 16-byte chunk c of row r is stored at slot c ^ ((-(r >> 2)) & 3) -- with four rows per
//     256-byte bank window this is the permutation that makes every 16-lane group of a ds_read_b128 fragment read hit 16
//     distinct slots; as in v2 the swizzle is applied to the per-lane SOURCE address of the LDS-DMA;
//   * one counted s_waitcnt vmcnt + one raw s_barrier per stage; a DMA group is issued in every iteration (zero page past
//     the end of the K range) so the count is uniform; one tile per workgroup; the drained ring is the transpose buffer of
//     the LDS-staged epilogue (gemm_args.h).
#include <cstdlib>

#include "common.h"
#include "../../include/mxdenoise.h"
#include "gemm_args.h"

#ifndef MX_EXP
#define MX_EXP 0   // tools/exp_build.sh: 1 = no MFMA, 2 = no LDS-DMA inside the K loop (diagnostics only)
#endif

namespace mx {

__device__ __attribute__((aligned(64))) unsigned int g_zero_page3[1024 / 4] = {0};

constexpr int BM3 = 256;
constexpr int BN3 = 256;
constexpr int BK3 = 32;

__device__ __forceinline__ int swz3(int row, int chunk) { return chunk ^ ((-(row >> 2)) & 3); }

__device__ __forceinline__ void glds16_3(const void* gsrc, void* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

template <int NSTG> __device__ __forceinline__ void wait_stage_landed();
template <> __device__ __forceinline__ void wait_stage_landed<5>() { asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); }
template <> __device__ __forceinline__ void wait_stage_landed<4>() { asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }

template <int NSTG>
__global__ __launch_bounds__(512, 2) void gemm_v3_kernel(const GemmArgs p) {
  constexpr int NI = 4;                        // 16-wide feature blocks per wave (64 features)
  constexpr int MI = 8;                        // 16-wide token blocks per wave (128 tokens)
  constexpr int LOADS = 4;                     // per-thread DMA instructions per stage: 2 for X, 2 for W
  constexpr int STAGE_ELEMS = (BM3 + BN3) * BK3;
  __shared__ __attribute__((aligned(16))) bf16_t smem[NSTG * STAGE_ELEMS];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2;                    // 0..1
  const int wn = wave & 3;                     // 0..3
  const int mt = (p.M + BM3 - 1) / BM3;
  const int total_tiles = mt * (p.N / BN3);
  const int nk = p.K / BK3;
  const char* zero = reinterpret_cast<const char*>(g_zero_page3);

  // ---- issue side: cursor (tile, k stage) of the next DMA group and ready-made per-thread source pointers for it.
  //      issue_group() is branch-free (it shares a basic block with the MFMAs so each LDS-DMA can sit in an MFMA shadow);
  //      advance_cursor() holds the control flow and runs after the MFMAs. ----
  int is_tile = blockIdx.x;
  int is_kt = 0;
  const char* xsrc[2];
  const char* wsrc[2];
  auto setup_tile = [&](int t) __attribute__((always_inline)) {
    const int m0 = (t % mt) * BM3;
    const int n0 = (t / mt) * BN3;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int q = i * 512 + tid;             // LDS slot: row q>>2, slot q&3 holds logical chunk swz3(row, slot)
      const int row = q >> 2;
      const int ch = swz3(row, q & 3);
      const int m = m0 + row;
      const int mc = m < p.M ? m : p.M - 1;    // clamped rows are computed and discarded by the epilogue mask
      xsrc[i] = reinterpret_cast<const char*>(p.a) + (gemm_in_row(p, mc) * p.lda + ch * 8) * 2;
      wsrc[i] = reinterpret_cast<const char*>(p.w) + ((long)(n0 + row) * p.K + ch * 8) * 2;
    }
  };
  auto issue_group = [&](int stage) __attribute__((always_inline)) {
    bf16_t* st = smem + stage * STAGE_ELEMS;
    bf16_t* sw = st + BM3 * BK3;
#pragma unroll
    for (int i = 0; i < 2; ++i) glds16_3(xsrc[i], st + (i * 512 + wave * 64) * 8);
#pragma unroll
    for (int i = 0; i < 2; ++i) glds16_3(wsrc[i], sw + (i * 512 + wave * 64) * 8);
  };
  auto advance_cursor = [&]() __attribute__((always_inline)) {
    if (is_tile >= total_tiles) return;        // parked on the zero page
    if (++is_kt == nk) {
      is_kt = 0;
      is_tile = total_tiles;                   // past the end of the K range: same instruction count, harmless bytes
#pragma unroll
      for (int i = 0; i < 2; ++i) { xsrc[i] = zero + lane * 16; wsrc[i] = zero + lane * 16; }
      return;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) { xsrc[i] += BK3 * 2; wsrc[i] += BK3 * 2; }
  };

  // ---- fragment addresses: lane (fr, fq) reads row base + fr, chunk fq of a 16-row block ----
  const int fr = lane & 15;
  const int fq = lane >> 4;
  const int frag_off = fr * BK3 + swz3(fr, fq) * 8;                 // elements; the swizzle depends on (row >> 2) & 3 = fr >> 2
  const int x_off = (wm * 128) * BK3 + frag_off;                    // + j * 16 * BK3
  const int w_off = BM3 * BK3 + (wn * 64) * BK3 + frag_off;         // + i * 16 * BK3

  setup_tile(is_tile);
#pragma unroll
  for (int s = 0; s < NSTG - 1; ++s) { issue_group(s); advance_cursor(); }

  int stage = 0;
  const int tile = blockIdx.x;
  {
    f32x4 acc[NI][MI];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < MI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int kt = 0; kt < nk; ++kt) {
      wait_stage_landed<NSTG>();
      __builtin_amdgcn_s_barrier();
      const bf16_t* sb = smem + stage * STAGE_ELEMS;
      bf16x8 wf[NI], xf[MI];
#pragma unroll
      for (int i = 0; i < NI; ++i) wf[i] = *reinterpret_cast<const bf16x8*>(sb + w_off + i * 16 * BK3);
#pragma unroll
      for (int j = 0; j < MI / 2; ++j) xf[j] = *reinterpret_cast<const bf16x8*>(sb + x_off + j * 16 * BK3);
#if MX_EXP != 2
      issue_group(stage == 0 ? NSTG - 1 : stage - 1);  // the stage read in the previous iteration, which every wave has left
#endif
#pragma unroll
      for (int j = MI / 2; j < MI; ++j) xf[j] = *reinterpret_cast<const bf16x8*>(sb + x_off + j * 16 * BK3);
#if MX_EXP != 1
#pragma unroll
      for (int j = 0; j < MI; ++j)
#pragma unroll
        for (int i = 0; i < NI; ++i)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf[j], acc[i][j], 0, 0, 0);
#endif
      // schedule: the first 8 fragment reads, then MFMAs with the DMA issue and the other 4 reads in their shadows
      __builtin_amdgcn_sched_group_barrier(0x100, NI + MI / 2, 0);
#pragma unroll
      for (int s = 0; s < LOADS; ++s) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      }
#pragma unroll
      for (int s = 0; s < MI / 2; ++s) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, NI * MI - 2 * LOADS - MI / 2, 0);
#if MX_EXP != 2
      advance_cursor();
#endif
      stage = stage == NSTG - 1 ? 0 : stage + 1;
    }

    const int m0 = (tile % mt) * BM3, n0 = (tile / mt) * BN3;
#if MX_EXP == 4   // no epilogue: keep the accumulators alive with a store that never executes on real data
    {
      float t = 0.f;
      for (int i = 0; i < NI; ++i) for (int j = 0; j < MI; ++j) for (int q = 0; q < 4; ++q) t += acc[i][j][q];
      if (t == 12345.678f) reinterpret_cast<bf16_t*>(p.c)[m0 + n0] = f32_to_bf16(t);
    }
#else
    // every DMA has landed and every wave has left the K loop: the ring becomes the epilogue's transpose buffer
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    float* slab = reinterpret_cast<float*>(smem);
    if (p.flags & MX_EPI_GEGLU) gemm_epilogue_staged<NI, MI, 2, 4, true>(p, acc, slab, m0, n0, wm, wn, fr, fq, tid);
    else gemm_epilogue_staged<NI, MI, 2, 4, false>(p, acc, slab, m0, n0, wm, wn, fr, fq, tid);
#endif
  }
}

int launch_v3(hipStream_t s, const GemmArgs& a) {
  static const int nstg = [] { const char* e = getenv("MX_V3_STAGES"); return e ? atoi(e) : 4; }();
  dim3 grid(cdiv(a.M, BM3) * (a.N / BN3)), block(512);
  if (nstg == 5) hipLaunchKernelGGL((gemm_v3_kernel<5>), grid, block, 0, s, a);
  else hipLaunchKernelGGL((gemm_v3_kernel<4>), grid, block, 0, s, a);
  return 0;
}

}  // namespace mx
