// bf16 MFMA GEMM / implicit-GEMM conv3x3, 256 tokens x {160, 128} features, PING-PONG schedule of the two wave-row pairs (round 3).
//
// Same tile, loader (LDS-DMA with the XOR swizzle on the source address, zero page for padding taps, per-tap row pointers), math, orientation
// and epilogue as the 256-row form of gemm_bf16_v2.hip -- what changes is how the eight waves share the CU.  gemm_v2's loop is lock-step: all
// eight waves wait, pass ONE barrier per K tile, read their 18 fragments together (LDS saturated, matrix pipe idle), then compete for the
// matrix pipe together.  Measured there (profiles/r02_g_shape_profile_b4.txt): 1.15 us per K tile against 0.62 us of matrix-pipe time and
// 0.62 us of operand streaming -- the two ADD instead of overlapping.  gemm_v4 (256 x 256) removed that for the large-N launches; this kernel
// does it for the N <= 1280 family and the convs, a third of the step each.
//
//   waves 0-3 (token rows 0-127 of the tile) = group A, waves 4-7 (rows 128-255) = group B: every SIMD holds one wave of each group.
//   per K tile and wave:   L | M        (| = s_barrier; group B runs the same program ONE barrier behind group A)
//     L  read ALL fragments of the K tile (10 W + 8 X ds_read_b128 for BN 160), issue the wave's share of the LDS-DMA for K tile t + 2,
//        move the loader's cursor (tap changes of the conv included), wait until its own DMA of tile t + 1 has landed (counted vmcnt) and
//        its fragment reads have returned (lgkmcnt 0)
//     M  40 MFMAs on registers only
//   so in every phase one group streams operands while the other owns the matrix pipe.
//   * LDS: the same three-stage ring (all 160 KB).  K tile t + 2 overwrites the stage of tile t - 1, last read by group B one phase before
//     group A issues (reads retired by the lgkmcnt(0) in front of the barrier: cdna guide, "restage a buffer 1 phase after when an lgkmcnt
//     before the reading phase's first barrier retired those reads") and two phases before group B issues;
//   * RAW: a wave's DMA of tile t + 1 is waited for in its L phase of tile t, at least one barrier before either group reads it;
//   * the groups re-align for the register-exchange epilogue (group A takes one extra barrier after the loop, group B one before it).
#include <cstdlib>

#include "common.h"
#include "../../include/mxdenoise.h"
#include "gemm_args.h"
#include "gemm_v5_body.h"

namespace mx {

// MI: 16-wide token blocks per wave; tile rows BM5 = 64 * MI (256, or 128 for small M); FEAT / GEGLU: the epilogue features compiled in
// (gemm_args.h EPI_F_*; the launcher picks the smallest instantiation that serves the launch)
// VEC: the per-sample vectors (row bias, gate) are compiled in -- 40 registers of the epilogue; without them the QKV form does not spill
template <int BN, int MI, bool CONV, int FEAT, bool GEGLU, bool VEC>
__global__ __launch_bounds__(512, 2) void gemm_v5_kernel(const GemmArgs pk) {
  constexpr int STAGE_ELEMS = (64 * MI + BN) * BK5;
  __shared__ __attribute__((aligned(16))) bf16_t smem[NSTAGE5 * STAGE_ELEMS];
  int tm, tn;
  gemm_tile_of_block(blockIdx.x, gemm_m_tiles(pk, 64 * MI), pk.N / BN, pk.xcd_map, tm, tn);
  gemm_v5_tile<BN, MI, CONV, FEAT, GEGLU, VEC>(pk, tm, tn, smem);
}

// bn: 160 or 128 features per tile; rows: 256, or 128 when the 256-row tiling would leave most CUs idle (small M)
int launch_v5(hipStream_t s, const GemmArgs& a, bool conv, int bn, int rows) {
  const int tiles = (a.nseg > 0 ? a.mt_total : cdiv(a.M, rows)) * (a.N / bn);
  dim3 grid(tiles), block(512);
#define MX_V5(BN_, CONV_, FEAT_, GEGLU_, VEC_) hipLaunchKernelGGL((gemm_v5_kernel<BN_, 4, CONV_, FEAT_, GEGLU_, VEC_>), grid, block, 0, s, a)
  (void)rows;                                 // (the 128-row instantiation MI = 2 was measured and is not built: see the dispatcher in gemm_bf16.hip)
  const int feat = gemm_epi_features(a.flags);
  const bool vec = a.rowbias || a.gate;       // per-sample vectors: compiled in only where asked for
  if (a.flags & MX_EPI_GEGLU) {               // (pick_tile: 128 features only; the gated epilogue takes no per-sample vectors)
    if (feat & EPI_F_ACT) MX_V5(128, false, EPI_F_ACT, true, false); else MX_V5(128, false, 0, true, false);
  } else if (conv) {
    if (bn == 160) { if (feat == 0 && !vec) MX_V5(160, true, 0, false, false); else if (feat == 0) MX_V5(160, true, 0, false, true); else MX_V5(160, true, EPI_F_ALL, false, true); }
    else { if (feat == 0 && !vec) MX_V5(128, true, 0, false, false); else if (feat == 0) MX_V5(128, true, 0, false, true); else MX_V5(128, true, EPI_F_ALL, false, true); }
  } else if (bn == 160) {
    if (feat == 0 && !vec) MX_V5(160, false, 0, false, false); else if (feat == EPI_F_QKV && !vec) MX_V5(160, false, EPI_F_QKV, false, false);
    else MX_V5(160, false, EPI_F_ALL, false, true);
  } else {
    if (feat == 0 && !vec) MX_V5(128, false, 0, false, false); else if (feat == EPI_F_QKV && !vec) MX_V5(128, false, EPI_F_QKV, false, false);
    else MX_V5(128, false, EPI_F_ALL, false, true);
  }
#undef MX_V5
  return 0;
}

}  // namespace mx

#if defined(MX_EXP) && MX_EXP == 8
extern "C" int mx_debug_v5_stamps(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(mx::g_v5_stamps), sizeof(mx::g_v5_stamps)); }
#endif
