// bf16 MFMA GEMM / implicit-GEMM conv3x3, large-tile pipelined variant for gfx950 (MI355X).
//
// Same math, orientation, swizzle and epilogue semantics as gemm_bf16.hip (the generic fallback); what changes is the
// schedule, built for the SDXL / SD3.5 step shapes (M = B*H*W in {8192, 32768, 131072}, N in multiples of 320 or 128):
//   * tile 256 tokens x BN features (BN = 160 or 128) x BK 64, 512 threads = 8 waves as 4(m) x 2(n): every SDXL
//     GEMM/conv at UNet batch 8 then decomposes into a multiple of 256 tiles -- whole rounds of the chip's 256 CUs,
//     no tail round (with 128x128 tiles the N=1280 layers ran 640 tiles = 2.5 rounds); one tile per workgroup;
//   * operands go HBM/L2 -> LDS directly (global_load_lds_dwordx4, no staging VGPRs / ds_write), XOR swizzle applied on
//     the per-lane SOURCE address (the LDS image of an LDS-DMA is lane-linear; cdna guide rule 21);
//   * 3-stage LDS ring, counted s_waitcnt vmcnt(N) + raw s_barrier, one barrier per K tile (cdna guide "Pipelining
//     across barriers").  All LDS lives in ONE __shared__ array and the K loop contains no ordinary global load.  A DMA
//     group is issued in EVERY iteration (past the end of the K range it reads a zero page into a stage nobody reads), so
//     one counted wait serves every iteration.  The DMA issue is branch-free and shares a basic block with the MFMAs:
//     each LDS-DMA and the fragment reads of k-step 1 sit in MFMA shadows (sched_group_barrier); all control flow of the
//     loader (next K tile / next conv tap) runs after the MFMAs;
//   * conv3x3: per-row source pointers are recomputed only when the tap changes (every Cin/64 K tiles) and otherwise
//     just advance by one K tile; out-of-image taps and rows beyond M walk a zero page instead of branching;
//   * epilogue: gemm_epilogue_regs (gemm_args.h) transposes the accumulators in registers, so global stores and residual loads move whole
//     128-byte lines per token without LDS or a barrier (round 1 staged the tile through the drained ring instead; that form bought 20-40 %
//     over per-lane stores and was replaced in round 2).
#include <cstdlib>

#include "common.h"
#include "../../include/mxdenoise.h"
#include "gemm_args.h"


namespace mx {

// zero page the loaders read for padding taps / past-the-end DMAs: as long as the widest input channel count (2*Cin bytes)
constexpr int kZeroPageBytes = 16384;
__device__ __attribute__((aligned(64))) unsigned int g_zero_page[kZeroPageBytes / 4] = {0};

constexpr int BK2 = 64;
// LDS ring depth.  The 256-row tiles fill the 160 KB with three stages (loads two K tiles ahead).  The 128-row tiles, which serve the small
// launches (one request, mixed batches: M <= ~4k rows), have smaller stages, and a small launch is LATENCY-bound: a CU streams its operands
// from HBM / a remote L2 at ~2 us per round trip, so with two tiles in flight an iteration cannot be shorter than ~1 us whatever the tile
// (measured: 22.5 us for M 512, 23.2 us for M 2048 at N 1280, K 1280 = 20 K tiles).  They therefore run four (BN 160: 147 KB) or five
// (BN 128: 160 KB) stages, loads three / four tiles ahead.
template <int BN, int MI> struct RingDepth { static constexpr int value = MI == 4 ? 3 : (BN == 160 ? 4 : 5); };

__device__ __forceinline__ int swz2(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

// MI: 16-wide token blocks per wave; tile rows BM2 = 64 * MI (256, or 128 for small M); FEAT / GEGLU: the epilogue features compiled in
// (gemm_args.h EPI_F_*; the launcher picks the smallest instantiation that serves the launch)
template <int BN, int MI, bool CONV, int FEAT, bool GEGLU>
__global__ __launch_bounds__(512, 2) void gemm_v2_kernel(const GemmArgs pk) {
  constexpr int BM2 = 64 * MI;
  constexpr int NI = BN / 32;                 // 16-wide feature blocks per wave (wave covers BN/2 features)
  constexpr int WCH = BN * 8;                 // 16-byte chunks of the W tile
  constexpr int XI = BM2 * 8 / 512;           // X load instructions per thread per tile (4, or 2 for the 128-row tile)
  constexpr int WI = (WCH + 511) / 512;       // W load instructions per thread per tile (3 for BN=160, 2 for 128)
  constexpr int LOADS = XI + WI;              // per-thread DMA instructions per K tile
  constexpr int STAGE_ELEMS = (BM2 + BN) * BK2;
  constexpr int NSTAGE = RingDepth<BN, MI>::value;
  __shared__ __attribute__((aligned(16))) bf16_t smem[NSTAGE * STAGE_ELEMS];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1;
  const int wn = wave & 1;
  // this workgroup's output tile; in a grouped launch (gemm_args.h) also its problem: p is that problem from here on
  GemmArgs p = pk;
  int tm, tn;
  // split-K: workgroups [s * tiles, (s + 1) * tiles) are slice s of every tile (tiles % 8 == 0 keeps a tile's slices on one XCD: speed only)
  const int n_tiles = gemm_m_tiles(pk, BM2) * (pk.N / BN);
  const int slice = pk.splitk > 1 ? (int)blockIdx.x / n_tiles : 0;
  const int tile_id = (int)blockIdx.x - slice * n_tiles;
  gemm_tile_of_block(tile_id, gemm_m_tiles(pk, BM2), pk.N / BN, pk.xcd_map, tm, tn);
  gemm_select_seg(p, pk, tm);
  const int nk_all = p.K / BK2;
  const int k_first = pk.splitk > 1 ? (int)((long)nk_all * slice / pk.splitk) : 0;
  const int nk = pk.splitk > 1 ? (int)((long)nk_all * (slice + 1) / pk.splitk) - k_first : nk_all;
  const char* zero = reinterpret_cast<const char*>(g_zero_page);
  const int cs = tid & 7;
  const int tiles_per_tap = CONV ? p.Cin / BK2 : 1;

  // ---- issue-side state: the (tile, K tile) the NEXT DMA group belongs to, and ready-made per-thread source pointers for
  //      it.  issue_group() is branch-free so that it shares a basic block with the MFMAs (the scheduler can then place each
  //      LDS-DMA in an MFMA shadow); everything with control flow -- moving to the next K tile, tap or output tile, or off the
  //      end of the stream -- happens in advance_cursor(), after the MFMAs. ----
  bool parked = false;          // the cursor ran past the end of this workgroup's (single) tile
  int is_kt = 0;
  const char* xsrc[XI];         // source of the thread's X chunks for the next group
  long xjump[XI];               // split A operand: extra byte step of the thread's X chunks when K reaches k_split (into the second source)
  const char* wsrc[WI];
  int cb[XI], cy[XI], cx[XI];   // CONV: image, y, x of the row's output pixel (input coordinates of the centre tap)
  unsigned xchb[XI];            // CONV: byte offset of the thread's swizzled chunk inside a K tile
  int tap_next = 0, in_tap = 0;

  // CONV: (re)compute the row pointers for tap `tap` at channel offset 0
  auto conv_set_tap = [&](int tap, int cbyte = 0) __attribute__((always_inline)) {
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
      xsrc[i] = (ok ? reinterpret_cast<const char*>(p.a) + off + cbyte : zero) + xchb[i];
    }
  };

  // per-thread sources of K tile 0 of tile `t` (m fastest: workgroups with equal id mod 8 -- one XCD -- share X panels)
  auto setup_tile = [&]() __attribute__((always_inline)) {
    const int m0 = tm * BM2;
    const int n0 = tn * BN;
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      const int row = (i * 512 + tid) >> 3;
      const int ch = swz2(row, cs);           // logical k-chunk this thread fetches for its slot
      const int m = m0 + row;
      if constexpr (!CONV) {
        const int mc = m < p.M ? m : p.M - 1; // clamped rows are computed and discarded by the epilogue mask
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
    // (split-K: this workgroup's K range starts at K tile k_first -- tap k_first / tiles_per_tap, channel tile k_first % tiles_per_tap)
    tap_next = CONV ? k_first / tiles_per_tap : 0; in_tap = CONV ? k_first - tap_next * tiles_per_tap : 0;
    if constexpr (CONV) conv_set_tap(tap_next, in_tap * BK2 * 2);
    else {
#pragma unroll
      for (int i = 0; i < XI; ++i) xsrc[i] += (long)k_first * BK2 * 2;
    }
#pragma unroll
    for (int i = 0; i < WI; ++i) {
      int q = i * 512 + tid;
      if (q >= WCH) q -= WCH;                 // BN=160: the last instruction re-fetches rows 0..31 (same bytes, same slot)
      const int row = q >> 3;
      wsrc[i] = reinterpret_cast<const char*>(p.w) + ((long)(n0 + row) * p.K + (long)k_first * BK2 + swz2(row, cs) * 8) * 2;
    }
  };
  auto park_on_zero_page = [&]() __attribute__((always_inline)) {            // past the end of the stream: same instruction count, harmless bytes
#pragma unroll
    for (int i = 0; i < XI; ++i) xsrc[i] = zero + lane * 16;
#pragma unroll
    for (int i = 0; i < WI; ++i) wsrc[i] = zero + lane * 16;
  };

  // issue the DMA group at the cursor into ring stage `stage` (no control flow)
  auto issue_group = [&](int stage) __attribute__((always_inline)) {
    bf16_t* st = smem + stage * STAGE_ELEMS;
    bf16_t* sw = st + BM2 * BK2;
#pragma unroll
    for (int i = 0; i < XI; ++i) glds16(xsrc[i], st + (i * 512 + wave * 64) * 8);
#pragma unroll
    for (int i = 0; i < WI; ++i) {
      const int qb = (i * 512 + wave * 64 >= WCH) ? i * 512 + wave * 64 - WCH : i * 512 + wave * 64;  // wave-uniform slot base
      glds16(wsrc[i], sw + qb * 8);
    }
  };
  // move the cursor (and the source pointers) to the next K tile of the stream
  auto advance_cursor = [&]() __attribute__((always_inline)) {
    if (parked) return;
    if (++is_kt == nk) {                      // one tile per workgroup: the rest of the ring slots get harmless bytes
      is_kt = 0;
      parked = true;
      park_on_zero_page();
      return;
    }
#pragma unroll
    for (int i = 0; i < WI; ++i) wsrc[i] += BK2 * 2;
    if constexpr (!CONV) {
      // at K = k_split the A operand continues in its second source (gemm_args.h): one more byte step, selected without a branch
      // (the branchy form of this switch was miscompiled once the epilogue grew: the prologue's second advance lost its increment)
      const bool to_a2 = p.a2 != nullptr && is_kt * BK2 == p.k_split;
#pragma unroll
      for (int i = 0; i < XI; ++i) xsrc[i] += BK2 * 2 + (to_a2 ? xjump[i] : 0L);
    } else {
      if (++in_tap == tiles_per_tap) {
        in_tap = 0;
        conv_set_tap(++tap_next);
      } else {
#pragma unroll
        for (int i = 0; i < XI; ++i) xsrc[i] += BK2 * 2;
      }
    }
  };
  auto issue_next = [&](int stage) __attribute__((always_inline)) { issue_group(stage); advance_cursor(); };   // prologue form

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
      const int row = wm * 16 * MI + j * 16 + fr;
      xf[j] = *reinterpret_cast<const bf16x8*>(&sx[row * BK2 + swz2(row, ks * 4 + fq) * 8]);
    }
  };
  constexpr int NM = NI * MI;
  constexpr int NF = NI + MI;

  setup_tile();
#pragma unroll
  for (int st = 0; st < NSTAGE - 1; ++st) issue_next(st);

  int stage = 0;   // ring stage of the K tile being computed (stream position modulo 3)
  {
    f32x4 acc[NI][MI];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < MI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float ln_rstd[MI];                          // folded LayerNorm (gemm_ln_init): while the first operand tiles are in flight
#pragma unroll
    for (int j = 0; j < MI; ++j) ln_rstd[j] = 1.0f;
    if constexpr (!CONV) {
      if (p.ln_stats != nullptr) {
        gemm_ln_init<NI, MI>(p, acc, tm * BM2 + wm * 16 * MI, tn * BN + wn * (BN / 2), fr, fq, ln_rstd);
        if (slice != 0) {                      // split-K: the -mean * colsum term enters the sum once, through slice 0
#pragma unroll
          for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < MI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
    }

    for (int kt = 0; kt < nk; ++kt) {
      // all but the NSTAGE - 2 youngest DMA groups of this thread have completed => the K tile of this iteration has landed
      constexpr int INFLIGHT = LOADS * (NSTAGE - 2);
      if constexpr (INFLIGHT == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
      else if constexpr (INFLIGHT == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else if constexpr (INFLIGHT == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
      else if constexpr (INFLIGHT == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      else static_assert(INFLIGHT == 6 || INFLIGHT == 7 || INFLIGHT == 10 || INFLIGHT == 12, "counted wait");
      __builtin_amdgcn_s_barrier();
      bf16x8 wf0[NI], xf0[MI], wf1[NI], xf1[MI];
      load_frags(stage, 0, wf0, xf0);
      const int st2 = stage >= 1 ? stage - 1 : NSTAGE - 1;   // (g + NSTAGE - 1) % NSTAGE: last read in iteration g-1, which every wave has left
      issue_group(st2);
      load_frags(stage, 1, wf1, xf1);
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < MI; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf0[i], xf0[j], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < MI; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf1[i], xf1[j], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, NF, 0);                 // fragment reads of k-step 0
#pragma unroll
      for (int s = 0; s < LOADS; ++s) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                // MFMA
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                // one LDS-DMA (VMEM read)
      }
#pragma unroll
      for (int s = 0; s < (NF + 1) / 2; ++s) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                // fragment reads of k-step 1
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 2 * NM - LOADS - (NF + 1) / 2, 0);
      advance_cursor();
      stage = stage == NSTAGE - 1 ? 0 : stage + 1;
    }

    const int m0 = tm * BM2, n0 = tn * BN;
    if (pk.splitk > 1) {                       // only the last-arriving slice of the tile goes on, with the sum of all slices
      if (!splitk_combine<NI, MI>(pk, acc, tile_id, slice, BM2 * BN, reinterpret_cast<volatile int*>(smem))) return;
    }
    // register-exchange epilogue (gemm_args.h): no LDS, no barrier; the past-the-end DMAs are drained before the workgroup retires
    static_assert(!GEGLU || (NI % 4 == 0 && !CONV), "the gated epilogue pairs whole 32-feature halves");
    gemm_epilogue_regs<NI, MI, GEGLU, true, true, true, FEAT>(p, acc, m0 + wm * 16 * MI, n0 + wn * (BN / 2), fr, fq, ln_rstd);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
}

// bn: 160 or 128 features per tile.  This file serves the 128-row tiles (small M: one request, mixed batches); the 256-row tiles run the
// ping-pong schedule of gemm_bf16_v5.hip (the lock-step 256-row instantiation it replaced: git history, A/B in profiles/r03_*gemm_bench*).
int launch_v2(hipStream_t s, const GemmArgs& a, bool conv, int bn, int rows) {
  (void)rows;
  const int tiles = (a.nseg > 0 ? a.mt_total : cdiv(a.M, 128)) * (a.N / bn) * (a.splitk > 1 ? a.splitk : 1);
  dim3 grid(tiles), block(512);
#define MX_V2(BN_, CONV_, FEAT_, GEGLU_) hipLaunchKernelGGL((gemm_v2_kernel<BN_, 2, CONV_, FEAT_, GEGLU_>), grid, block, 0, s, a)
  const int feat = gemm_epi_features(a.flags);
  if (a.flags & MX_EPI_GEGLU) {               // (pick_tile: 128 features only)
    if (feat & EPI_F_ACT) MX_V2(128, false, EPI_F_ACT, true); else MX_V2(128, false, 0, true);
  } else if (conv) {
    if (bn == 160) { if (feat == 0) MX_V2(160, true, 0, false); else MX_V2(160, true, EPI_F_ALL, false); }
    else { if (feat == 0) MX_V2(128, true, 0, false); else MX_V2(128, true, EPI_F_ALL, false); }
  } else if (bn == 160) {
    if (feat == 0) MX_V2(160, false, 0, false); else if (feat == EPI_F_QKV) MX_V2(160, false, EPI_F_QKV, false); else MX_V2(160, false, EPI_F_ALL, false);
  } else {
    if (feat == 0) MX_V2(128, false, 0, false); else if (feat == EPI_F_QKV) MX_V2(128, false, EPI_F_QKV, false); else MX_V2(128, false, EPI_F_ALL, false);
  }
#undef MX_V2
  return 0;
}

}  // namespace mx
