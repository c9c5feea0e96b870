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
//     reason to drain the DMA queue early.
//   * conv3x3: out-of-image taps and rows beyond M read a 16-byte zero page instead of branching.
#include "common.h"
#include "../../include/mxdenoise.h"
#include "gemm_args.h"

namespace mx {

__device__ __attribute__((aligned(64))) unsigned int g_zero_page[16] = {0};

constexpr int BM2 = 256;
constexpr int BK2 = 64;
constexpr int NSTAGE = 3;

__device__ __forceinline__ int swz2(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

template <int BN, bool CONV>
__global__ __launch_bounds__(512, 2) void gemm_v2_kernel(const GemmArgs p) {
  constexpr int NI = BN / 32;                 // 16-wide feature blocks per wave (wave covers BN/2 features)
  constexpr int MI = 4;                       // 16-wide token blocks per wave (wave covers 64 tokens)
  constexpr int XCH = BM2 * 8;                // 16-byte chunks of the X tile
  constexpr int WCH = BN * 8;                 // ... of the W tile
  constexpr int XI = XCH / 512;               // X load instructions per thread per tile (4)
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
  const int m0 = (blockIdx.x % mt) * BM2;     // m fastest: workgroups sharing an XCD (id mod 8) share W panels
  const int n0 = (blockIdx.x / mt) * BN;
  const int nk = p.K / BK2;
  const bf16_t* zero = reinterpret_cast<const bf16_t*>(g_zero_page);

  // ---- per-thread source descriptors (thread -> LDS chunk slot q = i*512 + tid; row = q>>3, slot c' = q&7) ----
  const int cs = tid & 7;
  const bf16_t* xptr[XI];   // GEMM: row base + swizzled chunk (advances by BK2 per tile)
  int cb[XI], cy[XI], cx[XI], xch[XI];
#pragma unroll
  for (int i = 0; i < XI; ++i) {
    const int row = (i * 512 + tid) >> 3;
    const int ch = swz2(row, cs);             // logical k-chunk this thread fetches for its slot
    xch[i] = ch;
    const int m = m0 + row;
    if constexpr (!CONV) {
      xptr[i] = (m < p.M) ? p.a + (long)m * p.lda + ch * 8 : nullptr;
    } else {
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
  const bf16_t* wptr[WI];
#pragma unroll
  for (int i = 0; i < WI; ++i) {
    int q = i * 512 + tid;
    if (q >= WCH) q -= WCH;                   // BN=160: the last instruction re-fetches rows 0..31 (same bytes, same slot)
    const int row = q >> 3;
    wptr[i] = p.w + (long)(n0 + row) * p.K + swz2(row, cs) * 8;
  }

  auto issue_tile = [&](int kt) {
    bf16_t* st = smem + (kt % NSTAGE) * STAGE_ELEMS;
    const int k0 = kt * BK2;
    if constexpr (!CONV) {
#pragma unroll
      for (int i = 0; i < XI; ++i) {
        const bf16_t* src = xptr[i] ? xptr[i] + k0 : zero;
        glds16(src, st + (i * 512 + wave * 64) * 8);
      }
    } else {
      const int tap = k0 / p.Cin;
      const int c0 = k0 - tap * p.Cin;
      const int dy = tap / 3 - 1;
      const int dx = tap - (tap / 3) * 3 - 1;
      const int Hv = p.Hin << p.up, Wv = p.Win << p.up;
      const int P = p.corner_patch;
#pragma unroll
      for (int i = 0; i < XI; ++i) {
        int iy = cy[i] + dy;
        const int ix = cx[i] + dx;
        if (P > 0 && dy != 0 && dx != 0) {
          const bool cross_r = ((iy + P) / P) != ((cy[i] + P) / P);
          const bool cross_c = ((ix + P) / P) != ((cx[i] + P) / P);
          if (cross_r && cross_c) iy = cy[i];
        }
        const bool ok = (cb[i] >= 0) && (iy >= 0) && (iy < Hv) && (ix >= 0) && (ix < Wv);
        const long off = (((long)cb[i] * p.Hin + (iy >> p.up)) * p.Win + (ix >> p.up)) * p.Cin + c0 + xch[i] * 8;
        const bf16_t* src = ok ? p.a + off : zero;
        glds16(src, st + (i * 512 + wave * 64) * 8);
      }
    }
    bf16_t* sw = st + BM2 * BK2;
#pragma unroll
    for (int i = 0; i < WI; ++i) {
      const int qb = (i * 512 + wave * 64 >= WCH) ? i * 512 + wave * 64 - WCH : i * 512 + wave * 64;  // wave-uniform slot base
      glds16(wptr[i] + k0, sw + qb * 8);
    }
  };

  f32x4 acc[NI][MI];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < MI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15;
  const int fq = lane >> 4;

  issue_tile(0);
  if (nk > 1) issue_tile(1);

  for (int kt = 0; kt < nk; ++kt) {
    // tile kt has landed once at most the younger tile's LOADS are still outstanding
    if (kt + 1 < nk) {
      if constexpr (LOADS == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (kt + 2 < nk) issue_tile(kt + 2);     // its stage was last read in iteration kt-1, which every wave has left

    const bf16_t* sx = smem + (kt % NSTAGE) * STAGE_ELEMS;
    const bf16_t* sw = sx + BM2 * BK2;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 wf[NI], xf[MI];
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
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < MI; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf[j], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
    }
  }

  gemm_epilogue<NI, MI, BN>(p, acc, m0 + wm * 64, n0 + wn * (BN / 2), fr, fq);
}

int launch_v2(hipStream_t s, const GemmArgs& a, bool conv, int bn) {
  const int mt = cdiv(a.M, BM2);
  dim3 grid(mt * (a.N / bn)), block(512);
  if (bn == 160) {
    if (conv) hipLaunchKernelGGL((gemm_v2_kernel<160, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((gemm_v2_kernel<160, false>), grid, block, 0, s, a);
  } else {
    if (conv) hipLaunchKernelGGL((gemm_v2_kernel<128, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((gemm_v2_kernel<128, false>), grid, block, 0, s, a);
  }
  return 0;
}

}  // namespace mx
