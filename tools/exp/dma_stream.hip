// Microbenchmark: the L2 -> LDS operand stream of the 256 x 256 x 64 GEMM tile in isolation (no MFMA, no LDS reads).
// Question: is the K loop's 44-48 GB/s per CU (profiles/r01_e_gemm_component_removal.txt; r02 stamps: 1.48 us per 64-KB K tile)
// a latency x bytes-in-flight limit (then a deeper ring helps) or a rate limit of the L2 / fabric path?
//   * persistent: one 512-thread workgroup per CU walks tiles t = blockIdx, + grid, ... with the GEMM's XCD-aware tile map;
//   * a K tile = 4 half-tiles of 16 KB (X rows 0-127, X rows 128-255, W rows 0-127, W rows 128-255), 2 x 16-byte LDS-DMA per thread each,
//     swizzled source addresses exactly as the GEMM issues them;
//   * ring of NS 16-KB slots; before issuing half-tile h the wave waits until half-tile h - DEPTH has landed (counted vmcnt), and one
//     s_barrier per K tile keeps the 8 waves together as the GEMM's K loop does.
//   hipcc --offload-arch=gfx950 -O3 tools/exp/dma_stream.hip -o tools/exp/dma_stream.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ void glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
__device__ __forceinline__ void tile_of_block(int b, int mt, int nt, int xcd_map, int& tm, int& tn) {
  constexpr int MB = 4;
  if (!xcd_map || (mt & 7)) { tm = b % mt; tn = b / mt; return; }
  const int xcd = b & 7, local = b >> 3, mloc = mt >> 3, band_tiles = MB * nt, band = local / band_tiles, r = local - band * band_tiles;
  const int rows = (mloc - band * MB) < MB ? (mloc - band * MB) : MB;
  tn = r / rows; tm = ((band * MB + (r - tn * rows)) << 3) + xcd;
}

template <int DEPTH, bool BARRIER>   // DEPTH: half-tiles in flight per wave (each = 2 DMA instructions per thread)
__global__ __launch_bounds__(512, 2) void stream(const unsigned short* a, const unsigned short* w, int M, int N, int K, int xcd_map, int same_panel,
                                                 unsigned* sink) {
  constexpr int NS = 10;
  __shared__ __attribute__((aligned(16))) unsigned short smem[NS * 8192];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), cs = tid & 7;
  const int mt = M / 256, nt = N / 256, nk = K / 64, total = mt * nt;
  int slot = 0;
  long issued = 0;
  for (int tile = blockIdx.x; tile < total; tile += gridDim.x) {
    int tm, tn;
    tile_of_block(same_panel ? 0 : tile, mt, nt, xcd_map, tm, tn);
    const char* src[4][2];
#pragma unroll
    for (int h = 0; h < 4; ++h)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = (i * 512 + tid) >> 3;                     // 0..127 inside the half-tile
        const int r256 = (h & 1) * 128 + row;
        const int ch = cs ^ ((row >> 1) & 7);
        src[h][i] = (h < 2) ? reinterpret_cast<const char*>(a) + ((long)(tm * 256 + r256) * K + ch * 8) * 2
                            : reinterpret_cast<const char*>(w) + ((long)(tn * 256 + r256) * K + ch * 8) * 2;
      }
    for (int kt = 0; kt < nk; ++kt) {
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        // keep at most DEPTH half-tiles in flight: wait until all but the youngest 2 * (DEPTH - 1) DMA instructions have landed
        if (issued >= DEPTH) {
          if constexpr (DEPTH == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          else if constexpr (DEPTH == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
          else if constexpr (DEPTH == 3) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
          else if constexpr (DEPTH == 4) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
          else if constexpr (DEPTH == 6) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
          else if constexpr (DEPTH == 8) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        }
        unsigned short* st = smem + slot * 8192;
#pragma unroll
        for (int i = 0; i < 2; ++i) { glds16(src[h][i], st + (i * 512 + wave * 64) * 8); src[h][i] += 128; }
        slot = slot == NS - 1 ? 0 : slot + 1;
        ++issued;
      }
      if constexpr (BARRIER) __builtin_amdgcn_s_barrier();
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0 && sink) sink[blockIdx.x] = smem[0];
}

template <int DEPTH, bool BARRIER>
static float run(const unsigned short* a, const unsigned short* w, int M, int N, int K, int xcd, int same, unsigned* sink, int grid) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9;
  for (int it = 0; it < 4; ++it) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((stream<DEPTH, BARRIER>), dim3(grid), dim3(512), 0, 0, a, w, M, N, K, xcd, same, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  return best;
}

int main() {
  struct Shape { int M, N, K; } shapes[] = {{32768, 4608, 1536}, {8192, 10240, 1280}};
  for (auto s : shapes) {
    unsigned short *a, *w; unsigned* sink;
    hipMalloc(&a, (size_t)s.M * s.K * 2); hipMalloc(&w, (size_t)s.N * s.K * 2); hipMalloc(&sink, 4096);
    hipMemset(a, 1, (size_t)s.M * s.K * 2); hipMemset(w, 1, (size_t)s.N * s.K * 2);
    const int tiles = (s.M / 256) * (s.N / 256);
    const double bytes = (double)tiles * (s.K / 64) * 65536.0;
    printf("M%d N%d K%d: %d tiles, %.2f GB through the L2 -> LDS path per launch\n", s.M, s.N, s.K, tiles, bytes / 1e9);
    for (int same = 0; same < 2; ++same)
      for (int xcd = 1; xcd >= 0; --xcd) {
        if (same && !xcd) continue;
        float t[8];
        t[0] = run<1, true>(a, w, s.M, s.N, s.K, xcd, same, sink, 256);
        t[1] = run<2, true>(a, w, s.M, s.N, s.K, xcd, same, sink, 256);
        t[2] = run<3, true>(a, w, s.M, s.N, s.K, xcd, same, sink, 256);
        t[3] = run<4, true>(a, w, s.M, s.N, s.K, xcd, same, sink, 256);
        t[4] = run<6, true>(a, w, s.M, s.N, s.K, xcd, same, sink, 256);
        t[5] = run<8, true>(a, w, s.M, s.N, s.K, xcd, same, sink, 256);
        t[6] = run<9, true>(a, w, s.M, s.N, s.K, xcd, same, sink, 256);
        t[7] = run<6, false>(a, w, s.M, s.N, s.K, xcd, same, sink, 256);
        const char* names[8] = {"depth 1", "depth 2", "depth 3", "depth 4", "depth 6", "depth 8", "depth 9", "depth 6 no barrier"};
        for (int i = 0; i < 8; ++i)
          printf("  %s xcd_map %d %-20s %8.1f us  %6.1f GB/s per CU  (%.2f us per 64-KB K tile)\n", same ? "ONE panel pair (all L2 hits)" : "real tile addresses        ",
                 xcd, names[i], t[i] * 1e3, bytes / 256 / (t[i] * 1e-3) / 1e9, t[i] * 1e3 / (tiles / 256.0 * (s.K / 64)));
      }
    hipFree(a); hipFree(w); hipFree(sink);
  }
  return 0;
}
