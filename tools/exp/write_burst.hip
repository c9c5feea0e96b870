// Microbenchmark: store rate of a GEMM epilogue's write pattern.  G workgroups x 512 threads; each writes `reps` 256 x 256 bf16 tiles
// (128 KB) of a row-major [M, N] matrix with 16-byte stores.  pattern = contiguous bytes per row per wave instruction:
//   64  : 16 rows x 64 B   (register-exchange epilogue, round 2)
//   128 : 8 rows x 128 B   (full cache lines; needs one more lane-pair exchange)
//   512 : 2 rows x 512 B   (LDS-staged epilogue, round 1)
//   hipcc --offload-arch=gfx950 -O3 tools/exp/write_burst.hip -o tools/exp/write_burst.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int PAT>
__global__ __launch_bounds__(512) void burst(unsigned short* c, int ldc, int nt, int reps) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int r = 0; r < reps; ++r) {
    const int tile = blockIdx.x + r * gridDim.x;
    const int tm = tile / nt, tn = tile % nt;
    const u32x4 v = {(unsigned)tile, (unsigned)tid, 3u, 4u};
    // a wave owns 128 rows x 64 columns (128 B per row) of the tile, as in the GEMM (waves 2 x 4): 16 instructions of 1 KB
    const int wm = wave >> 2, wn = wave & 3;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      long row; int col;
      if (PAT == 64) { row = wm * 128 + (s >> 1) * 16 + (lane & 15); col = wn * 64 + (s & 1) * 32 + (lane >> 4) * 8; }
      else if (PAT == 128) { row = wm * 128 + s * 8 + (lane >> 3); col = wn * 64 + (lane & 7) * 8; }
      else { row = wave * 32 + s * 2 + (lane >> 5); col = (lane & 31) * 8; }      // 512 B per row: waves own whole rows
      *reinterpret_cast<u32x4*>(c + ((long)tm * 256 + row) * ldc + tn * 256 + col) = v;
    }
  }
}

int main() {
  const int M = 32768, N = 4608;
  unsigned short* c;
  hipMalloc(&c, (size_t)M * N * 2);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int grid : {256, 64, 8})
    for (int pat : {64, 128, 512})
      for (int reps : {1, 9}) {
        float best = 1e9;
        for (int it = 0; it < 5; ++it) {
          hipEventRecord(a);
          if (pat == 64) hipLaunchKernelGGL(burst<64>, dim3(grid), dim3(512), 0, 0, c, N, N / 256, reps);
          else if (pat == 128) hipLaunchKernelGGL(burst<128>, dim3(grid), dim3(512), 0, 0, c, N, N / 256, reps);
          else hipLaunchKernelGGL(burst<512>, dim3(grid), dim3(512), 0, 0, c, N, N / 256, reps);
          hipEventRecord(b);
          hipEventSynchronize(b);
          float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
          if (ms < best) best = ms;
        }
        const double bytes = (double)grid * reps * 131072;
        printf("CUs %3d pattern %3d B/row tiles/CU %d: %8.1f us  %6.2f TB/s  %6.1f GB/s per CU  (%.2f us per tile)\n", grid, pat, reps, best * 1e3,
               bytes / (best * 1e-3) / 1e12, bytes / grid / (best * 1e-3) / 1e9, best * 1e3 / reps);
      }
  return 0;
}
