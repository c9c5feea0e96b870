// Wave-tile microbenchmark (round 4): the MFMA + fragment-read part of a 256 x 256 x 64 K tile, re-read from a fixed LDS image, under the chip's own clock management.
//   8 waves (2 per SIMD), wave tile 128 x 64  : 24 ds_read_b128 per 64 MFMAs  -- the shipped 256 x 256 kernel's shape
//   4 waves (1 per SIMD), wave tile 128 x 128 : 32 ds_read_b128 per 128 MFMAs -- a 512-register kernel family (DESIGN.md "what comes next")
// No global traffic in the loop, one barrier per K tile, random bf16 operands.  Prints TFLOP/s of both at one workgroup per CU.
// Build + run on the GPU box: hipcc -O3 --offload-arch=gfx950 tools/exp/wave_tile_bench.hip -o /tmp/wtb && /tmp/wtb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64, WAVES == 8 ? 2 : 1) void wave_tile_kernel(const unsigned short* __restrict__ init, float* __restrict__ out, int iters) {
  constexpr int WN = WAVES == 8 ? 4 : 2;       // wave columns; wave rows = 2
  constexpr int NI = 256 / WN / 16;            // 16-wide feature blocks per wave (4 or 8)
  constexpr int MI = 8;                        // 16-wide token blocks per wave (128 tokens)
  __shared__ __attribute__((aligned(16))) unsigned short smem[2 * 256 * 64];     // X tile | W tile, 128-byte rows, chunk ^= (row >> 1) & 7
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  for (int c = tid; c < 2 * 256 * 8; c += WAVES * 64) {
    const int row = c >> 3, ch = c & 7;
    *reinterpret_cast<uint4*>(&smem[row * 64 + ((ch ^ ((row >> 1) & 7)) * 8)]) = *reinterpret_cast<const uint4*>(&init[(size_t)((row * 8 + ch) * 8 + blockIdx.x * 64) % (1 << 20)]);
  }
  __syncthreads();
  const int fr = lane & 15, fq = lane >> 4;
  f32x4 acc[NI][MI];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < MI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const char* xs = reinterpret_cast<const char*>(smem);
  const char* ws = xs + 256 * 64 * 2;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 xf[MI], wf[NI];
#pragma unroll
      for (int j = 0; j < MI; ++j) {
        const int row = wm * 128 + j * 16 + fr;
        xf[j] = *reinterpret_cast<const bf16x8*>(xs + row * 128 + (((ks * 4 + fq) ^ ((row >> 1) & 7)) * 16));
      }
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int row = wn * (16 * NI) + i * 16 + fr;
        wf[i] = *reinterpret_cast<const bf16x8*>(ws + row * 128 + (((ks * 4 + fq) ^ ((row >> 1) & 7)) * 16));
      }
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < MI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf[j], acc[i][j], 0, 0, 0);
    }
    asm volatile("" ::: "memory");             // the fragments are re-read every K tile, as from a ring that moves on
    __builtin_amdgcn_s_barrier();
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < MI; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  out[blockIdx.x * (WAVES * 64) + tid] = s;
}

template <int WAVES>
static double run(const unsigned short* init, float* out, int ncu, int iters) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(wave_tile_kernel<WAVES>, dim3(ncu), dim3(WAVES * 64), 0, 0, init, out, iters);
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int r = 0; r < 20; ++r) hipLaunchKernelGGL(wave_tile_kernel<WAVES>, dim3(ncu), dim3(WAVES * 64), 0, 0, init, out, iters);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms = 0; hipEventElapsedTime(&ms, a, b);
  return 20.0 * ncu * (double)iters * 256.0 * 256.0 * 64.0 * 2.0 / (ms * 1e-3) / 1e12;
}

int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int ncu = p.multiProcessorCount;
  std::vector<unsigned short> h(1 << 20);
  srand(1);
  for (auto& v : h) { const float f = (rand() / (float)RAND_MAX - 0.5f) * 2.f; unsigned u; memcpy(&u, &f, 4); v = (unsigned short)(u >> 16); }
  unsigned short* init; float* out;
  hipMalloc(&init, h.size() * 2); hipMalloc(&out, (size_t)ncu * 512 * 4);
  hipMemcpy(init, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  for (int rep = 0; rep < 3; ++rep) {
    const double t8 = run<8>(init, out, ncu, 4000), t4 = run<4>(init, out, ncu, 4000);
    printf("%d CUs, random bf16, 4000 K tiles per launch: 8 waves x (128 x 64) %.0f TFLOP/s | 4 waves x (128 x 128) %.0f TFLOP/s | ratio %.3f\n", ncu, t8, t4, t4 / t8);
  }
  return 0;
}
