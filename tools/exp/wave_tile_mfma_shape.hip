// Wave-tile probe, round 5: does the MFMA SHAPE change what the chip sustains under its power limit?  The loop of tools/exp/wave_tile_bench.hip (8 waves, wave tile 128 x 64,
// fragments re-read from a fixed LDS image every K tile, one barrier per K tile, no global traffic) with v_mfma_f32_16x16x32_bf16 (64 per K tile and wave: the shipped
// kernels) against v_mfma_f32_32x32x16_bf16 (32 per K tile and wave: same FLOP, same 24 ds_read_b128, same 128 accumulator registers, half the operand-register reads per FLOP).
// Build + run on the GPU box: hipcc -O3 --offload-arch=gfx950 tools/exp/wave_tile_mfma_shape.hip -o /tmp/wtm && /tmp/wtm
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int SHAPE>   // 0: 16x16x32, 1: 32x32x16
__global__ __launch_bounds__(512, 2) void wave_tile_kernel(const unsigned short* __restrict__ init, float* __restrict__ out, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned short smem[2 * 256 * 64];     // X tile | W tile (128-byte rows, chunk ^= (row >> 1) & 7)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / 4, wn = wave % 4;
  for (int c = tid; c < 2 * 256 * 8; c += 512) {
    const int row = c >> 3, ch = c & 7;
    *reinterpret_cast<uint4*>(&smem[row * 64 + ((ch ^ ((row >> 1) & 7)) * 8)]) = *reinterpret_cast<const uint4*>(&init[(size_t)((row * 8 + ch) * 8 + blockIdx.x * 64) % (1 << 20)]);
  }
  __syncthreads();
  const char* xs = reinterpret_cast<const char*>(smem);
  const char* ws = xs + 256 * 64 * 2;
  float s = 0.f;
  if constexpr (SHAPE == 0) {
    const int fr = lane & 15, fq = lane >> 4;
    f32x4 acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 xf[8], wf[4];
#pragma unroll
        for (int j = 0; j < 8; ++j) { const int row = wm * 128 + j * 16 + fr; xf[j] = *reinterpret_cast<const bf16x8*>(xs + row * 128 + (((ks * 4 + fq) ^ ((row >> 1) & 7)) * 16)); }
#pragma unroll
        for (int i = 0; i < 4; ++i) { const int row = wn * 64 + i * 16 + fr; wf[i] = *reinterpret_cast<const bf16x8*>(ws + row * 128 + (((ks * 4 + fq) ^ ((row >> 1) & 7)) * 16)); }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf[j], acc[i][j], 0, 0, 0);
      }
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  } else {
    const int r = lane & 31, hh = lane >> 5;
    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {                 // 16-deep k-steps: chunk 2 ks + hh of the 128-byte row
        bf16x8 xf[4], wf[2];
#pragma unroll
        for (int j = 0; j < 4; ++j) { const int row = wm * 128 + j * 32 + r; xf[j] = *reinterpret_cast<const bf16x8*>(xs + row * 128 + (((ks * 2 + hh) ^ ((row >> 1) & 7)) * 16)); }
#pragma unroll
        for (int i = 0; i < 2; ++i) { const int row = wn * 64 + i * 32 + r; wf[i] = *reinterpret_cast<const bf16x8*>(ws + row * 128 + (((ks * 2 + hh) ^ ((row >> 1) & 7)) * 16)); }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[i], xf[j], acc[i][j], 0, 0, 0);
      }
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc[i][j][e];
  }
  out[blockIdx.x * 512 + tid] = s;
}

template <int SHAPE>
static double run(const unsigned short* init, float* out, int ncu, int iters) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((wave_tile_kernel<SHAPE>), dim3(ncu), dim3(512), 0, 0, init, out, iters);
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int r = 0; r < 20; ++r) hipLaunchKernelGGL((wave_tile_kernel<SHAPE>), dim3(ncu), dim3(512), 0, 0, init, out, iters);
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
  for (int rep = 0; rep < 4; ++rep) {
    const double t0 = run<0>(init, out, ncu, 4000), t1 = run<1>(init, out, ncu, 4000);
    printf("%d CUs, random bf16, 4000 K tiles per launch, 8 waves x (128 x 64): 16x16x32 %.0f TFLOP/s | 32x32x16 %.0f TFLOP/s | ratio %.3f\n", ncu, t0, t1, t1 / t0);
  }
  return 0;
}
