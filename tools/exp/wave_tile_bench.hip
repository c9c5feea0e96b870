// Wave-tile microbenchmark (round 4): the MFMA + fragment-read part of a 256 x 256 x 64 K tile, re-read from a fixed LDS image, under the chip's own clock management.
//   8 waves (2 per SIMD), wave tile 128 x 64  : 24 ds_read_b128 per 64 MFMAs  -- the shipped 256 x 256 kernel's shape
//   4 waves (1 per SIMD), wave tile 128 x 128 : 32 ds_read_b128 per 128 MFMAs -- a 512-register kernel family (DESIGN.md "what comes next")
// No global traffic in the loop, one barrier per K tile, random bf16 operands.  Prints TFLOP/s of both at one workgroup per CU.
// Second table (8 waves only): the same loop with an OPERAND STREAM beside it -- 64 KB per K tile fetched into a landing area of LDS that nobody reads (so the
// MFMAs never wait for it): by LDS-DMA (global_load_lds, 8 per thread and K tile, one tile in flight: the shipped loaders' form) or staged through registers
// (global_load_dwordx4 x 8, then ds_write_b128 x 8 one K tile later).  48 distinct streams over the chip, six per XCD (the rest hit in L2): the beyond-L2 traffic of the shipped launches (~2 TB/s).
// Build + run on the GPU box: hipcc -O3 --offload-arch=gfx950 tools/exp/wave_tile_bench.hip -o /tmp/wtb && /tmp/wtb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// STREAM: 0 none, 1 LDS-DMA, 2 register-staged (8 waves only)
template <int WAVES, int STREAM = 0>
__global__ __launch_bounds__(WAVES * 64, WAVES == 8 ? 2 : 1) void wave_tile_kernel(const unsigned short* __restrict__ init, float* __restrict__ out, int iters,
                                                                                  const char* __restrict__ big = nullptr, size_t big_bytes = 0) {
  constexpr int WN = WAVES == 8 ? 4 : 2;       // wave columns; wave rows = 2
  constexpr int NI = 256 / WN / 16;            // 16-wide feature blocks per wave (4 or 8)
  constexpr int MI = 8;                        // 16-wide token blocks per wave (128 tokens)
  __shared__ __attribute__((aligned(16))) unsigned short smem[(STREAM ? 4 : 2) * 256 * 64];     // X tile | W tile (128-byte rows, chunk ^= (row >> 1) & 7) | landing area
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
  [[maybe_unused]] char* land = reinterpret_cast<char*>(smem) + 2 * 256 * 64 * 2;
  [[maybe_unused]] uint4 stage[8];
  for (int it = 0; it < iters; ++it) {
    // K tile `it` of this workgroup's stream: 64 KB at ((it mod 64) * 48 + stream) * 64 KB of the 256-MB buffer.  Six streams per XCD (workgroup id & 7 = XCD, the
    // XCD's 32 workgroups share its six streams): 48 x 64 KB = 3 MB from beyond the L2s per K tile, ~2 TB/s -- what the shipped 256 x 256 launches fetch (PMC)
    [[maybe_unused]] const size_t src = ((size_t)(it & 63) * 48 + (size_t)(blockIdx.x & 7) * 6 + ((blockIdx.x >> 3) % 6)) * 65536 + (size_t)tid * 16;
    if constexpr (STREAM == 1) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(big + src + i * 8192),
                                         (__attribute__((address_space(3))) void*)(land + i * 8192 + wave * 1024), 16, 0, 0);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");     // the previous K tile's eight have landed
    } else if constexpr (STREAM == 2) {
      if (it > 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) *reinterpret_cast<uint4*>(land + i * 8192 + tid * 16) = stage[i];
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) stage[i] = *reinterpret_cast<const uint4*>(big + src + i * 8192);
    }
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
  if constexpr (STREAM == 2) s += __uint_as_float(stage[0].x & 1u);
  if constexpr (STREAM == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < MI; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  out[blockIdx.x * (WAVES * 64) + tid] = s;
}

template <int WAVES, int STREAM = 0>
static double run(const unsigned short* init, float* out, int ncu, int iters, const char* big = nullptr, size_t big_bytes = 0) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((wave_tile_kernel<WAVES, STREAM>), dim3(ncu), dim3(WAVES * 64), 0, 0, init, out, iters, big, big_bytes);
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int r = 0; r < 20; ++r) hipLaunchKernelGGL((wave_tile_kernel<WAVES, STREAM>), dim3(ncu), dim3(WAVES * 64), 0, 0, init, out, iters, big, big_bytes);
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
  char* big; const size_t big_bytes = (size_t)256 << 20;
  hipMalloc(&big, big_bytes); hipMemset(big, 0x3c, big_bytes);
  for (int rep = 0; rep < 3; ++rep) {
    const double t0 = run<8, 0>(init, out, ncu, 4000), t1 = run<8, 1>(init, out, ncu, 4000, big, big_bytes), t2 = run<8, 2>(init, out, ncu, 4000, big, big_bytes);
    printf("8 waves x (128 x 64), operand stream of 64 KB per K tile beside the loop: none %.0f TFLOP/s | LDS-DMA %.0f | through registers %.0f\n", t0, t1, t2);
  }
  return 0;
}
