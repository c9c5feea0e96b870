// 3x3 convolution with a handful of output channels (round 5): the UNet's conv_out, 320 -> 4 channels over 128 x 128 latents (unet.py:514-517).
// As an implicit GEMM it is M = B H W rows, K = 9 Cin, N = 4: the generic 128 x 64 tile kernel pads N to 64 and fetched 617 MB per launch for
// 84 MB of input (132 us at the headline batch; profiles/r05_z_pmc_traffic_sdxl_step.txt).  The work is reading the input once.  Here a wave owns
// 16 consecutive pixels of an image row and all N <= 16 outputs: one v_mfma_f32_16x16x32_bf16 per (tap, 32 input channels) with A = the weights
// (rows >= N zero, held in LDS: N x K bf16) and B = the 16 pixels' channels.  A workgroup is a tile of four rows x 16 columns: per 64-channel chunk
// its halo'd 6 x 18 pixels are staged in LDS as whole 128-byte lines (the next chunk's reads in flight while this one multiplies), and the nine taps
// are nine shifted reads of that stage.  Stride 1, zero padding, the sliced path's halo-corner rule; bias only.
#include <algorithm>
#include <cstdlib>

#include "common.h"
#include "../../include/mxdenoise.h"
#include "gemm_args.h"

namespace mx {

constexpr int kSnPix = 6 * 18;                 // a tile's pixels with their halo: (4 + 2) rows x (16 + 2) columns
constexpr int kSnStride = 144;                 // bytes per staged pixel: 64 channels + 16 of padding (16 pixels x 4 pieces read conflict-free)

__global__ __launch_bounds__(256) void conv3x3_small_n_kernel(const GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];        // weights [N][K] bf16, then one staged 64-channel chunk of the tile
  bf16_t* sw = reinterpret_cast<bf16_t*>(smem);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int j = lane & 15, kq = lane >> 4;
  const int H = p.Hin, W = p.Win, Cin = p.Cin, K = p.K, N = p.N;
  char* sx = smem + (((size_t)N * K * 2 + 255) & ~(size_t)255);
  {
    const int chunks = N * K / 8;                      // 16-byte pieces (K % 8 == 0)
    for (int c = tid; c < chunks; c += 256) *reinterpret_cast<u32x4*>(sw + (long)c * 8) = *reinterpret_cast<const u32x4*>(p.w + (long)c * 8);
  }
  const int tiles_x = (W + 15) >> 4, tiles_y = (H + 3) >> 2;
  const int tiles = p.B * tiles_x * tiles_y;
  const int nch = Cin >> 6;                            // 64-channel chunks (Cin % 64 == 0: the launcher)
  const int P = p.corner_patch;
  const bool wrow = j < N;                              // lane (j, kq) holds weight row n = j as the A operand
  const bf16_t* swl = sw + (long)(wrow ? j : 0) * K + kq * 8;
  const bf16x8 zero8 = __builtin_bit_cast(bf16x8, u32x4{0u, 0u, 0u, 0u});
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  // this thread's pieces of a staged chunk: piece i = pixel i / 8 of the halo'd tile, 16-byte piece i % 8 of its 128 bytes
  int ppy[4], ppx[4], pc16[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int i = tid + 256 * q;
    const int pix = i >> 3;
    pc16[q] = i & 7;
    ppy[q] = pix / 18; ppx[q] = pix - ppy[q] * 18;
  }
  // the whole halo'd tile of one chunk comes in as full 128-byte lines (the direct form asked L1 for 16 half-lines per load and ran at one load per
  // ~64 cycles per CU whatever the occupancy: 87 us); zero padding is zeros in LDS
  auto fetch = [&](u32x4 (&r)[4], int tile, int ch) __attribute__((always_inline)) {
    const int b = tile / (tiles_x * tiles_y);
    const int t = tile - b * tiles_x * tiles_y;
    const int ty = t / tiles_x, tx = t - ty * tiles_x;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int gy = ty * 4 - 1 + ppy[q], gx = tx * 16 - 1 + ppx[q];
      const bool ok = (tid + 256 * q) < kSnPix * 8 && gy >= 0 && gy < H && gx >= 0 && gx < W;
      r[q] = ok ? *reinterpret_cast<const u32x4*>(p.a + ((long)(b * H + gy) * W + gx) * Cin + ch * 64 + pc16[q] * 8) : zero4;
    }
  };
  auto stage = [&](const u32x4 (&r)[4]) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if ((tid + 256 * q) < kSnPix * 8) *reinterpret_cast<u32x4*>(sx + ((tid + 256 * q) >> 3) * kSnStride + pc16[q] * 16) = r[q];
  };
  int tile = blockIdx.x;
  if (tile >= tiles) return;                           // (the launcher starts no more workgroups than tiles)
  u32x4 pre[4];
  fetch(pre, tile, 0);
  __syncthreads();                                      // the weights are in place
  while (tile < tiles) {
    const int b = tile / (tiles_x * tiles_y);
    const int t = tile - b * tiles_x * tiles_y;
    const int ty = t / tiles_x, tx = t - ty * tiles_x;
    const int y = ty * 4 + wave, x = tx * 16 + j;
    // per tap: the staged pixel this lane reads (row wave + 1 + dy, column j + 1 + dx of the halo'd tile; the sliced path's halo-corner rule turns a
    // diagonal tap that leaves the patch through a corner into the horizontal neighbour: norm_silu_concat.cu:210-221, 228-239)
    int sp[9];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int dy = tap / 3 - 1, dx = tap % 3 - 1;
      int ry = wave + 1 + dy;
      if (P > 0 && dy != 0 && dx != 0) {
        const bool cross_r = ((y + dy + P) / P) != ((y + P) / P);
        const bool cross_c = ((x + dx + P) / P) != ((x + P) / P);
        if (cross_r && cross_c) ry = wave + 1;
      }
      sp[tap] = (ry * 18 + j + 1 + dx) * kSnStride + kq * 16;
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int next_tile = tile + gridDim.x;
    for (int ch = 0; ch < nch; ++ch) {
      stage(pre);
      __syncthreads();
      if (ch + 1 < nch) fetch(pre, tile, ch + 1);       // in flight while this chunk multiplies
      else if (next_tile < tiles) fetch(pre, next_tile, 0);
#pragma unroll
      for (int sub = 0; sub < 2; ++sub) {
        bf16x8 xf[9], wf[9];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          xf[tap] = *reinterpret_cast<const bf16x8*>(sx + sp[tap] + sub * 64);
          wf[tap] = wrow ? *reinterpret_cast<const bf16x8*>(swl + tap * Cin + ch * 64 + sub * 32) : zero8;
        }
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[tap], xf[tap], acc, 0, 0, 0);
      }
      __syncthreads();                                  // every wave has read the chunk: the stage may be rewritten
    }
    // lane (j, kq) holds outputs n = 4 kq + {0..3} of pixel x
    const int n = 4 * kq;
    if (y < H && x < W && n < N) {
      f32x4 v = acc;
      if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + n);
      bf16_t* dst = reinterpret_cast<bf16_t*>(p.c) + ((long)(b * H + y) * W + x) * p.ldc + n;
      *reinterpret_cast<u32x2*>(dst) = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
    }
    tile = next_tile;
  }
}

// ---- the mirror case: a handful of INPUT channels (the UNet's conv_in: 4 latent channels -> 320, stored zero-padded to 64 channels per pixel so that the
// tile kernels can read it; unet.py:344).  As an implicit GEMM over the padded input it is K = 9 x 64 with 4 useful channels per tap: 95 us at the headline batch
// for 84 MB of output.  Here K is 9 taps x 8 channels = 72: a lane's MFMA fragment is the first 16 bytes of ONE tap's pixel (three k-steps of four taps; taps 9..11
// are zeros), the repacked weights [N][72] sit in LDS, a wave owns 16 pixels and walks the output channels 80 at a time -- 15 MFMAs, then the 16 x 80 results go
// through a wave-private LDS patch so that they leave as 160 contiguous bytes per pixel.  Bound by the output write.
constexpr int kScWRow = 72;                    // repacked weight row: 9 taps x 8 channels (144 bytes: 16 rows x 4 pieces read conflict-free)
constexpr int kScStage = 176;                  // bytes per staged pixel: 80 channels + 16 of padding

__global__ __launch_bounds__(256) void conv3x3_small_cin_kernel(const GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];        // weights [N][72] bf16, then 4 waves x 16 pixels x 176 bytes of output staging
  bf16_t* sw = reinterpret_cast<bf16_t*>(smem);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int j = lane & 15, kq = lane >> 4;
  const int H = p.Hin, W = p.Win, Cin = p.Cin, N = p.N;
  char* stage = smem + (((size_t)N * kScWRow * 2 + 255) & ~(size_t)255) + wave * (16 * kScStage);
  for (int c = tid; c < N * 9; c += 256) {             // (feature, tap): the first 8 channels of the tap
    const int n = c / 9, tap = c - n * 9;
    *reinterpret_cast<u32x4*>(sw + (long)n * kScWRow + tap * 8) = *reinterpret_cast<const u32x4*>(p.w + ((long)n * 9 + tap) * Cin);
  }
  const int tiles_x = (W + 15) >> 4, tiles_y = (H + 3) >> 2;
  const int tiles = p.B * tiles_x * tiles_y;
  const int P = p.corner_patch;
  const bf16x8 zero8 = __builtin_bit_cast(bf16x8, u32x4{0u, 0u, 0u, 0u});
  __syncthreads();
  for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const int b = tile / (tiles_x * tiles_y);
    const int t = tile - b * tiles_x * tiles_y;
    const int ty = t / tiles_x, tx = t - ty * tiles_x;
    const int y = ty * 4 + wave, x = tx * 16 + j;
    if (y >= H) continue;                               // (wave-uniform; the loop holds no workgroup barrier)
    // B operand: k-step s holds taps 4 s + kq of pixel j
    bf16x8 xf[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      const int tap = 4 * s + kq;
      const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
      int iy = y + dy;
      const int ix = x + dx;
      if (P > 0 && dy != 0 && dx != 0) {                // halo-corner rule of the reference's sliced path (norm_silu_concat.cu:210-221, 228-239)
        const bool cross_r = ((iy + P) / P) != ((y + P) / P);
        const bool cross_c = ((ix + P) / P) != ((x + P) / P);
        if (cross_r && cross_c) iy = y;
      }
      const bool ok = tap < 9 && x < W && iy >= 0 && iy < H && ix >= 0 && ix < W;
      xf[s] = ok ? *reinterpret_cast<const bf16x8*>(p.a + ((long)(b * H + iy) * W + ix) * Cin) : zero8;
    }
    for (int n0 = 0; n0 < N; n0 += 80) {                // (N % 80 == 0: the launcher)
      f32x4 acc[5];
#pragma unroll
      for (int i = 0; i < 5; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const int tap = 4 * s + kq;
#pragma unroll
        for (int i = 0; i < 5; ++i) {                   // A operand: lane (j, kq) holds weight row n0 + 16 i + j, tap 4 s + kq
          const bf16x8 wf = tap < 9 ? *reinterpret_cast<const bf16x8*>(sw + (long)(n0 + 16 * i + j) * kScWRow + tap * 8) : zero8;
          acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xf[s], acc[i], 0, 0, 0);
        }
      }
      // lane (j, kq) holds outputs n0 + 16 i + 4 kq + {0..3} of pixel x: through the wave's patch, out as 160 contiguous bytes per pixel
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        f32x4 v = acc[i];
        if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + n0 + 16 * i + 4 * kq);
        *reinterpret_cast<u32x2*>(stage + j * kScStage + (16 * i + 4 * kq) * 2) = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (one wave: its LDS operations complete in order)
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const int piece = lane + 64 * r;                // 16 pixels x 10 pieces of 16 bytes
        if (piece < 160) {
          const int pj = piece / 10, c16 = piece - pj * 10;
          const u32x4 o = *reinterpret_cast<const u32x4*>(stage + pj * kScStage + c16 * 16);
          const int px = tx * 16 + pj;
          if (px < W) *reinterpret_cast<u32x4*>(reinterpret_cast<bf16_t*>(p.c) + ((long)(b * H + y) * W + px) * p.ldc + n0 + c16 * 8) = o;
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the patch is read before the next 80 channels overwrite it
    }
  }
}

bool conv_small_n_serves(const mx_gemm_desc* d) {
  static const bool off = [] { const char* e = getenv("MX_CONV_SMALL_N"); return e && e[0] == '0'; }();      // A/B: the generic tile kernel
  if (off || d->n_segs != 0 || d->N > 16 || d->N % 4 != 0 || d->Cin % 64 != 0) return false;
  if (d->stride != 1 || d->up != 0 || d->vhalo != 0 || d->flags != 0) return false;
  if (d->rowbias || d->residual || d->gate || d->out_scale != 0.f || d->gn_part_out || d->splitk > 1) return false;
  if ((((long)d->N * d->K * 2 + 255) & ~255L) + (long)kSnPix * kSnStride > 64 * 1024 || d->ldc % 4 != 0) return false;      // weights + one staged chunk within 64 KB of LDS
  if ((long)d->B * d->Hin * d->Win * d->Cin >= 2147483647L) return false;          // 32-bit source offsets
  return true;
}

bool conv_small_cin_serves(const mx_gemm_desc* d) {
  static const bool off = [] { const char* e = getenv("MX_CONV_SMALL_CIN"); return e && e[0] == '0'; }();     // A/B: the tile kernels over the padded input
  if (off || d->n_segs != 0 || d->cin_valid <= 0 || d->cin_valid > 8 || d->Cin % 8 != 0 || d->N % 80 != 0) return false;
  if (d->stride != 1 || d->up != 0 || d->vhalo != 0 || d->flags != 0) return false;
  if (d->rowbias || d->residual || d->gate || d->out_scale != 0.f || d->gn_part_out || d->splitk > 1 || d->ldc % 8 != 0) return false;
  if ((((long)d->N * kScWRow * 2 + 255) & ~255L) + 4L * 16 * kScStage > 64 * 1024) return false;      // repacked weights + output staging within 64 KB of LDS
  return true;
}

int launch_conv_small_cin(hipStream_t s, const GemmArgs& a) {
  const int tiles = a.B * ((a.Hin + 3) / 4) * ((a.Win + 15) / 16);
  const size_t lds = (((size_t)a.N * kScWRow * 2 + 255) & ~(size_t)255) + 4 * 16 * kScStage;
  const int per_cu = std::max(1, std::min(4, (int)((160 * 1024) / (lds + 256))));
  hipLaunchKernelGGL(conv3x3_small_cin_kernel, dim3(std::min(tiles, per_cu * cu_count())), dim3(256), lds, s, a);
  return 0;
}

int launch_conv_small_n(hipStream_t s, const GemmArgs& a) {
  const int tiles = a.B * ((a.Hin + 3) / 4) * ((a.Win + 15) / 16);
  const size_t lds = (((size_t)a.N * a.K * 2 + 255) & ~(size_t)255) + (size_t)kSnPix * kSnStride;
  const int per_cu = std::max(1, std::min(6, (int)((160 * 1024) / (lds + 256))));        // workgroups a CU holds (LDS)
  hipLaunchKernelGGL(conv3x3_small_n_kernel, dim3(std::min(tiles, per_cu * cu_count())), dim3(256), lds, s, a);
  return 0;
}

}  // namespace mx
