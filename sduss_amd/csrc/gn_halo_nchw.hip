// Drop-in for the reference's only native op, esymred_mp.groupnorm / mock_groupnorm
// (sduss/model_executor/modules/kernels/norm_silu_concat.cu, .cpp), on NCHW patch batches, for gfx950.
//
// Written from the op's semantics, not from the CUDA source: 64-wide waves, fp32 statistics, an
// out-of-place cross-patch merge (the reference merges in place and races, cu:377-384), no device
// synchronisation, and no zero pass over the padded output (torch::zeros in the reference).
//
//   moments_kernel      per (patch, group) mean / biased variance               (cu:41-81)
//   merge_kernel        mean of patch means, rsqrt(mean of patch variances+eps) (cu:361-386)
//   apply_gather_kernel y = x*(rstd*gamma) + (beta - rstd*gamma*mean) into the interior of the [H+2, W+2] plane, and the plane's
//                       1-pixel frame GATHERED from the neighbour patches' edge pixels (the reference scatters them from the
//                       sender, cu:164-241, 285-357; the cells and values are the same: round 4 -- the writer of a halo side is looked up
//                       in the INVERSE of padding_idx, built per block in LDS, and the sender's mean / rstd are applied, so no symmetry
//                       of the table and no same-latent assumption is needed).  Every block writes its
//                       whole padded plane and nothing else: no zero pass (torch::zeros, cpp:71,92), no cross-block writes, and the
//                       plane leaves as aligned 8-byte / 16-byte stores after a transpose-free staging in LDS (round 2: the round-1
//                       form moved 2 bytes per lane).  Moments read 16 bytes per lane.
#include "common.h"
#include "../../include/mxdenoise.h"

namespace mx {

template <typename T> __device__ __forceinline__ float ldf(const T* p, long i);
template <> __device__ __forceinline__ float ldf<float>(const float* p, long i) { return p[i]; }
template <> __device__ __forceinline__ float ldf<bf16_t>(const bf16_t* p, long i) { return bf16_to_f32(p[i]); }
template <> __device__ __forceinline__ float ldf<_Float16>(const _Float16* p, long i) { return (float)p[i]; }
template <typename T> __device__ __forceinline__ T cvt(float v);
template <> __device__ __forceinline__ float cvt<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t cvt<bf16_t>(float v) { return f32_to_bf16(v); }
template <> __device__ __forceinline__ _Float16 cvt<_Float16>(float v) { return (_Float16)v; }

template <typename T>
__global__ __launch_bounds__(256) void moments_kernel(const T* __restrict__ x, float* __restrict__ mean,
                                                      float* __restrict__ var, int count) {
  __shared__ double sh[8];
  const long base = (long)blockIdx.x * count;
  float s = 0.f, q = 0.f;
  constexpr int VE = 16 / (int)sizeof(T);                   // elements per 16-byte load
  if ((count % VE) == 0 && ((uintptr_t)(x + base) & 15) == 0) {
    const u32x4* xv = reinterpret_cast<const u32x4*>(x + base);
    for (int i = threadIdx.x; i < count / VE; i += 256) {
      const u32x4 w = xv[i];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if constexpr (sizeof(T) == 4) { const float v = __uint_as_float(w[e]); s += v; q += v * v; }
        else {
          const T lo = __builtin_bit_cast(T, (unsigned short)(w[e] & 0xffffu)), hi = __builtin_bit_cast(T, (unsigned short)(w[e] >> 16));
          const float a = ldf<T>(&lo, 0), b = ldf<T>(&hi, 0);
          s += a; q += a * a; s += b; q += b * b;
        }
      }
    }
  } else {
    for (int i = threadIdx.x; i < count; i += 256) {
      const float v = ldf<T>(x, base + i);
      s += v;
      q += v * v;
    }
  }
  double ds = (double)wave_sum(s), dq = (double)wave_sum(q);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) { sh[wave * 2] = ds; sh[wave * 2 + 1] = dq; }
  __syncthreads();
  if (threadIdx.x == 0) {
    ds = sh[0] + sh[2] + sh[4] + sh[6];
    dq = sh[1] + sh[3] + sh[5] + sh[7];
    const double m = ds / count;
    double v = dq / count - m * m;
    if (v < 0.0) v = 0.0;
    mean[blockIdx.x] = (float)m;
    var[blockIdx.x] = (float)v;
  }
}

__global__ void merge_kernel(const float* __restrict__ mean, const float* __restrict__ var, float* __restrict__ mean2,
                             float* __restrict__ rstd2, const int* __restrict__ latent_offset,
                             const int* __restrict__ patch_map, int N, int G, float eps) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= N * G) return;
  const int n = idx / G, g = idx - n * G;
  const int li = patch_map[n];  // 1-based latent index (unet.py:136,159)
  const int lo = latent_offset[li - 1], hi = latent_offset[li];
  float m = 0.f, v = 0.f;
  for (int p = lo; p < hi; ++p) { m += mean[p * G + g]; v += var[p * G + g]; }
  const float cnt = (float)(hi - lo);
  mean2[idx] = m / cnt;
  rstd2[idx] = rsqrtf(v / cnt + eps);
}

// PB consecutive (patch, channel) planes per block (PB = 1 for 32 x 32 planes, more for the small planes of the deeper levels):
// the [H+2, W+2] output planes are assembled in LDS and leave as ONE contiguous run of wide stores
template <typename T, bool AFFINE>
__global__ __launch_bounds__(256) void apply_gather_kernel(const T* __restrict__ x, T* __restrict__ y, const T* __restrict__ gamma,
                                                           const T* __restrict__ beta, const float* __restrict__ mean,
                                                           const float* __restrict__ rstd, const int* __restrict__ padding_idx, int C,
                                                           int H, int W, int cpg, int PB, long planes, int N) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const int HW = H * W, W2 = W + 2, total = (H + 2) * W2;
  T* tile = reinterpret_cast<T*>(lds_raw);                  // [PB][total]
  float* coef = reinterpret_cast<float*>(lds_raw + (((size_t)PB * total * sizeof(T)) + 15) / 16 * 16);   // [PB][2]
  int* inv = reinterpret_cast<int*>(coef + 2 * PB);          // [patches of this block][4]: who WRITES my top / left / bottom / right halo
  const long plane0 = (long)blockIdx.x * PB;
  const int npl = (int)((planes - plane0) < PB ? (planes - plane0) : PB);
  const int t = threadIdx.x;
  // The reference scatters: sender b writes into the halo of padding_idx[b][d] (cu:164-241).  This kernel gathers, so it needs the inverse:
  // the sender of receiver r's side d is the b with padding_idx[b][opposite(d)] == r.  For split_sample's tables that is padding_idx[r][d]
  // itself (symmetric adjacency); computing it makes the op equal to the reference's for ANY table in which a halo cell has one writer
  // (with several the reference races).  The planes of a block span at most PB / C + 2 patches; the whole table is N * 4 ints in L2.
  const int n_first = (int)(plane0 / C), n_last = (int)((plane0 + npl - 1) / C);
  for (int e = t; e < (n_last - n_first + 1) * 4; e += 256) inv[e] = -1;
  __syncthreads();
  for (int e = t; e < N * 4; e += 256) {
    const int nb = padding_idx[e];
    if (nb >= n_first && nb <= n_last) inv[(nb - n_first) * 4 + (((e & 3) + 2) & 3)] = e >> 2;
  }
  // (published by the barrier below / the one in front of the frame pass)
  if (AFFINE) {
    for (int pl = t; pl < npl; pl += 256) {
      const long plane = plane0 + pl;
      const int n = (int)(plane / C), c = (int)(plane - (long)n * C);
      const int G = C / cpg, g = c / cpg;
      const float r = rstd[n * G + g], m = mean[n * G + g];
      const float sc = r * (gamma ? ldf<T>(gamma, c) : 1.f);
      coef[2 * pl] = sc;
      coef[2 * pl + 1] = (beta ? ldf<T>(beta, c) : 0.f) - sc * m;
    }
    __syncthreads();
  }
  auto tr = [&](T v, int pl) __attribute__((always_inline)) -> T { return AFFINE ? cvt<T>(ldf<T>(&v, 0) * coef[2 * pl] + coef[2 * pl + 1]) : v; };
  // a halo cell carries the SENDER's normalisation (it is a copy of the sender's output pixel): the same numbers as the receiver's when both
  // lie in one latent, which split_sample guarantees; evaluated here for the sender so that tables linking two latents are served too
  auto tr_from = [&](T v, int nb, int c) __attribute__((always_inline)) -> T {
    if (!AFFINE) return v;
    const int G = C / cpg, g = c / cpg;
    const float sc = rstd[nb * G + g] * (gamma ? ldf<T>(gamma, c) : 1.f);
    return cvt<T>(ldf<T>(&v, 0) * sc + ((beta ? ldf<T>(beta, c) : 0.f) - sc * mean[nb * G + g]));
  };
  // ---- interiors: the npl input planes are one contiguous run ----
  const T* xp = x + plane0 * HW;
  constexpr int VE = 16 / (int)sizeof(T);
  if ((W % VE) == 0 && ((uintptr_t)xp & 15) == 0) {
    for (int i = t; i < npl * HW / VE; i += 256) {
      const u32x4 w = reinterpret_cast<const u32x4*>(xp)[i];
      const int e0 = i * VE;
      const int pl = e0 / HW, r0 = e0 - pl * HW;
      const int row = r0 / W, col = r0 - row * W;
      T* dst = tile + pl * total + (row + 1) * W2 + col + 1;
      T v[VE];
      __builtin_memcpy(v, &w, 16);
#pragma unroll
      for (int e = 0; e < VE; ++e) dst[e] = tr(v[e], pl);
    }
  } else {
    for (int i = t; i < npl * HW; i += 256) {
      const int pl = i / HW, r0 = i - pl * HW;
      const int row = r0 / W, col = r0 - row * W;
      tile[pl * total + (row + 1) * W2 + col + 1] = tr(xp[i], pl);
    }
  }
  // ---- frames: rows from the top / bottom neighbours, columns and corners from the left / right neighbours (cu:186-241) ----
  __syncthreads();                                           // inv complete (also when !AFFINE, which has no barrier above)
  const T zero = cvt<T>(0.f);
  const int per = 2 * W2 + 2 * H;
  for (int kk = t; kk < npl * per; kk += 256) {
    const int pl = kk / per, k = kk - pl * per;
    const long plane = plane0 + pl;
    const int n = (int)(plane / C), c = (int)(plane - (long)n * C);
    int row, col;
    if (k < W2) { row = 0; col = k; }
    else if (k < 2 * W2) { row = H + 1; col = k - W2; }
    else { const int j = k - 2 * W2; row = 1 + (j >> 1); col = (j & 1) ? W + 1 : 0; }
    T v = zero;
    if (col == 0 || col == W + 1) {
      const int nb = inv[(n - n_first) * 4 + (col == 0 ? 1 : 3)];   // written by the patch on my left (its last column) / right (its first);
      if (nb != -1) {                                          // corners replicate that neighbour's corner pixel (cu:210-221, 228-239)
        const int sr = row == 0 ? 0 : row == H + 1 ? H - 1 : row - 1;
        v = tr_from(x[((long)nb * C + c) * HW + (long)sr * W + (col == 0 ? W - 1 : 0)], nb, c);
      }
    } else {
      const int nb = inv[(n - n_first) * 4 + (row == 0 ? 0 : 2)];   // written by the patch above (its last row) / below (its first row), columns 1..W
      if (nb != -1) v = tr_from(x[((long)nb * C + c) * HW + (long)(row == 0 ? H - 1 : 0) * W + (col - 1)], nb, c);
    }
    tile[pl * total + row * W2 + col] = v;
  }
  __syncthreads();
  // ---- the padded planes leave as one contiguous run ----
  T* yp = y + plane0 * total;
  const long bytes = (long)npl * total * (long)sizeof(T);
  if ((bytes & 15) == 0 && ((uintptr_t)yp & 15) == 0) {
    for (int i = t; i < bytes / 16; i += 256) reinterpret_cast<u32x4*>(yp)[i] = reinterpret_cast<const u32x4*>(tile)[i];
  } else if ((bytes & 7) == 0 && ((uintptr_t)yp & 7) == 0) {
    for (int i = t; i < bytes / 8; i += 256) reinterpret_cast<u32x2*>(yp)[i] = reinterpret_cast<const u32x2*>(tile)[i];
  } else {
    for (int i = t; i < npl * total; i += 256) yp[i] = tile[i];
  }
}

// !PAD: plain y = x * scale + shift per plane, 16 bytes per lane
template <typename T>
__global__ __launch_bounds__(256) void apply_plain_kernel(const T* __restrict__ x, T* __restrict__ y, const T* __restrict__ gamma,
                                                          const T* __restrict__ beta, const float* __restrict__ mean,
                                                          const float* __restrict__ rstd, int C, int HW, int cpg) {
  const int plane = blockIdx.x;
  const int n = plane / C, c = plane - n * C;
  const int G = C / cpg, g = c / cpg;
  const float r = rstd[n * G + g], m = mean[n * G + g];
  const float scale = r * (gamma ? ldf<T>(gamma, c) : 1.f);
  const float shift = (beta ? ldf<T>(beta, c) : 0.f) - scale * m;
  const T* xp = x + (long)plane * HW;
  T* yp = y + (long)plane * HW;
  constexpr int VE = 16 / (int)sizeof(T);
  if ((HW % VE) == 0 && (((uintptr_t)xp | (uintptr_t)yp) & 15) == 0) {
    for (int i = threadIdx.x; i < HW / VE; i += 256) {
      const u32x4 w = reinterpret_cast<const u32x4*>(xp)[i];
      T v[VE];
      __builtin_memcpy(v, &w, 16);
#pragma unroll
      for (int e = 0; e < VE; ++e) v[e] = cvt<T>(ldf<T>(&v[e], 0) * scale + shift);
      u32x4 o;
      __builtin_memcpy(&o, v, 16);
      reinterpret_cast<u32x4*>(yp)[i] = o;
    }
  } else {
    for (int i = threadIdx.x; i < HW; i += 256) yp[i] = cvt<T>(ldf<T>(xp, i) * scale + shift);
  }
}

template <typename T>
static int run(hipStream_t s, const void* x, const void* gamma, const void* beta, void* y, int N, int C, int H, int W,
               int cpg, float eps, bool affine, bool pad, const int* latent_offset, const int* patch_map,
               const int* padding_idx, float* ws) {
  const int G = affine ? C / cpg : 0;
  float* mean = ws; float* var = ws + (size_t)N * G; float* mean2 = var + (size_t)N * G; float* rstd2 = mean2 + (size_t)N * G;
  if (affine) {
    hipLaunchKernelGGL((moments_kernel<T>), dim3(N * G), dim3(256), 0, s, (const T*)x, mean, var, cpg * H * W);
    MX_LAUNCH_CHECK();
    hipLaunchKernelGGL(merge_kernel, dim3(cdiv(N * G, 256)), dim3(256), 0, s, (const float*)mean, (const float*)var,
                       mean2, rstd2, latent_offset, patch_map, N, G, eps);
    MX_LAUNCH_CHECK();
  }
  dim3 grid(N * C), block(256);
  if (pad) {
    const int total = (H + 2) * (W + 2);
    int PB = 2048 / (H * W);                   // ~2k input elements per block
    if (PB < 1) PB = 1;
    if (PB > 32) PB = 32;
    const size_t lds = (((size_t)PB * total * sizeof(T)) + 15) / 16 * 16 + (size_t)PB * 2 * sizeof(float) + (size_t)(PB / C + 2) * 4 * sizeof(int);
    MX_CHECK(lds <= 64 * 1024, "groupnorm_halo: patch plane too large for the LDS staging (H, W <= 126)");
    const long planes = (long)N * C;
    dim3 pgrid((unsigned)cdiv64(planes, PB));
    if (affine) hipLaunchKernelGGL((apply_gather_kernel<T, true>), pgrid, block, lds, s, (const T*)x, (T*)y, (const T*)gamma, (const T*)beta, (const float*)mean2, (const float*)rstd2, padding_idx, C, H, W, cpg, PB, planes, N);
    else hipLaunchKernelGGL((apply_gather_kernel<T, false>), pgrid, block, lds, s, (const T*)x, (T*)y, (const T*)nullptr, (const T*)nullptr, (const float*)nullptr, (const float*)nullptr, padding_idx, C, H, W, 1, PB, planes, N);
  } else {
    hipLaunchKernelGGL((apply_plain_kernel<T>), grid, block, 0, s, (const T*)x, (T*)y, (const T*)gamma, (const T*)beta, (const float*)mean2, (const float*)rstd2, C, H * W, cpg);
  }
  MX_LAUNCH_CHECK();
  return 0;
}

}  // namespace mx

extern "C" size_t mx_groupnorm_halo_workspace_bytes(int N, int C, int cpg) {
  if (cpg <= 0) return 0;
  return (size_t)4 * N * (C / cpg) * sizeof(float) + 64;
}

extern "C" int mx_groupnorm_halo(void* stream, const void* x, const void* gamma, const void* beta, void* y,
                                 int N, int C, int H, int W, int cpg, double eps, int padding,
                                 const int32_t* latent_offset, int n_latents, const int32_t* patch_map,
                                 const int32_t* padding_idx, int dtype, void* workspace) {
  using namespace mx;
  MX_CHECK(x && y && workspace && latent_offset && patch_map, "groupnorm_halo: null operand");
  MX_CHECK(N > 0 && C > 0 && H > 1 && W > 1, "groupnorm_halo: bad shape");
  MX_CHECK(cpg > 0 && C % cpg == 0, "groupnorm_halo: C % channels_per_group != 0");
  MX_CHECK(n_latents > 0, "groupnorm_halo: n_latents must be > 0");
  MX_CHECK(!padding || padding_idx, "groupnorm_halo: padding needs padding_idx");
  hipStream_t s = (hipStream_t)stream;
  if (dtype == MX_F32) return run<float>(s, x, gamma, beta, y, N, C, H, W, cpg, (float)eps, true, padding != 0, latent_offset, patch_map, padding_idx, (float*)workspace);
  if (dtype == MX_F16) return run<_Float16>(s, x, gamma, beta, y, N, C, H, W, cpg, (float)eps, true, padding != 0, latent_offset, patch_map, padding_idx, (float*)workspace);
  if (dtype == MX_BF16) return run<bf16_t>(s, x, gamma, beta, y, N, C, H, W, cpg, (float)eps, true, padding != 0, latent_offset, patch_map, padding_idx, (float*)workspace);
  MX_CHECK(false, "groupnorm_halo: bad dtype");
}

extern "C" int mx_halo_only(void* stream, const void* x, void* y, int N, int C, int H, int W,
                            const int32_t* padding_idx, int dtype) {
  using namespace mx;
  MX_CHECK(x && y && padding_idx, "halo_only: null operand");
  MX_CHECK(N > 0 && C > 0 && H > 1 && W > 1, "halo_only: bad shape");
  hipStream_t s = (hipStream_t)stream;
  if (dtype == MX_F32) return run<float>(s, x, nullptr, nullptr, y, N, C, H, W, 1, 0.f, false, true, nullptr, nullptr, padding_idx, nullptr);
  if (dtype == MX_F16) return run<_Float16>(s, x, nullptr, nullptr, y, N, C, H, W, 1, 0.f, false, true, nullptr, nullptr, padding_idx, nullptr);
  if (dtype == MX_BF16) return run<bf16_t>(s, x, nullptr, nullptr, y, N, C, H, W, 1, 0.f, false, true, nullptr, nullptr, padding_idx, nullptr);
  MX_CHECK(false, "halo_only: bad dtype");
}
