// Drop-in for the reference's only native op, esymred_mp.groupnorm / mock_groupnorm
// (sduss/model_executor/modules/kernels/norm_silu_concat.cu, .cpp), on NCHW patch batches, for gfx950.
//
// Written from the op's semantics, not from the CUDA source: 64-wide waves, fp32 statistics, an
// out-of-place cross-patch merge (the reference merges in place and races, cu:377-384), no device
// synchronisation, and frame-only zeroing instead of a full torch::zeros pass over the padded output.
//
//   moments_kernel      per (patch, group) mean / biased variance               (cu:41-81)
//   merge_kernel        mean of patch means, rsqrt(mean of patch variances+eps) (cu:361-386)
//   zero_frame_kernel   zero the 1-pixel frame of every [H+2, W+2] plane        (torch::zeros, cpp:71,92)
//   apply_scatter_kernel y = x*(rstd*gamma) + (beta - rstd*gamma*mean) into the interior, and the sender-driven
//                       scatter of edge rows / columns / corners into the neighbours' frames (cu:164-241, 285-357)
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
  for (int i = threadIdx.x; i < count; i += 256) {
    const float v = ldf<T>(x, base + i);
    s += v;
    q += v * v;
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

template <typename T>
__global__ void zero_frame_kernel(T* __restrict__ y, long planes, int H, int W) {
  const int per = 2 * (W + 2) + 2 * H;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= planes * per) return;
  const long pl = idx / per;
  const int k = (int)(idx - pl * per);
  int row, col;
  if (k < W + 2) { row = 0; col = k; }
  else if (k < 2 * (W + 2)) { row = H + 1; col = k - (W + 2); }
  else { const int j = k - 2 * (W + 2); row = 1 + (j >> 1); col = (j & 1) ? W + 1 : 0; }
  y[(pl * (H + 2) + row) * (W + 2) + col] = cvt<T>(0.f);
}

// one block per (patch, channel) plane
template <typename T, bool AFFINE, bool PAD>
__global__ __launch_bounds__(256) void apply_scatter_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                            const T* __restrict__ gamma, const T* __restrict__ beta,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const int* __restrict__ padding_idx, int C, int H, int W,
                                                            int cpg) {
  const int plane = blockIdx.x;
  const int n = plane / C, c = plane - n * C;
  float scale = 1.f, shift = 0.f;
  if (AFFINE) {
    const int G = C / cpg;
    const int g = c / cpg;
    const float r = rstd[n * G + g], m = mean[n * G + g];
    const float ga = gamma ? ldf<T>(gamma, c) : 1.f;
    const float be = beta ? ldf<T>(beta, c) : 0.f;
    scale = r * ga;
    shift = be - scale * m;
  }
  int top = -1, left = -1, bottom = -1, right = -1;
  if (PAD) {
    top = padding_idx[n * 4]; left = padding_idx[n * 4 + 1];
    bottom = padding_idx[n * 4 + 2]; right = padding_idx[n * 4 + 3];
  }
  const int HW = H * W;
  const int W2 = W + 2;
  const long pstride = (long)(H + 2) * W2;
  const T* xp = x + (long)plane * HW;
  for (int i = threadIdx.x; i < HW; i += 256) {
    const int row = i / W, col = i - row * W;
    float v = ldf<T>(xp, i);
    if (AFFINE) v = v * scale + shift;
    const T o = cvt<T>(v);
    if (!PAD) {
      y[(long)plane * HW + i] = o;
      continue;
    }
    y[(long)plane * pstride + (long)(row + 1) * W2 + col + 1] = o;
    if (row == 0 && top != -1) y[((long)top * C + c) * pstride + (long)(H + 1) * W2 + col + 1] = o;
    if (row == H - 1 && bottom != -1) y[((long)bottom * C + c) * pstride + col + 1] = o;
    if (col == 0 && left != -1) {
      T* d = y + ((long)left * C + c) * pstride;
      d[(long)(row + 1) * W2 + W + 1] = o;
      if (row == 0) d[W + 1] = o;                            // corner replicated by the sender (cu:210-215)
      if (row == H - 1) d[(long)(H + 1) * W2 + W + 1] = o;   // (cu:216-221)
    }
    if (col == W - 1 && right != -1) {
      T* d = y + ((long)right * C + c) * pstride;
      d[(long)(row + 1) * W2] = o;
      if (row == 0) d[0] = o;                                // (cu:228-233)
      if (row == H - 1) d[(long)(H + 1) * W2] = o;           // (cu:234-239)
    }
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
  if (pad) {
    const long planes = (long)N * C;
    const long total = planes * (2 * (W + 2) + 2 * H);
    hipLaunchKernelGGL((zero_frame_kernel<T>), dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, s, (T*)y, planes, H, W);
    MX_LAUNCH_CHECK();
  }
  dim3 grid(N * C), block(256);
  if (affine && pad) hipLaunchKernelGGL((apply_scatter_kernel<T, true, true>), grid, block, 0, s, (const T*)x, (T*)y, (const T*)gamma, (const T*)beta, (const float*)mean2, (const float*)rstd2, padding_idx, C, H, W, cpg);
  else if (affine) hipLaunchKernelGGL((apply_scatter_kernel<T, true, false>), grid, block, 0, s, (const T*)x, (T*)y, (const T*)gamma, (const T*)beta, (const float*)mean2, (const float*)rstd2, padding_idx, C, H, W, cpg);
  else hipLaunchKernelGGL((apply_scatter_kernel<T, false, true>), grid, block, 0, s, (const T*)x, (T*)y, (const T*)nullptr, (const T*)nullptr, (const float*)nullptr, (const float*)nullptr, padding_idx, C, H, W, 1);
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
