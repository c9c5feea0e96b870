// Small HBM-bound glue kernels of the step plan (gfx950): layout conversion at the model boundary,
// sinusoidal embeddings, channel concat, and the fused scheduler / CFG steps either side of the UNet.
#include <algorithm>

#include "common.h"
#include "../../include/mxdenoise.h"

// The scheduler kernels reproduce torch's op-by-op IEEE evaluation bit for bit: no mul+add contraction into FMA in this
// translation unit (HIP's __fmul_rn/__fadd_rn are plain operators in inlined headers and do get contracted; so plain operators here, plus
// -ffp-contract=off for this file in the Makefile).
#pragma clang fp contract(off)

namespace mx {

template <typename T> __device__ __forceinline__ float load_as_f32(const T* p, long i);
template <> __device__ __forceinline__ float load_as_f32<float>(const float* p, long i) { return p[i]; }
template <> __device__ __forceinline__ float load_as_f32<bf16_t>(const bf16_t* p, long i) { return bf16_to_f32(p[i]); }
template <> __device__ __forceinline__ float load_as_f32<_Float16>(const _Float16* p, long i) { return (float)p[i]; }
template <typename T> __device__ __forceinline__ void store_from_f32(T* p, long i, float v);
template <> __device__ __forceinline__ void store_from_f32<float>(float* p, long i, float v) { p[i] = v; }
template <> __device__ __forceinline__ void store_from_f32<bf16_t>(bf16_t* p, long i, float v) { p[i] = f32_to_bf16(v); }
template <> __device__ __forceinline__ void store_from_f32<_Float16>(_Float16* p, long i, float v) { p[i] = (_Float16)v; }

// value after a round trip through T (emulates an op whose result tensor has dtype T)
template <typename T> __device__ __forceinline__ float rnd(float v) {
  T tmp;
  store_from_f32<T>(&tmp, 0, v);
  return load_as_f32<T>(&tmp, 0);
}

// NCHW latents of any io dtype -> NHWC bf16 with the channel dim zero-padded to CP (=64), so conv_in
// runs on the same implicit-GEMM kernel as every other conv (unet.py:344).
template <typename T>
__global__ void prep_latent_kernel(const T* __restrict__ in, bf16_t* __restrict__ out, int B, int Cin, int HW, int CP) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;  // (pixel, 8-channel chunk)
  const int chunks = CP / 8;
  const long total = (long)B * HW * chunks;
  if (idx >= total) return;
  const int ch = (int)(idx % chunks);
  const long pix = idx / chunks;
  const int b = (int)(pix / HW);
  const int p = (int)(pix - (long)b * HW);
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = ch * 8 + e;
    v[e] = (c < Cin) ? load_as_f32<T>(in, ((long)b * Cin + c) * HW + p) : 0.f;
  }
  u32x4 o = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
  *reinterpret_cast<u32x4*>(out + idx * 8) = o;
}

// NHWC bf16 [B*HW, ld] (first C columns) -> NCHW io dtype
template <typename T>
__global__ void nhwc_to_nchw_kernel(const bf16_t* __restrict__ in, T* __restrict__ out, int B, int C, int HW, int ld) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;  // over B*C*HW, p fastest
  const long total = (long)B * C * HW;
  if (idx >= total) return;
  const int p = (int)(idx % HW);
  const long bc = idx / HW;
  const int c = (int)(bc % C);
  const int b = (int)(bc / C);
  store_from_f32<T>(out, idx, bf16_to_f32(in[((long)b * HW + p) * ld + c]));
}

// diffusers Timesteps(dim, flip_sin_to_cos=True, downscale_freq_shift=0): [cos | sin], bf16 out.
// tsin [B, d0] from timesteps; addin [B, text_dim + 6*da] = [text_embeds | sinusoid(time_ids)] (unet.py:314-334).
__global__ void time_embed_kernel(const float* __restrict__ timesteps, const bf16_t* __restrict__ text_embeds,
                                  const float* __restrict__ time_ids, bf16_t* __restrict__ tsin,
                                  bf16_t* __restrict__ addin, int B, int d0, int text_dim, int da) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int addw = text_dim + 6 * da;
  if (i < d0) {
    const int half = d0 / 2;
    const int j = (i < half) ? i : i - half;
    const float freq = __expf(-9.210340371976184f * (float)j / (float)half);
    const float ang = timesteps[b] * freq;
    tsin[(long)b * d0 + i] = f32_to_bf16((i < half) ? cosf(ang) : sinf(ang));
  }
  if (i < addw) {
    bf16_t o;
    if (i < text_dim) {
      o = text_embeds[(long)b * text_dim + i];
    } else {
      const int k = i - text_dim;
      const int which = k / da;
      const int e = k - which * da;
      const int half = da / 2;
      const int j = (e < half) ? e : e - half;
      const float freq = __expf(-9.210340371976184f * (float)j / (float)half);
      const float ang = time_ids[b * 6 + which] * freq;
      o = f32_to_bf16((e < half) ? cosf(ang) : sinf(ang));
    }
    addin[(long)b * addw + i] = o;
  }
}

// out[m, 0:C1] = a[m, :], out[m, C1:C1+C2] = b[m, :]   (torch.cat([hidden, skip], dim=1) in NHWC)
__global__ void concat_channels_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b,
                                       bf16_t* __restrict__ out, long M, int C1, int C2) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int chunks = (C1 + C2) / 8;
  if (idx >= M * chunks) return;
  const long m = idx / chunks;
  const int ch = (int)(idx - m * chunks);
  const int c1 = C1 / 8;
  const u32x4 v = (ch < c1) ? *reinterpret_cast<const u32x4*>(a + m * C1 + ch * 8)
                            : *reinterpret_cast<const u32x4*>(b + m * C2 + (ch - c1) * 8);
  *reinterpret_cast<u32x4*>(out + idx * 8) = v;
}

// batch_scale_model_input with the CFG duplication fused (scheduling_euler_discrete.py:161-184)
template <typename T>
__global__ void euler_scale_input_kernel(const T* __restrict__ lat, T* __restrict__ out, const float* __restrict__ sigma,
                                         int n_lat, int n_rows, long elems) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)n_rows * elems) return;
  const int row = (int)(idx / elems);
  const long e = idx - (long)row * elems;
  const int src = row % n_lat;
  // the reference evaluates samples / ((sigmas ** 2 + 1) ** 0.5) with sigmas cast to the tensor dtype
  // (scheduling_euler_discrete.py:175-182): one rounding to T per op
  const float s = rnd<T>(sigma[src]);
  const float denom = rnd<T>(__fsqrt_rn(rnd<T>(rnd<T>(s * s) + 1.0f)));   // correctly rounded sqrt, as torch.pow(x, 0.5) -> sqrt
  store_from_f32<T>(out, idx, load_as_f32<T>(lat, (long)src * elems + e) / denom);
}

// CFG combine + epsilon Euler step, fp32 math (pipeline_..._esymred.py:382-385; scheduling_euler_discrete.py:210-268)
template <typename T>
__global__ void cfg_euler_step_kernel(const T* __restrict__ noise, T* __restrict__ lat, const float* __restrict__ sigma,
                                      const float* __restrict__ sigma_next, float g, int n_lat, long elems) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)n_lat * elems) return;
  const int row = (int)(idx / elems);
  float eps;
  if (g > 0.f) {
    const float u = load_as_f32<T>(noise, idx);
    const float t = load_as_f32<T>(noise, (long)n_lat * elems + idx);
    // the combine runs in the model dtype (pipeline_..._esymred.py:383-385): one rounding per op
    eps = rnd<T>(u + rnd<T>(g * rnd<T>(t - u)));
  } else {
    eps = load_as_f32<T>(noise, idx);
  }
  const float x = load_as_f32<T>(lat, idx);
  const float s = sigma[row], sn = sigma_next[row];
  // separate IEEE ops (no fma contraction) so the fp32 chain matches torch's op-by-op evaluation
  const float pred_x0 = (x - (s * eps));
  const float d = ((x - pred_x0) / s);
  const float step = d * (sn - s);
  store_from_f32<T>(lat, idx, x + step);
}

}  // namespace mx

using namespace mx;

// ---- internal launchers used by the step plan (C++ linkage inside the library) ----
namespace mx {
int launch_prep_latent(hipStream_t s, const void* in, int dtype, void* out, int B, int Cin, int HW, int CP) {
  MX_CHECK(CP % 8 == 0 && Cin <= CP, "prep_latent: bad channel padding");
  const long total = (long)B * HW * (CP / 8);
  dim3 grid((unsigned)cdiv64(total, 256)), block(256);
  if (dtype == MX_F32) hipLaunchKernelGGL((prep_latent_kernel<float>), grid, block, 0, s, (const float*)in, (bf16_t*)out, B, Cin, HW, CP);
  else if (dtype == MX_F16) hipLaunchKernelGGL((prep_latent_kernel<_Float16>), grid, block, 0, s, (const _Float16*)in, (bf16_t*)out, B, Cin, HW, CP);
  else if (dtype == MX_BF16) hipLaunchKernelGGL((prep_latent_kernel<bf16_t>), grid, block, 0, s, (const bf16_t*)in, (bf16_t*)out, B, Cin, HW, CP);
  else MX_CHECK(false, "prep_latent: bad dtype");
  MX_LAUNCH_CHECK();
  return 0;
}
int launch_nhwc_to_nchw(hipStream_t s, const void* in, void* out, int dtype, int B, int C, int HW, int ld) {
  const long total = (long)B * C * HW;
  dim3 grid((unsigned)cdiv64(total, 256)), block(256);
  if (dtype == MX_F32) hipLaunchKernelGGL((nhwc_to_nchw_kernel<float>), grid, block, 0, s, (const bf16_t*)in, (float*)out, B, C, HW, ld);
  else if (dtype == MX_F16) hipLaunchKernelGGL((nhwc_to_nchw_kernel<_Float16>), grid, block, 0, s, (const bf16_t*)in, (_Float16*)out, B, C, HW, ld);
  else if (dtype == MX_BF16) hipLaunchKernelGGL((nhwc_to_nchw_kernel<bf16_t>), grid, block, 0, s, (const bf16_t*)in, (bf16_t*)out, B, C, HW, ld);
  else MX_CHECK(false, "nhwc_to_nchw: bad dtype");
  MX_LAUNCH_CHECK();
  return 0;
}
int launch_time_embed(hipStream_t s, const float* timesteps, const void* text_embeds, const float* time_ids,
                      void* tsin, void* addin, int B, int d0, int text_dim, int da) {
  const int w = (d0 > text_dim + 6 * da) ? d0 : text_dim + 6 * da;
  dim3 grid(cdiv(w, 256), B), block(256);
  hipLaunchKernelGGL(time_embed_kernel, grid, block, 0, s, timesteps, (const bf16_t*)text_embeds, time_ids,
                     (bf16_t*)tsin, (bf16_t*)addin, B, d0, text_dim, da);
  MX_LAUNCH_CHECK();
  return 0;
}
int launch_concat(hipStream_t s, const void* a, const void* b, void* out, long M, int C1, int C2) {
  MX_CHECK(C1 % 8 == 0 && C2 % 8 == 0, "concat: channels must be multiples of 8");
  const long total = M * ((C1 + C2) / 8);
  dim3 grid((unsigned)cdiv64(total, 256)), block(256);
  hipLaunchKernelGGL(concat_channels_kernel, grid, block, 0, s, (const bf16_t*)a, (const bf16_t*)b, (bf16_t*)out, M, C1, C2);
  MX_LAUNCH_CHECK();
  return 0;
}
}  // namespace mx

extern "C" int mx_euler_scale_input(void* stream, const void* latents, void* out, const float* sigma,
                                    int n_lat, int n_rows, int64_t elems, int dtype) {
  MX_CHECK(latents && out && sigma && n_lat > 0 && n_rows >= n_lat && elems > 0, "euler_scale_input: bad arguments");
  const long total = (long)n_rows * elems;
  dim3 grid((unsigned)cdiv64(total, 256)), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == MX_F32) hipLaunchKernelGGL((euler_scale_input_kernel<float>), grid, block, 0, s, (const float*)latents, (float*)out, sigma, n_lat, n_rows, (long)elems);
  else if (dtype == MX_F16) hipLaunchKernelGGL((euler_scale_input_kernel<_Float16>), grid, block, 0, s, (const _Float16*)latents, (_Float16*)out, sigma, n_lat, n_rows, (long)elems);
  else if (dtype == MX_BF16) hipLaunchKernelGGL((euler_scale_input_kernel<bf16_t>), grid, block, 0, s, (const bf16_t*)latents, (bf16_t*)out, sigma, n_lat, n_rows, (long)elems);
  else MX_CHECK(false, "euler_scale_input: bad dtype");
  MX_LAUNCH_CHECK();
  return 0;
}

extern "C" int mx_cfg_euler_step(void* stream, const void* noise, void* latents, const float* sigma,
                                 const float* sigma_next, float guidance_scale, int n_lat, int64_t elems, int dtype) {
  MX_CHECK(noise && latents && sigma && sigma_next && n_lat > 0 && elems > 0, "cfg_euler_step: bad arguments");
  const long total = (long)n_lat * elems;
  dim3 grid((unsigned)cdiv64(total, 256)), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == MX_F32) hipLaunchKernelGGL((cfg_euler_step_kernel<float>), grid, block, 0, s, (const float*)noise, (float*)latents, sigma, sigma_next, guidance_scale, n_lat, (long)elems);
  else if (dtype == MX_F16) hipLaunchKernelGGL((cfg_euler_step_kernel<_Float16>), grid, block, 0, s, (const _Float16*)noise, (_Float16*)latents, sigma, sigma_next, guidance_scale, n_lat, (long)elems);
  else if (dtype == MX_BF16) hipLaunchKernelGGL((cfg_euler_step_kernel<bf16_t>), grid, block, 0, s, (const bf16_t*)noise, (bf16_t*)latents, sigma, sigma_next, guidance_scale, n_lat, (long)elems);
  else MX_CHECK(false, "cfg_euler_step: bad dtype");
  MX_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------
// MMDiT edge kernels (SD3Transformer.py:82-83, 250-259; scheduling_flow_match_euler_discrete.py:186-196)
// ------------------------------------------------------------------------------------------
namespace mx {

// PatchEmbed im2col: latents NCHW -> tokens [B*h*w, ps*ps*C] bf16, k = (dy*ps + dx)*C + c
template <typename T>
__global__ void patchify_kernel(const T* __restrict__ in, bf16_t* __restrict__ out, int B, int Cc, int H, int W, int ps) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int K = ps * ps * Cc;
  const int h = H / ps, w = W / ps;
  if (idx >= (long)B * h * w * K) return;
  const int k = (int)(idx % K);
  const long tok = idx / K;
  const int c = k % Cc;
  const int dd = k / Cc;
  const int dy = dd / ps, dx = dd % ps;
  const int x = (int)(tok % w);
  const int y = (int)((tok / w) % h);
  const int b = (int)(tok / ((long)w * h));
  out[idx] = f32_to_bf16(load_as_f32<T>(in, (((long)b * Cc + c) * H + y * ps + dy) * W + x * ps + dx));
}

// tokens [B*h*w, ld] (first ps*ps*C columns, (p, q, c) order) -> NCHW [B, C, h*ps, w*ps]   ("nhwpqc->nchpwq")
template <typename T>
__global__ void unpatchify_kernel(const bf16_t* __restrict__ in, T* __restrict__ out, int B, int Cc, int H, int W, int ps, int ld) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;   // over B*C*H*W
  if (idx >= (long)B * Cc * H * W) return;
  const int X = (int)(idx % W);
  const int Y = (int)((idx / W) % H);
  const int c = (int)((idx / ((long)W * H)) % Cc);
  const int b = (int)(idx / ((long)W * H * Cc));
  const int w = W / ps, h = H / ps;
  const long tok = ((long)b * h + Y / ps) * w + X / ps;
  const int k = ((Y % ps) * ps + (X % ps)) * Cc + c;
  store_from_f32<T>(out, idx, bf16_to_f32(in[tok * ld + k]));
}

// rows (top + y, left + x) of a [m, m, d] table -> [h*w, d]   (PatchEmbed.cropped_pos_embed)
__global__ void crop_pos_kernel(const bf16_t* __restrict__ table, bf16_t* __restrict__ out, int m, int h, int w, int d) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;   // 16-byte chunks
  const int chunks = d / 8;
  if (idx >= (long)h * w * chunks) return;
  const int ch = (int)(idx % chunks);
  const long tok = idx / chunks;
  const int x = (int)(tok % w), y = (int)(tok / w);
  const int top = (m - h) / 2, left = (m - w) / 2;
  const long src = ((long)(top + y) * m + left + x) * d + ch * 8;
  *reinterpret_cast<u32x4*>(out + idx * 8) = *reinterpret_cast<const u32x4*>(table + src);
}

// [cos | sin] sinusoid of one scalar per row (Timesteps(dim, flip_sin_to_cos=True, freq_shift 0)), bf16 out
__global__ void sinus_embed_kernel(const float* __restrict__ t, bf16_t* __restrict__ out, int dim) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= dim) return;
  const int half = dim / 2;
  const int j = (i < half) ? i : i - half;
  const float ang = t[b] * __expf(-9.210340371976184f * (float)j / (float)half);
  out[(long)b * dim + i] = f32_to_bf16((i < half) ? cosf(ang) : sinf(ang));
}

// CFG combine + flow-match Euler step: x <- x + (sigma_next - sigma) * v, fp32 math
template <typename T>
__global__ void cfg_flow_step_kernel(const T* __restrict__ noise, T* __restrict__ lat, const float* __restrict__ sigma,
                                     const float* __restrict__ sigma_next, float g, int n_lat, long elems) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)n_lat * elems) return;
  const int row = (int)(idx / elems);
  float v;
  if (g > 0.f) {
    const float u = load_as_f32<T>(noise, idx);
    const float t = load_as_f32<T>(noise, (long)n_lat * elems + idx);
    v = rnd<T>(u + rnd<T>(g * rnd<T>(t - u)));   // pipeline_stable_diffusion_3_esymred.py:365-367
  } else {
    v = load_as_f32<T>(noise, idx);
  }
  const float x = load_as_f32<T>(lat, idx);
  const float step = (sigma_next[row] - sigma[row]) * v;
  store_from_f32<T>(lat, idx, x + step);
}

int launch_patchify(hipStream_t s, const void* in, int dtype, void* out, int B, int C, int H, int W, int ps) {
  const long total = (long)B * H * W * C;
  dim3 grid((unsigned)cdiv64(total, 256)), block(256);
  if (dtype == MX_F32) hipLaunchKernelGGL((patchify_kernel<float>), grid, block, 0, s, (const float*)in, (bf16_t*)out, B, C, H, W, ps);
  else if (dtype == MX_F16) hipLaunchKernelGGL((patchify_kernel<_Float16>), grid, block, 0, s, (const _Float16*)in, (bf16_t*)out, B, C, H, W, ps);
  else if (dtype == MX_BF16) hipLaunchKernelGGL((patchify_kernel<bf16_t>), grid, block, 0, s, (const bf16_t*)in, (bf16_t*)out, B, C, H, W, ps);
  else MX_CHECK(false, "patchify: bad dtype");
  MX_LAUNCH_CHECK();
  return 0;
}
int launch_unpatchify(hipStream_t s, const void* in, void* out, int dtype, int B, int C, int H, int W, int ps, int ld) {
  const long total = (long)B * H * W * C;
  dim3 grid((unsigned)cdiv64(total, 256)), block(256);
  if (dtype == MX_F32) hipLaunchKernelGGL((unpatchify_kernel<float>), grid, block, 0, s, (const bf16_t*)in, (float*)out, B, C, H, W, ps, ld);
  else if (dtype == MX_F16) hipLaunchKernelGGL((unpatchify_kernel<_Float16>), grid, block, 0, s, (const bf16_t*)in, (_Float16*)out, B, C, H, W, ps, ld);
  else if (dtype == MX_BF16) hipLaunchKernelGGL((unpatchify_kernel<bf16_t>), grid, block, 0, s, (const bf16_t*)in, (bf16_t*)out, B, C, H, W, ps, ld);
  else MX_CHECK(false, "unpatchify: bad dtype");
  MX_LAUNCH_CHECK();
  return 0;
}
int launch_crop_pos(hipStream_t s, const void* table, void* out, int m, int h, int w, int d) {
  MX_CHECK(d % 8 == 0 && h <= m && w <= m, "crop_pos: bad geometry");
  const long total = (long)h * w * (d / 8);
  hipLaunchKernelGGL(crop_pos_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, s, (const bf16_t*)table, (bf16_t*)out, m, h, w, d);
  MX_LAUNCH_CHECK();
  return 0;
}
int launch_sinus_embed(hipStream_t s, const float* t, void* out, int B, int dim) {
  hipLaunchKernelGGL(sinus_embed_kernel, dim3(cdiv(dim, 256), B), dim3(256), 0, s, t, (bf16_t*)out, dim);
  MX_LAUNCH_CHECK();
  return 0;
}
}  // namespace mx

extern "C" int mx_cfg_flow_step(void* stream, const void* noise, void* latents, const float* sigma, const float* sigma_next,
                                float guidance_scale, int n_lat, int64_t elems, int dtype) {
  MX_CHECK(noise && latents && sigma && sigma_next && n_lat > 0 && elems > 0, "cfg_flow_step: bad arguments");
  const long total = (long)n_lat * elems;
  dim3 grid((unsigned)cdiv64(total, 256)), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == MX_F32) hipLaunchKernelGGL((cfg_flow_step_kernel<float>), grid, block, 0, s, (const float*)noise, (float*)latents, sigma, sigma_next, guidance_scale, n_lat, (long)elems);
  else if (dtype == MX_F16) hipLaunchKernelGGL((cfg_flow_step_kernel<_Float16>), grid, block, 0, s, (const _Float16*)noise, (_Float16*)latents, sigma, sigma_next, guidance_scale, n_lat, (long)elems);
  else if (dtype == MX_BF16) hipLaunchKernelGGL((cfg_flow_step_kernel<bf16_t>), grid, block, 0, s, (const bf16_t*)noise, (bf16_t*)latents, sigma, sigma_next, guidance_scale, n_lat, (long)elems);
  else MX_CHECK(false, "cfg_flow_step: bad dtype");
  MX_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------
// Row softmax in place on a bf16 [rows, L] score matrix (fp32 math): the VAE decoder's single-head, 512-wide mid-block
// attention (diffusers Attention with heads = 1; AutoencoderKL decoder) is computed as GEMM -> softmax -> GEMM.
// ------------------------------------------------------------------------------------------
namespace mx {
__global__ __launch_bounds__(256) void softmax_rows_kernel(bf16_t* __restrict__ s, int L, long ld) {
  __shared__ float red[8];
  bf16_t* row = s + (long)blockIdx.x * ld;
  const int t = threadIdx.x;
  float mx_ = -INFINITY;
  for (int i = t; i < L / 8; i += 256) {
    const u32x4 v = reinterpret_cast<const u32x4*>(row)[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) mx_ = fmaxf(mx_, fmaxf(bf16lo_to_f32(v[e]), bf16hi_to_f32(v[e])));
  }
  mx_ = wave_max(mx_);
  if ((t & 63) == 0) red[t >> 6] = mx_;
  __syncthreads();
  mx_ = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float sum = 0.f;
  for (int i = t; i < L / 8; i += 256) {
    const u32x4 v = reinterpret_cast<const u32x4*>(row)[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) sum += __expf(bf16lo_to_f32(v[e]) - mx_) + __expf(bf16hi_to_f32(v[e]) - mx_);
  }
  sum = wave_sum(sum);
  if ((t & 63) == 0) red[4 + (t >> 6)] = sum;
  __syncthreads();
  const float inv = 1.0f / (red[4] + red[5] + red[6] + red[7]);
  for (int i = t; i < L / 8; i += 256) {
    const u32x4 v = reinterpret_cast<const u32x4*>(row)[i];
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = pack_bf16x2(__expf(bf16lo_to_f32(v[e]) - mx_) * inv, __expf(bf16hi_to_f32(v[e]) - mx_) * inv);
    reinterpret_cast<u32x4*>(row)[i] = o;
  }
}
int launch_softmax_rows(hipStream_t s, void* scores, long rows, int L, long ld) {
  MX_CHECK(L % 8 == 0 && ld % 8 == 0 && rows > 0, "softmax_rows: L and ld must be multiples of 8");
  hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)rows), dim3(256), 0, s, (bf16_t*)scores, L, ld);
  MX_LAUNCH_CHECK();
  return 0;
}
}  // namespace mx

// ----------------------------------------------------------------------------------------------------------------------
// CLIP text encoder glue (clip_text.cpp): token + position embedding lookup, and the pooled row of each prompt
// ----------------------------------------------------------------------------------------------------------------------
namespace mx {
// out[b, t, :] = tok[ids[b, t], :] + pos[t, :]   (bf16 tables, fp32 add, one rounding; pos == nullptr: token embedding only, T5); ids are clamped to the vocabulary
__global__ void clip_embed_kernel(const int* __restrict__ ids, const bf16_t* __restrict__ tok, const bf16_t* __restrict__ pos,
                                  bf16_t* __restrict__ out, int rows, int L, int H, int vocab) {
  const int row = blockIdx.x;
  if (row >= rows) return;
  int id = ids[row];
  id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
  const bf16_t* tr = tok + (long)id * H;
  const bf16_t* pr = pos + (long)(row % L) * H;
  bf16_t* orow = out + (long)row * H;
  for (int c = threadIdx.x * 8; c < H; c += blockDim.x * 8) {
    const u32x4 a = *reinterpret_cast<const u32x4*>(tr + c), b = pos ? *reinterpret_cast<const u32x4*>(pr + c) : u32x4{0u, 0u, 0u, 0u};
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = pack_bf16x2(bf16lo_to_f32(a[e]) + bf16lo_to_f32(b[e]), bf16hi_to_f32(a[e]) + bf16hi_to_f32(b[e]));
    *reinterpret_cast<u32x4*>(orow + c) = o;
  }
}
// pooled[b, :] = x[b, eos(b), :], eos(b) = first position holding eos_id, or (eos_id < 0: the legacy CLIP configs with eos_token_id == 2)
// the position of the largest id (transformers CLIPTextTransformer.forward)
__global__ void clip_pool_kernel(const int* __restrict__ ids, const bf16_t* __restrict__ x, bf16_t* __restrict__ out, int L, int H, int eos_id) {
  const int b = blockIdx.x;
  __shared__ int pos_s;
  if (threadIdx.x == 0) {
    int best = 0;
    if (eos_id >= 0) {
      best = L - 1;                                      // (argmax of an all-false mask is 0 in the reference; a prompt always has its EOS)
      for (int t = 0; t < L; ++t) if (ids[b * L + t] == eos_id) { best = t; break; }
      bool any = false;
      for (int t = 0; t < L; ++t) any |= ids[b * L + t] == eos_id;
      if (!any) best = 0;
    } else {
      int mx_id = ids[b * L];
      for (int t = 1; t < L; ++t) if (ids[b * L + t] > mx_id) { mx_id = ids[b * L + t]; best = t; }
    }
    pos_s = best;
  }
  __syncthreads();
  const bf16_t* src = x + ((long)b * L + pos_s) * H;
  for (int c = threadIdx.x * 8; c < H; c += blockDim.x * 8) *reinterpret_cast<u32x4*>(out + (long)b * H + c) = *reinterpret_cast<const u32x4*>(src + c);
}
int launch_clip_embed(hipStream_t s, const int* ids, const bf16_t* tok, const bf16_t* pos, bf16_t* out, int rows, int L, int H, int vocab) {
  MX_CHECK(H % 8 == 0, "clip_embed: hidden size must be a multiple of 8");
  hipLaunchKernelGGL(clip_embed_kernel, dim3(rows), dim3(128), 0, s, ids, tok, pos, out, rows, L, H, vocab);
  MX_LAUNCH_CHECK();
  return 0;
}
int launch_clip_pool(hipStream_t s, const int* ids, const bf16_t* x, bf16_t* out, int B, int L, int H, int eos_id) {
  hipLaunchKernelGGL(clip_pool_kernel, dim3(B), dim3(128), 0, s, ids, x, out, L, H, eos_id);
  MX_LAUNCH_CHECK();
  return 0;
}
}  // namespace mx

// ----------------------------------------------------------------------------------------------------------------------
// Block-skip cache (unet_sdxl.cpp, mx_unet_forward_cached): per-sample squared difference of a block input against the input the block saw
// when it last ran -- the feature of the reference's CacheManager.get_mask (modules/cache_manager.py:105-159: mse_loss(...).mean(dim=(-1,-2,-3)))
// ----------------------------------------------------------------------------------------------------------------------
namespace mx {
constexpr int kMseChunks = 64;
// partial[b][chunk] = sum over the chunk of (a - b)^2 (fp32 in-thread, fp64 across the block); the host adds the 64 partials of a sample: the
// result does not depend on launch order
// slot (optional): sample i of `b` lives at row slot[i] (the block-skip cache keeps one row per request, not per batch position)
__global__ __launch_bounds__(256) void sq_diff_partial_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b, long elems, double* __restrict__ partial,
                                                              const int* __restrict__ slot) {
  const int smp = blockIdx.y, chunk = blockIdx.x;
  const long per = (elems / 8 + kMseChunks - 1) / kMseChunks;           // 16-byte vectors per chunk
  const long v0 = (long)chunk * per, v1 = min(v0 + per, elems / 8);
  const bf16_t* pa = a + (long)smp * elems; const bf16_t* pb = b + (long)(slot ? slot[smp] : smp) * elems;
  float acc = 0.f;
  for (long v = v0 + threadIdx.x; v < v1; v += 256) {
    const u32x4 x = *reinterpret_cast<const u32x4*>(pa + v * 8), y = *reinterpret_cast<const u32x4*>(pb + v * 8);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float d0 = bf16lo_to_f32(x[e]) - bf16lo_to_f32(y[e]), d1 = bf16hi_to_f32(x[e]) - bf16hi_to_f32(y[e]);
      acc += d0 * d0 + d1 * d1;
    }
  }
  __shared__ double red[4];
  double d = (double)acc;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) d += __shfl_xor(d, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = d;
  __syncthreads();
  if (threadIdx.x == 0) partial[(long)smp * kMseChunks + chunk] = red[0] + red[1] + red[2] + red[3];
}
int launch_sq_diff_partial(hipStream_t s, const void* a, const void* b, long elems_per_sample, int B, double* partial, const int* slot) {
  MX_CHECK(elems_per_sample % 8 == 0, "sq_diff: elements per sample must be a multiple of 8");
  hipLaunchKernelGGL(sq_diff_partial_kernel, dim3(kMseChunks, B), dim3(256), 0, s, (const bf16_t*)a, (const bf16_t*)b, elems_per_sample, partial, slot);
  MX_LAUNCH_CHECK();
  return 0;
}
// rows of `bytes` (a multiple of 16) between a batch-ordered tensor and a slot-ordered one: scatter: slotted[slot[i]] = batch[i]; gather: batch[i] =
// slotted[slot[i]]
__global__ __launch_bounds__(256) void copy_rows_kernel(char* __restrict__ batch, char* __restrict__ slotted, long vecs, const int* __restrict__ slot, int scatter) {
  const int smp = blockIdx.y;
  if (slot[smp] < 0) return;                   // a sample the copy does not concern (block cache: partial reuse inside a running block)
  u32x4* pb = reinterpret_cast<u32x4*>(batch) + (long)smp * vecs;
  u32x4* ps = reinterpret_cast<u32x4*>(slotted) + (long)slot[smp] * vecs;
  for (long v = (long)blockIdx.x * 256 + threadIdx.x; v < vecs; v += (long)gridDim.x * 256) {
    if (scatter) ps[v] = pb[v]; else pb[v] = ps[v];
  }
}
int launch_copy_rows(hipStream_t s, void* batch, void* slotted, size_t bytes_per_sample, int B, const int* slot, int scatter) {
  MX_CHECK(bytes_per_sample % 16 == 0 && slot != nullptr, "copy_rows: rows must be multiples of 16 bytes");
  const long vecs = (long)(bytes_per_sample / 16);
  const int gx = (int)std::min<long>((vecs + 255) / 256, 256);
  hipLaunchKernelGGL(copy_rows_kernel, dim3(gx, B), dim3(256), 0, s, (char*)batch, (char*)slotted, vecs, slot, scatter);
  MX_LAUNCH_CHECK();
  return 0;
}
}  // namespace mx
