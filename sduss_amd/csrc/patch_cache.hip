// Kernels of the block-skip cache at the reference's unit, the 256-px PATCH (mx_unet_forward_cached_mixed, include/mxdenoise.h;
// sduss/model_executor/modules/cache_manager.py:84-161 with is_sliced=True).  All HBM-bound data movement on NHWC / token-major bf16:
//
//   pc_image_copy    a sample's image between the batch-ordered activations and its row of a state tensor (one row per REQUEST, slot table),
//                    state -> batch optionally adding a per-sample channel vector (the time embedding, resnet.py:421-426) and a residual
//   pc_gather        the asking patches of a whole-image tensor -> a compact batch of patches, optionally with the reference's 1-pixel halo
//                    (get_adjacency / the fused GroupNorm's scatter: rows from the top / bottom neighbour, columns from the left / right one,
//                    corners = the left / right neighbour's own corner pixel, zeros at the image border: norm_silu_concat.cu:186-241) and
//                    optionally through a nearest-2x upsample (PatchUpsample2D interpolates, then exchanges halos: resnet.py:316-329)
//   pc_scatter       a compact batch of patches (cropped) -> the patches' places in the requests' state rows (update_and_return's
//                    `output[mask] = new_output`, cache_manager.py:95-97)
//   pc_patch_sq_diff per patch: sum of (x - cached x)^2 (the predictor's feature, cache_manager.py:112-123, 141-144)
//
// Geometry: samples are described at LEVEL 0 (latent resolution) and every kernel takes the level of its tensor: image h >> level, w >> level,
// first row row0 >> 2 level, patch edge p0 >> level (the patch grid of a sample is the same at every level).
#include "common.h"
#include "patch_cache.h"
#include "../../include/mxdenoise.h"
#include <algorithm>

namespace mx {

__global__ __launch_bounds__(256) void pc_image_copy_kernel(bf16_t* batch, bf16_t* __restrict__ state, const PcSample* __restrict__ samp, int level,
                                                            int C, long state_row_elems, int to_batch, const float* __restrict__ vec, int ldvec,
                                                            const bf16_t* residual, int gate) {
  const PcSample s = samp[blockIdx.y];
  const long n = ((long)(s.h >> level) * (s.w >> level) * C) >> 3;        // 16-byte vectors of the image
  bf16_t* pb = batch + (s.row0 >> (2 * level)) * C;
  bf16_t* ps = state + (long)s.slot * state_row_elems;
  const bf16_t* pr = residual ? residual + (s.row0 >> (2 * level)) * C : nullptr;
  const float* pv = vec ? vec + (long)blockIdx.y * ldvec : nullptr;
  for (long v = (long)blockIdx.x * 256 + threadIdx.x; v < n; v += (long)gridDim.x * 256) {
    if (!to_batch) { reinterpret_cast<u32x4*>(ps)[v] = reinterpret_cast<const u32x4*>(pb)[v]; continue; }
    u32x4 x = reinterpret_cast<const u32x4*>(ps)[v];
    if (pv || pr) {
      const int c0 = (int)((v << 3) % C);
      u32x4 r = pr ? reinterpret_cast<const u32x4*>(pr)[v] : u32x4{0u, 0u, 0u, 0u};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float lo = bf16lo_to_f32(x[e]), hi = bf16hi_to_f32(x[e]);
        if (pv && gate) { lo *= pv[c0 + 2 * e]; hi *= pv[c0 + 2 * e + 1]; }       // AdaLN-Zero gate (transformer.py:344-345): residual + gate * state
        else if (pv) { lo += pv[c0 + 2 * e]; hi += pv[c0 + 2 * e + 1]; }
        if (pr) { lo += bf16lo_to_f32(r[e]); hi += bf16hi_to_f32(r[e]); }
        x[e] = pack_bf16x2(lo, hi);
      }
    }
    reinterpret_cast<u32x4*>(pb)[v] = x;
  }
}

// state -> batch with a residual, TOKEN ROW by token row (round 5): one wave per row, so that the row statistics of the merged hidden state come with it --
// the slab form the folded LayerNorm of the 256 / 128-row GEMMs reads (stats1[m][4][2]: one slab (sum, sum of squares)) and / or the finalised form of the
// 256 x 256 kernel (fin[m] = (mean, rstd)) -- and the (x - mean) * rstd passes in front of attn2.to_q and the GEGLU projection disappear from the cached
// forward (2.2 ms of its 64 at 8 x 1024^2: profiles/r05_e_cache_breakdown.txt).  Statistics of the bf16 values as stored (what the pass read).
__global__ __launch_bounds__(256) void pc_rows_load_stats_kernel(bf16_t* batch, const bf16_t* __restrict__ state, const PcSample* __restrict__ samp, int level, int C,
                                                                 long state_row_elems, const bf16_t* residual, float* __restrict__ stats1, float* __restrict__ fin, float eps) {
  const PcSample s = samp[blockIdx.y];
  const int rows = (s.h >> level) * (s.w >> level);
  const long r0 = s.row0 >> (2 * level);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int cv = C >> 3;                       // 16-byte vectors per row
  for (int r = blockIdx.x * 4 + wave; r < rows; r += gridDim.x * 4) {
    const u32x4* ps = reinterpret_cast<const u32x4*>(state + (long)s.slot * state_row_elems + (long)r * C);
    const u32x4* pr = reinterpret_cast<const u32x4*>(residual + (r0 + r) * C);
    u32x4* pb = reinterpret_cast<u32x4*>(batch + (r0 + r) * C);
    float a = 0.f, q = 0.f;
    for (int v = lane; v < cv; v += 64) {
      u32x4 x = ps[v];
      const u32x4 rr = pr[v];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const unsigned w = pack_bf16x2(bf16lo_to_f32(x[e]) + bf16lo_to_f32(rr[e]), bf16hi_to_f32(x[e]) + bf16hi_to_f32(rr[e]));
        const float lo = bf16lo_to_f32(w), hi = bf16hi_to_f32(w);
        a += lo + hi; q += lo * lo + hi * hi;
        x[e] = w;
      }
      pb[v] = x;
    }
    a = wave_sum(a); q = wave_sum(q);
    if (lane == 0) {
      const long m = r0 + r;
      if (stats1) *reinterpret_cast<f32x2*>(stats1 + m * 8) = f32x2{a, q};
      if (fin) {
        const float mean = a / (float)C;
        const float var = fmaxf(q / (float)C - mean * mean, 0.f);
        *reinterpret_cast<f32x2*>(fin + m * 2) = f32x2{mean, rsqrtf(var + eps)};
      }
    }
  }
}

// one block per (output pixel row of the halo'd patch, compact patch): P = p + halo_lo + halo_hi pixels of C channels
__global__ __launch_bounds__(256) void pc_gather_kernel(const bf16_t* __restrict__ src, int ld_src, int C, bf16_t* __restrict__ dst, const PcPatch* __restrict__ list,
                                                        const PcSample* __restrict__ samp, int level, int p, int halo_lo, int halo_hi, int up) {
  const PcPatch q = list[blockIdx.y];
  const PcSample s = samp[q.b];
  const int P = p + halo_lo + halo_hi;
  const int yy = blockIdx.x;
  // image at the gather's resolution (the upsampled one when up): h, w; the source tensor holds it at (h >> up, w >> up)
  const int h = (s.h >> level) << up, w = (s.w >> level) << up;
  const int ws = w >> up;
  const bf16_t* img = src + (s.row0 >> (2 * level)) * ld_src;
  bf16_t* out = dst + ((long)blockIdx.y * P + yy) * (long)P * C;
  const int y = yy - halo_lo;                         // row inside the patch: -1 .. p
  const int cv = C >> 3;
  for (int i = threadIdx.x; i < P * cv; i += 256) {
    const int xx = i / cv, c8 = i - xx * cv;
    const int x = xx - halo_lo;
    const bool yin = y >= 0 && y < p, xin = x >= 0 && x < p;
    int gy = q.py * p + y, gx = q.px * p + x;
    bool ok = true;
    if (!xin) {                                        // column halo and corners: the left / right neighbour's pixel of MY row range (corner replication)
      ok = gx >= 0 && gx < w;
      if (!yin) gy = q.py * p + (y < 0 ? 0 : p - 1);
    } else if (!yin) {
      ok = gy >= 0 && gy < h;
    }
    if (y < -1 || x < -1) ok = false;                 // halo_lo == 2: an outer ring of zeros in front of the halo (stride-2 convs, see unet_sdxl.cpp pc_conv)
    u32x4 v = {0u, 0u, 0u, 0u};
    if (ok) v = *reinterpret_cast<const u32x4*>(img + ((long)(gy >> up) * ws + (gx >> up)) * ld_src + c8 * 8);
    reinterpret_cast<u32x4*>(out)[i] = v;
  }
}

// compact [n][Ps][Ps][C] -> the p x p interior starting at (o0, o0) goes to the patch's place in its request's state row
__global__ __launch_bounds__(256) void pc_scatter_kernel(const bf16_t* __restrict__ src, int Ps, int o0, int C, bf16_t* __restrict__ state, long state_row_elems,
                                                         const PcPatch* __restrict__ list, const PcSample* __restrict__ samp, int level, int p) {
  const PcPatch q = list[blockIdx.y];
  const PcSample s = samp[q.b];
  const int w = s.w >> level;
  const int y = blockIdx.x;
  const bf16_t* in = src + (((long)blockIdx.y * Ps + (y + o0)) * Ps + o0) * (long)C;
  bf16_t* out = state + (long)s.slot * state_row_elems + ((long)(q.py * p + y) * w + q.px * p) * C;
  const int n = p * (C >> 3);
  for (int i = threadIdx.x; i < n; i += 256) reinterpret_cast<u32x4*>(out)[i] = reinterpret_cast<const u32x4*>(in)[i];
}

// the listed patches of a whole-image batch tensor -> their places in the requests' state rows (a conv that was cheaper to run on the whole
// images than on the compact halo'd patches still renews the asking patches only)
__global__ __launch_bounds__(256) void pc_patch_store_kernel(const bf16_t* __restrict__ batch, bf16_t* __restrict__ state, long state_row_elems, int C,
                                                             const PcPatch* __restrict__ list, const PcSample* __restrict__ samp, int level, int p) {
  const PcPatch q = list[blockIdx.y];
  const PcSample s = samp[q.b];
  const int w = s.w >> level;
  const long off = ((long)(q.py * p + blockIdx.x) * w + q.px * p) * C;
  const u32x4* in = reinterpret_cast<const u32x4*>(batch + (s.row0 >> (2 * level)) * C + off);
  u32x4* out = reinterpret_cast<u32x4*>(state + (long)s.slot * state_row_elems + off);
  const int n = p * (C >> 3);
  for (int i = threadIdx.x; i < n; i += 256) out[i] = in[i];
}

// partial[patch][row] = sum over the patch's pixel row of (x - cached)^2; the host adds the p partials of a patch
__global__ __launch_bounds__(256) void pc_patch_sq_diff_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ state, long state_row_elems, int C,
                                                               const PcPatch* __restrict__ list, const PcSample* __restrict__ samp, int level, int p,
                                                               double* __restrict__ partial) {
  const PcPatch q = list[blockIdx.y];
  const PcSample s = samp[q.b];
  const int w = s.w >> level;
  const long off = ((long)(q.py * p + blockIdx.x) * w + q.px * p) * C;
  const bf16_t* pa = x + (s.row0 >> (2 * level)) * C + off;
  const bf16_t* pb = state + (long)s.slot * state_row_elems + off;
  float acc = 0.f;
  const int n = p * (C >> 3);
  for (int i = threadIdx.x; i < n; i += 256) {
    const u32x4 a = reinterpret_cast<const u32x4*>(pa)[i], b = reinterpret_cast<const u32x4*>(pb)[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float d0 = bf16lo_to_f32(a[e]) - bf16lo_to_f32(b[e]), d1 = bf16hi_to_f32(a[e]) - bf16hi_to_f32(b[e]);
      acc += d0 * d0 + d1 * d1;
    }
  }
  __shared__ double red[4];
  double d = (double)acc;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) d += __shfl_xor(d, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = d;
  __syncthreads();
  if (threadIdx.x == 0) partial[(long)blockIdx.y * p + blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// rows [row0, row0 + rows) of a token-major batch tensor <-> rows [srow0, srow0 + rows) of a request's state row (MMDiT: the chunks are token ranges)
__global__ __launch_bounds__(256) void pc_range_copy_kernel(bf16_t* __restrict__ batch, bf16_t* __restrict__ state, long state_row_elems, int C,
                                                            const PcRange* __restrict__ ranges, int to_batch) {
  const PcRange r = ranges[blockIdx.y];
  const long n = ((long)r.rows * C) >> 3;
  u32x4* pb = reinterpret_cast<u32x4*>(batch + r.row0 * C);
  u32x4* ps = reinterpret_cast<u32x4*>(state + (long)r.slot * state_row_elems + (long)r.srow0 * C);
  for (long v = (long)blockIdx.x * 256 + threadIdx.x; v < n; v += (long)gridDim.x * 256) {
    if (to_batch) pb[v] = ps[v]; else ps[v] = pb[v];
  }
}
__global__ __launch_bounds__(256) void pc_range_sq_diff_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ state, long state_row_elems, int C,
                                                               const PcRange* __restrict__ ranges, double* __restrict__ partial) {
  const PcRange r = ranges[blockIdx.y];
  const long n = ((long)r.rows * C) >> 3;
  const long per = (n + 63) / 64, v0 = (long)blockIdx.x * per, v1 = v0 + per < n ? v0 + per : n;
  const u32x4* pa = reinterpret_cast<const u32x4*>(x + r.row0 * C);
  const u32x4* pb = reinterpret_cast<const u32x4*>(state + (long)r.slot * state_row_elems + (long)r.srow0 * C);
  float acc = 0.f;
  for (long v = v0 + threadIdx.x; v < v1; v += 256) {
    const u32x4 a = pa[v], b = pb[v];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float d0 = bf16lo_to_f32(a[e]) - bf16lo_to_f32(b[e]), d1 = bf16hi_to_f32(a[e]) - bf16hi_to_f32(b[e]);
      acc += d0 * d0 + d1 * d1;
    }
  }
  __shared__ double red[4];
  double d = (double)acc;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) d += __shfl_xor(d, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = d;
  __syncthreads();
  if (threadIdx.x == 0) partial[(long)blockIdx.y * 64 + blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
int launch_pc_range_copy(hipStream_t st, void* batch, void* state, long state_row_elems, int C, const void* ranges, int n, int to_batch, long max_range_elems) {
  MX_CHECK(C % 8 == 0 && n > 0, "pc_range_copy: bad shape");
  const int gx = (int)std::max<long>(1, std::min<long>((max_range_elems / 8 + 255) / 256, 128));
  hipLaunchKernelGGL(pc_range_copy_kernel, dim3(gx, n), dim3(256), 0, st, (bf16_t*)batch, (bf16_t*)state, state_row_elems, C, (const PcRange*)ranges, to_batch);
  MX_LAUNCH_CHECK();
  return 0;
}
int launch_pc_range_sq_diff(hipStream_t st, const void* x, const void* state, long state_row_elems, int C, const void* ranges, int n, double* partial) {
  MX_CHECK(C % 8 == 0 && n > 0, "pc_range_sq_diff: bad shape");
  hipLaunchKernelGGL(pc_range_sq_diff_kernel, dim3(64, n), dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)state, state_row_elems, C, (const PcRange*)ranges, partial);
  MX_LAUNCH_CHECK();
  return 0;
}

int launch_pc_image_copy(hipStream_t st, void* batch, void* state, const void* samp, int B, int level, int C, long state_row_elems, int to_batch,
                         const float* vec, int ldvec, const void* residual, long max_image_elems, int gate) {
  MX_CHECK(C % 8 == 0 && B > 0, "pc_image_copy: C must be a multiple of 8");
  const int gx = (int)std::max<long>(1, std::min<long>((max_image_elems / 8 + 255) / 256, 128));
  hipLaunchKernelGGL(pc_image_copy_kernel, dim3(gx, B), dim3(256), 0, st, (bf16_t*)batch, (bf16_t*)state, (const PcSample*)samp, level, C, state_row_elems,
                     to_batch, vec, ldvec, (const bf16_t*)residual, gate);
  MX_LAUNCH_CHECK();
  return 0;
}
int launch_pc_rows_load_stats(hipStream_t st, void* batch, const void* state, const void* samp, int B, int level, int C, long state_row_elems, const void* residual,
                              float* stats1, float* fin, float eps, long max_image_rows) {
  MX_CHECK(C % 8 == 0 && B > 0 && residual != nullptr && (stats1 || fin), "pc_rows_load_stats: bad arguments");
  const int gx = (int)std::max<long>(1, std::min<long>((max_image_rows + 3) / 4, 256));
  hipLaunchKernelGGL(pc_rows_load_stats_kernel, dim3(gx, B), dim3(256), 0, st, (bf16_t*)batch, (const bf16_t*)state, (const PcSample*)samp, level, C, state_row_elems,
                     (const bf16_t*)residual, stats1, fin, eps);
  MX_LAUNCH_CHECK();
  return 0;
}
int launch_pc_gather(hipStream_t st, const void* src, int ld_src, int C, void* dst, const void* list, int n, const void* samp, int level, int p, int halo_lo,
                     int halo_hi, int up) {
  MX_CHECK(C % 8 == 0 && ld_src % 8 == 0 && n > 0 && p > 0, "pc_gather: bad shape");
  hipLaunchKernelGGL(pc_gather_kernel, dim3(p + halo_lo + halo_hi, n), dim3(256), 0, st, (const bf16_t*)src, ld_src, C, (bf16_t*)dst, (const PcPatch*)list,
                     (const PcSample*)samp, level, p, halo_lo, halo_hi, up);
  MX_LAUNCH_CHECK();
  return 0;
}
int launch_pc_scatter(hipStream_t st, const void* src, int Ps, int o0, int C, void* state, long state_row_elems, const void* list, int n, const void* samp,
                      int level, int p) {
  MX_CHECK(C % 8 == 0 && n > 0 && p > 0 && o0 >= 0 && o0 + p <= Ps, "pc_scatter: bad shape");
  hipLaunchKernelGGL(pc_scatter_kernel, dim3(p, n), dim3(256), 0, st, (const bf16_t*)src, Ps, o0, C, (bf16_t*)state, state_row_elems, (const PcPatch*)list,
                     (const PcSample*)samp, level, p);
  MX_LAUNCH_CHECK();
  return 0;
}
int launch_pc_patch_store(hipStream_t st, const void* batch, void* state, long state_row_elems, int C, const void* list, int n, const void* samp, int level, int p) {
  MX_CHECK(C % 8 == 0 && n > 0 && p > 0, "pc_patch_store: bad shape");
  hipLaunchKernelGGL(pc_patch_store_kernel, dim3(p, n), dim3(256), 0, st, (const bf16_t*)batch, (bf16_t*)state, state_row_elems, C, (const PcPatch*)list,
                     (const PcSample*)samp, level, p);
  MX_LAUNCH_CHECK();
  return 0;
}
int launch_pc_patch_sq_diff(hipStream_t st, const void* x, const void* state, long state_row_elems, int C, const void* list, int n, const void* samp, int level,
                            int p, double* partial) {
  MX_CHECK(C % 8 == 0 && n > 0 && p > 0, "pc_patch_sq_diff: bad shape");
  hipLaunchKernelGGL(pc_patch_sq_diff_kernel, dim3(p, n), dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)state, state_row_elems, C, (const PcPatch*)list,
                     (const PcSample*)samp, level, p, partial);
  MX_LAUNCH_CHECK();
  return 0;
}

}  // namespace mx
