// NHWC GroupNorm (+SiLU) and LayerNorm for gfx950.  HBM-bound kernels: 16-byte vector accesses,
// fp32 statistics, one read for the moments and one read + one write for the apply pass.
//
// GroupNorm serves the reference's PatchGroupNorm.forward (sduss/model_executor/modules/groupnorm.py:42-61):
//   patch == 0  -> nn.GroupNorm (is_sliced False, groupnorm.py:52)
//   patch  > 0  -> the sliced statistics of esymred_mp.groupnorm: per-(patch, group) mean and biased
//                  variance (norm_silu_concat.cu:41-81), merged over the patches of a latent as mean of
//                  means / mean of variances (cu:361-386), then y = x*(rstd*gamma) + (beta - rstd*gamma*mean)
//                  (cu:157-163).  The halo the reference materialises is not needed here: the implicit-GEMM
//                  conv reads neighbours from the whole NHWC image (see gemm_bf16.hip).
// SiLU (resnet.py:402,446; unet.py:514) is fused into the apply pass.
//
// Pass 1  gn_stats_kernel : per (image, spatial tile, channel) sum / sum-of-squares partials (fp32)
// Pass 2  gn_fold_kernel  : fp64 fold of the partials -> per (image, channel) scale/shift
// Pass 3  gn_apply_kernel : y = silu?(x*scale + shift)
#include <cstdlib>

#include "common.h"
#include "../../include/mxdenoise.h"

namespace mx {

// thread -> (channel vector cv = t % tpr, pixel lane pl = t / tpr); tpr = C/8 threads cover one pixel.
struct GnGeom {
  int B, H, W, C;
  int tpr;      // C / 8
  int L;        // pixel lanes per block
  int th, tw;   // spatial tile
  int tiles_y, tiles_x;
  // channel concatenation read in place (the UNet's skip connections, unet.py:458-462): channels [0, C1) come from x (row
  // stride C1), channels [C1, C) from x2 (row stride C - C1); x2 == nullptr: one source of C channels
  const bf16_t* x2;
  int C1;
};

// Grouped launch (mx_groupnorm_nhwc_grouped): the GroupNorms of all resolutions present in a mixed batch as ONE stats / fold / apply launch
// each.  The problems share C, the affine and the thread geometry (both depend on C only); each has its own images, spatial tiles and
// scratch.  A workgroup finds its problem from its index (blk0 = first workgroup of every problem in that launch); n == 1: an ordinary launch.
struct GnProb {
  GnGeom g;
  const bf16_t* x; bf16_t* y;
  float* part; float* coef;
  float* gpart;     // round 5: per (image, spatial tile, GROUP) sums instead of `part` (exact statistics only): the apply pass folds them itself, no fold launch
  long y_img;       // elements between the images of y
  int ppb;          // pixels per workgroup of the apply pass
  int patch;        // sliced statistics: patch edge in this problem's pixels (0 = exact)
};
struct GnGroup {
  GnProb p[MX_MAX_SEGS];
  int blk0[MX_MAX_SEGS + 1];
  int n;
};
__device__ __forceinline__ const GnProb& gn_locate(const GnGroup& G, int& local) {
  local = blockIdx.x;
  int s = 0;
#pragma unroll
  for (int i = 1; i < MX_MAX_SEGS; ++i) if (i < G.n && local >= G.blk0[i]) s = i;
  local -= G.blk0[s];
  return G.p[s];
}

__device__ __forceinline__ const bf16_t* gn_src(const bf16_t* x, const GnGeom& g, long pix, int ch) {
  if (g.x2 == nullptr) return x + pix * g.C + ch;
  return ch < g.C1 ? x + pix * g.C1 + ch : g.x2 + pix * (g.C - g.C1) + (ch - g.C1);
}

__global__ void gn_stats_kernel(const GnGroup G, const int groups) {
  extern __shared__ __attribute__((aligned(16))) float red[];  // [L][C][2], then [C][2] of channel sums (gpart form)
  const int t = threadIdx.x;
  int local;
  const GnProb& P = gn_locate(G, local);
  const GnGeom& g = P.g;
  const bf16_t* __restrict__ x = P.x;
  float* __restrict__ part = P.part;
  const int ntl = g.tiles_y * g.tiles_x;
  const int b = local / ntl;
  const int tile = local - b * ntl;
  const int ty = tile / g.tiles_x, tx = tile - ty * g.tiles_x;
  const int cv = t % g.tpr;
  const int pl = t / g.tpr;
  float s[8], ss[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { s[e] = 0.f; ss[e] = 0.f; }
  if (pl < g.L) {
    const int npix = g.th * g.tw;
    for (int pi = pl; pi < npix; pi += g.L) {
      const int py = pi / g.tw, px = pi - py * g.tw;
      const long pix = ((long)b * g.H + ty * g.th + py) * g.W + tx * g.tw + px;
      const u32x4 v = *reinterpret_cast<const u32x4*>(gn_src(x, g, pix, cv * 8));
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float lo = bf16lo_to_f32(v[e]), hi = bf16hi_to_f32(v[e]);
        s[2 * e] += lo; ss[2 * e] += lo * lo;
        s[2 * e + 1] += hi; ss[2 * e + 1] += hi * hi;
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      red[((long)pl * g.C + cv * 8 + e) * 2] = s[e];
      red[((long)pl * g.C + cv * 8 + e) * 2 + 1] = ss[e];
    }
  }
  __syncthreads();
  float* __restrict__ gpart = P.gpart;         // (uniform over the workgroup)
  float* chs = red + (long)g.L * g.C * 2;
  for (int c = t; c < g.C; c += blockDim.x) {
    float a = 0.f, q = 0.f;
    for (int l = 0; l < g.L; ++l) { a += red[((long)l * g.C + c) * 2]; q += red[((long)l * g.C + c) * 2 + 1]; }
    if (gpart != nullptr) { chs[2 * c] = a; chs[2 * c + 1] = q; continue; }
    float* dst = part + (((long)b * (g.tiles_y * g.tiles_x) + tile) * g.C + c) * 2;
    dst[0] = a; dst[1] = q;
  }
  if (gpart != nullptr) {                      // the tile's sums per GROUP, channels in order: what gn_apply_kernel<.., true> folds
    __syncthreads();
    const int cpg = g.C / groups;
    for (int gi = t; gi < groups; gi += blockDim.x) {
      float a = 0.f, q = 0.f;
      for (int i = 0; i < cpg; ++i) { a += chs[(gi * cpg + i) * 2]; q += chs[(gi * cpg + i) * 2 + 1]; }
      *reinterpret_cast<float2*>(gpart + (((long)b * ntl + tile) * groups + gi) * 2) = float2{a, q};
    }
  }
}

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// one workgroup (four waves) per (group, image): fp64 fold of tiles x channels-per-group, patch by patch.  Four waves because the fold is a
// chain of dependent loads: with one wave the 640..1280 items of a 128 x 128 level took 11 us per launch, 46 launches a step.
constexpr int kFoldThreads = 256;
// add_bias / add_rowbias (mx_groupnorm_nhwc_from_partials): the partial sums are those of x - c with c = add_bias[ch] + add_rowbias[image][ch] (what a
// producing conv's accumulators hold before its epilogue adds the bias and the time embedding); the sums of x follow in closed form per tile
__global__ __launch_bounds__(kFoldThreads) void gn_fold_kernel(const GnGroup G, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               int groups, float eps, const float* __restrict__ add_bias = nullptr,
                                                               const float* __restrict__ add_rowbias = nullptr, int ldrb = 0) {
  int local;
  const GnProb& P = gn_locate(G, local);
  const GnGeom& g = P.g;
  const float* __restrict__ part = P.part;
  float* __restrict__ coef = P.coef;
  const int patch = P.patch;
  const int b = local / groups;
  const int grp = local - b * groups;
  const int cpg = g.C / groups;
  const int ntiles = g.tiles_y * g.tiles_x;
  const int lane = threadIdx.x;
  // patches: (patch x patch) regions; tiles never straddle a patch (host guarantees th | patch, tw == patch)
  const int ppy = patch > 0 ? g.H / patch : 1;
  const int ppx = patch > 0 ? g.W / patch : 1;
  const int npatch = ppy * ppx;
  const int tpy = g.tiles_y / ppy, tpx = g.tiles_x / ppx;  // tiles per patch
  const int nitems = tpy * tpx * cpg;
  const double cnt = (double)tpy * tpx * g.th * g.tw * cpg;
  double acc_mean = 0.0, acc_var = 0.0;
  __shared__ double wsum[kFoldThreads / 64][2];
  for (int pidx = 0; pidx < npatch; ++pidx) {
    const int py = pidx / ppx, px = pidx - py * ppx;
    double s = 0.0, q = 0.0;
    for (int it = lane; it < nitems; it += kFoldThreads) {
      const int tl = it / cpg, c = it - tl * cpg;
      const int iy = tl / tpx, ix = tl - iy * tpx;
      const int tile = (py * tpy + iy) * g.tiles_x + px * tpx + ix;
      const float2 v = *reinterpret_cast<const float2*>(part + (((long)b * ntiles + tile) * g.C + grp * cpg + c) * 2);
      if (add_bias != nullptr) {
        const int chn = grp * cpg + c;
        const double cc = (double)add_bias[chn] + (add_rowbias != nullptr ? (double)add_rowbias[(long)b * ldrb + chn] : 0.0);
        const double npx = (double)(g.th * g.tw);
        s += (double)v.x + npx * cc;
        q += (double)v.y + 2.0 * cc * (double)v.x + npx * cc * cc;
      } else {
        s += (double)v.x;
        q += (double)v.y;
      }
    }
    s = wave_sum_d(s);
    q = wave_sum_d(q);
    __syncthreads();                                        // (the previous patch's readers are done with wsum)
    if ((lane & 63) == 0) { wsum[lane >> 6][0] = s; wsum[lane >> 6][1] = q; }
    __syncthreads();
    s = (wsum[0][0] + wsum[1][0]) + (wsum[2][0] + wsum[3][0]);   // fixed order: every thread holds the same sums
    q = (wsum[0][1] + wsum[1][1]) + (wsum[2][1] + wsum[3][1]);
    const double mean = s / cnt;
    double var = q / cnt - mean * mean;
    if (var < 0.0) var = 0.0;
    acc_mean += mean;   // mean of patch means      (norm_silu_concat.cu:383)
    acc_var += var;     // mean of patch variances  (norm_silu_concat.cu:384)
  }
  const double mean = acc_mean / npatch;
  const double rstd = 1.0 / sqrt(acc_var / npatch + (double)eps);
  for (int c = lane; c < cpg; c += kFoldThreads) {
    const int ch = grp * cpg + c;
    const float sc = (float)(rstd * (double)gamma[ch]);
    const float sf = (float)((double)beta[ch] - rstd * (double)gamma[ch] * mean);
    coef[((long)b * g.C + ch) * 2] = sc;
    coef[((long)b * g.C + ch) * 2 + 1] = sf;
  }
}

// FOLD (round 5): exact statistics whose pass left per-(image, tile, group) sums (GnProb::gpart).  Every workgroup folds its image's sums itself -- fp64, a
// FIXED order (eight strided sub-sums per group, then those in order), so every workgroup of every launch arrives at the same bits -- and derives the
// scale / shift of its own channels: the separate fold launch (46 a step, 6.6 us + a launch boundary each) goes.  <= kFoldGroups groups.
constexpr int kFoldGroups = 64;
template <bool SILU, bool FOLD>
__global__ void gn_apply_kernel(const GnGroup G, const float* __restrict__ gamma, const float* __restrict__ beta, const int groups, const float eps) {
  const int t = threadIdx.x;
  int local;
  const GnProb& P = gn_locate(G, local);
  const GnGeom& g = P.g;
  const bf16_t* __restrict__ x = P.x;
  bf16_t* __restrict__ y = P.y;
  const float* __restrict__ coef = P.coef;
  const int pix_per_block = P.ppb;
  const long y_img = P.y_img;
  const int nblk = (g.H * g.W + pix_per_block - 1) / pix_per_block;
  const int b = local / nblk;
  const int pblk = local - b * nblk;
  const int cv = t % g.tpr;
  const int pl = t / g.tpr;
  float sc[8], sf[8];
  if constexpr (FOLD) {
    __shared__ double fsub[kFoldGroups][8][2];
    __shared__ double mr[kFoldGroups][2];
    const int ntl = g.tiles_y * g.tiles_x;
    const float* __restrict__ gp = P.gpart + (long)b * ntl * groups * 2;
    for (int it = t; it < groups * 8; it += blockDim.x) {
      const int gi = it >> 3, sub = it & 7;
      double s = 0.0, q = 0.0;
      for (int tl = sub; tl < ntl; tl += 8) {
        const float2 v = *reinterpret_cast<const float2*>(gp + ((long)tl * groups + gi) * 2);
        s += (double)v.x; q += (double)v.y;
      }
      fsub[gi][sub][0] = s; fsub[gi][sub][1] = q;
    }
    __syncthreads();
    const int cpg = g.C / groups;
    const double cnt = (double)g.H * g.W * cpg;
    for (int gi = t; gi < groups; gi += blockDim.x) {
      double s = 0.0, q = 0.0;
#pragma unroll
      for (int sub = 0; sub < 8; ++sub) { s += fsub[gi][sub][0]; q += fsub[gi][sub][1]; }
      const double mean = s / cnt;
      double var = q / cnt - mean * mean;
      if (var < 0.0) var = 0.0;
      mr[gi][0] = mean;
      mr[gi][1] = 1.0 / sqrt(var + (double)eps);
    }
    __syncthreads();
    if (pl >= g.L) return;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int ch = cv * 8 + e;
      const int gi = ch / cpg;
      const double rg = mr[gi][1] * (double)gamma[ch];
      sc[e] = (float)rg;
      sf[e] = (float)((double)beta[ch] - rg * mr[gi][0]);
    }
  } else {
    if (pl >= g.L) return;
    const float* cf = coef + ((long)b * g.C + cv * 8) * 2;
#pragma unroll
    for (int e = 0; e < 8; ++e) { sc[e] = cf[2 * e]; sf[e] = cf[2 * e + 1]; }
  }
  const int hw = g.H * g.W;
  const int p0 = pblk * pix_per_block;
  const int p1 = min(p0 + pix_per_block, hw);
  for (int pi = p0 + pl; pi < p1; pi += g.L) {
    const u32x4 v = *reinterpret_cast<const u32x4*>(gn_src(x, g, (long)b * hw + pi, cv * 8));
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float lo = bf16lo_to_f32(v[e]) * sc[2 * e] + sf[2 * e];
      float hi = bf16hi_to_f32(v[e]) * sc[2 * e + 1] + sf[2 * e + 1];
      if (SILU) { lo = silu_f(lo); hi = silu_f(hi); }
      o[e] = pack_bf16x2(lo, hi);
    }
    *reinterpret_cast<u32x4*>(y + (long)b * y_img + (long)pi * g.C + cv * 8) = o;
  }
}

// ---- patch-parallel GroupNorm (mx_unet_forward_pp): the image's rows are split over `world` ranks; each rank folds its
//      partials to per-(image, group) {sum, sum of squares} in fp64, the ranks all-gather those 16-byte records, and every
//      rank finishes the same scale / shift (distrifuser modules/pp/groupnorm.py:9-98 exchanges E[x], E[x^2] likewise) ----
__global__ __launch_bounds__(64) void gn_pp_sums_kernel(const float* __restrict__ part, double* __restrict__ sums, GnGeom g, int groups) {
  const int grp = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
  const int cpg = g.C / groups;
  const int ntiles = g.tiles_y * g.tiles_x;
  double s = 0.0, q = 0.0;
  for (int it = lane; it < ntiles * cpg; it += 64) {
    const int tile = it / cpg, c = it - tile * cpg;
    const float* src = part + (((long)b * ntiles + tile) * g.C + grp * cpg + c) * 2;
    s += (double)src[0];
    q += (double)src[1];
  }
  s = wave_sum_d(s);
  q = wave_sum_d(q);
  if (lane == 0) { sums[((long)b * groups + grp) * 2] = s; sums[((long)b * groups + grp) * 2 + 1] = q; }
}

__global__ __launch_bounds__(64) void gn_pp_coef_kernel(const double* __restrict__ all_sums, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float* __restrict__ coef, int B, int C,
                                                        int groups, int world, double cnt, float eps,
                                                        const double* __restrict__ fresh_own, int own_rank) {
  const int grp = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
  const int cpg = C / groups;
  double s = 0.0, q = 0.0;
  for (int r = 0; r < world; ++r) {            // fixed rank order: every rank computes bit-identical coefficients
    s += all_sums[(((long)r * B + b) * groups + grp) * 2];
    q += all_sums[(((long)r * B + b) * groups + grp) * 2 + 1];
  }
  double var_fallback = 0.0;
  if (fresh_own) {
    // distrifuser "corrected_async_gn" (modules/pp/groupnorm.py:52-66): full = mean over ranks of the STALE slice moments + (fresh - stale) of
    // this rank's slice, the correction NOT divided by the number of ranks; in sums: sum_r stale_r + world * (fresh_own - stale_own)
    const long o = (((long)own_rank * B + b) * groups + grp) * 2, f = ((long)b * groups + grp) * 2;
    s += (double)world * (fresh_own[f] - all_sums[o]);
    q += (double)world * (fresh_own[f + 1] - all_sums[o + 1]);
    const double lc = cnt / world, lm = fresh_own[f] / lc;
    var_fallback = fresh_own[f + 1] / lc - lm * lm;       // "var = torch.where(var < 0, slice_var, var)"
    if (var_fallback < 0.0) var_fallback = 0.0;
  }
  const double mean = s / cnt;
  double var = q / cnt - mean * mean;
  if (var < 0.0) var = var_fallback;
  const double rstd = 1.0 / sqrt(var + (double)eps);
  for (int c = lane; c < cpg; c += 64) {
    const int ch = grp * cpg + c;
    coef[((long)b * C + ch) * 2] = (float)(rstd * (double)gamma[ch]);
    coef[((long)b * C + ch) * 2 + 1] = (float)((double)beta[ch] - rstd * (double)gamma[ch] * mean);
  }
}

static int gn_geom(GnGeom& g, int B, int H, int W, int C, int patch) {
  MX_CHECK(C % 8 == 0 && C / 8 <= 1024, "groupnorm: C must be a multiple of 8 and <= 8192");
  g.B = B; g.H = H; g.W = W; g.C = C;
  g.x2 = nullptr; g.C1 = C;
  g.tpr = C / 8;
  g.L = 1024 / g.tpr;
  if (g.L > 32) g.L = 32;
  // spatial tile: full rows of the image (or of the patch), about 256 pixels -- fewer at the small levels, where 256-pixel tiles leave
  // most CUs without a statistics block (8 images of 32 x 32: 32 blocks); halved until the launch has two blocks per CU
  constexpr int tile_pix = 256;
  int tw = (patch > 0) ? patch : W;
  int th = tile_pix / tw; if (th < 1) th = 1;
  const int hlim = (patch > 0) ? patch : H;
  while (hlim % th != 0) --th;
  while (th > 1 && th * tw >= 64 && (long)B * (H / th) * (W / tw) < 2L * cu_count()) { th /= 2; while (hlim % th != 0) --th; }   // >= 32 pixels per tile
  g.th = th; g.tw = tw;
  if (patch > 0) MX_CHECK(H % patch == 0 && W % patch == 0, "groupnorm: H, W must be multiples of patch");
  g.tiles_y = H / th; g.tiles_x = W / tw;
  return 0;
}

}  // namespace mx

namespace mx {
size_t gn_workspace_exact(int B, int H, int W, int C, int patch) {
  GnGeom g;
  if (patch >= H && patch >= W) patch = 0;
  if (gn_geom(g, B, H, W, C, patch)) return 0;
  return ((size_t)B * g.tiles_y * g.tiles_x * C * 2 + (size_t)B * C * 2) * sizeof(float);
}
}  // namespace mx

extern "C" size_t mx_groupnorm_nhwc_workspace_bytes(int B, int H, int W, int C) {
  // loose bound valid for every patch >= 2: the smallest spatial tile then holds >= 4 pixels
  const size_t tiles = (size_t)H * W / 4 + 1;
  return ((size_t)B * tiles * C * 2 + (size_t)B * C * 2) * sizeof(float) + 256;
}

extern "C" int mx_groupnorm_nhwc(void* stream, const void* x, void* y, const float* gamma, const float* beta,
                                 int B, int H, int W, int C, int groups, float eps, int silu, int patch,
                                 void* workspace) {
  return mx_groupnorm_nhwc_cat(stream, x, C, nullptr, y, gamma, beta, B, H, W, C, groups, eps, silu, patch, workspace);
}

namespace mx {
// fill problem i of G (geometry, operands, scratch carved from ws) -- returns the scratch bytes used, 0 on error
static size_t gn_fill(GnGroup& G, int i, const void* x, int C1, const void* x2, void* y, long y_img, int B, int H, int W, int C, int patch, char* ws) {
  GnProb& P = G.p[i];
  if (patch >= H && patch >= W) patch = 0;  // one patch per image == exact GroupNorm
  if (patch != 0 && patch < 2) { set_error("groupnorm: patch must be 0 or >= 2"); return 0; }
  if (gn_geom(P.g, B, H, W, C, patch)) return 0;
  if (x2) { P.g.x2 = (const bf16_t*)x2; P.g.C1 = C1; }
  const size_t ntiles = (size_t)P.g.tiles_y * P.g.tiles_x;
  P.x = (const bf16_t*)x; P.y = (bf16_t*)y; P.y_img = y_img; P.patch = patch;
  P.part = (float*)ws;
  P.gpart = nullptr;
  P.coef = P.part + (size_t)B * ntiles * C * 2;
  const int hw = H * W;
  P.ppb = P.g.L * 8 > hw ? hw : P.g.L * 8;
  return (((size_t)B * ntiles * C * 2 + (size_t)B * C * 2) * sizeof(float) + 255) & ~(size_t)255;
}
static void gn_prefix(GnGroup& G, int which, int groups) {      // which: 0 stats, 1 fold, 2 apply
  long t = 0;
  for (int i = 0; i < G.n; ++i) {
    G.blk0[i] = (int)t;
    const GnGeom& g = G.p[i].g;
    t += which == 0 ? (long)g.B * g.tiles_y * g.tiles_x : which == 1 ? (long)g.B * groups : (long)g.B * cdiv(g.H * g.W, G.p[i].ppb);
  }
  for (int i = G.n; i <= MX_MAX_SEGS; ++i) G.blk0[i] = (int)t;
}
// groups > 0: the per-group form (GnProb::gpart set on every problem)
static int gn_launch_stats(hipStream_t s, GnGroup& G, int C, int groups = 0) {
  const GnGeom& g = G.p[0].g;
  const int threads = ((g.tpr * g.L + 63) / 64) * 64;
  const size_t smem = ((size_t)g.L * C * 2 + (groups > 0 ? (size_t)C * 2 : 0)) * sizeof(float);
  MX_CHECK(smem <= 160 * 1024, "groupnorm: LDS budget exceeded");
  gn_prefix(G, 0, 0);
  hipLaunchKernelGGL(gn_stats_kernel, dim3(G.blk0[MX_MAX_SEGS]), dim3(threads), smem, s, G, groups);
  MX_LAUNCH_CHECK();
  return 0;
}
// fold: the workgroups fold the per-group sums themselves (gn_apply_kernel<.., true>); otherwise the coefficients a fold launch left are read
static int gn_launch_apply(hipStream_t s, GnGroup& G, int silu, bool fold = false, const float* gamma = nullptr, const float* beta = nullptr, int groups = 0, float eps = 0.f) {
  const GnGeom& g = G.p[0].g;
  const int threads = ((g.tpr * g.L + 63) / 64) * 64;
  gn_prefix(G, 2, 0);
  const dim3 grid(G.blk0[MX_MAX_SEGS]), block(threads);
  if (fold) {
    if (silu) hipLaunchKernelGGL((gn_apply_kernel<true, true>), grid, block, 0, s, G, gamma, beta, groups, eps);
    else hipLaunchKernelGGL((gn_apply_kernel<false, true>), grid, block, 0, s, G, gamma, beta, groups, eps);
  } else {
    if (silu) hipLaunchKernelGGL((gn_apply_kernel<true, false>), grid, block, 0, s, G, gamma, beta, groups, eps);
    else hipLaunchKernelGGL((gn_apply_kernel<false, false>), grid, block, 0, s, G, gamma, beta, groups, eps);
  }
  MX_LAUNCH_CHECK();
  return 0;
}
// MX_GN_FOLD=0: the three-launch form everywhere (A/B, tools/exp)
static bool gn_fold_in_apply() {
  static const bool on = [] { const char* e = getenv("MX_GN_FOLD"); return !(e && e[0] == '0'); }();
  return on;
}
}  // namespace mx

extern "C" size_t mx_groupnorm_nhwc_grouped_workspace_bytes(const mx_gn_problem* probs, int n, int C) {
  size_t t = 0;
  for (int i = 0; probs && i < n; ++i) t += mx_groupnorm_nhwc_workspace_bytes(probs[i].B, probs[i].H, probs[i].W, C);
  return t;
}

/* The GroupNorms of all resolutions present in a mixed batch as ONE stats / fold / apply launch each (see mx_gemm_seg).  The problems share
 * C, C1, the affine, groups, eps, silu and the sliced-statistics patch edge; each has its own images.  Per problem the arithmetic is
 * mx_groupnorm_nhwc_cat's. */
extern "C" int mx_groupnorm_nhwc_grouped(void* stream, const mx_gn_problem* probs, int n, int C1, const float* gamma, const float* beta, int C,
                                         int groups, float eps, int silu, int patch, void* workspace) {
  using namespace mx;
  MX_CHECK(probs && n >= 1 && n <= MX_MAX_SEGS && gamma && beta && workspace, "groupnorm: grouped launch needs 1..MX_MAX_SEGS problems and its operands");
  MX_CHECK(groups > 0 && C % groups == 0, "groupnorm: C % groups != 0");
  GnGroup G;
  G.n = n;
  char* ws = (char*)workspace;
  double bytes = 0;
  for (int i = 0; i < n; ++i) {
    const mx_gn_problem& q = probs[i];
    MX_CHECK(q.x && q.y && q.B > 0 && q.H > 0 && q.W > 0, "groupnorm: null operand / empty problem");
    MX_CHECK(q.x2 == nullptr || (C1 > 0 && C1 < C && C1 % 8 == 0 && (C - C1) % 8 == 0), "groupnorm: bad channel split");
    MX_CHECK((q.x2 != nullptr) == (probs[0].x2 != nullptr), "groupnorm: grouped launch: either every problem reads a channel concatenation or none");
    const size_t used = gn_fill(G, i, q.x, C1, q.x2, q.y, (long)q.H * q.W * C, q.B, q.H, q.W, C, patch, ws);
    if (!used) return 1;
    MX_CHECK(used <= mx_groupnorm_nhwc_workspace_bytes(q.B, q.H, q.W, C), "groupnorm: internal workspace bound exceeded");
    ws += mx_groupnorm_nhwc_workspace_bytes(q.B, q.H, q.W, C);
    bytes += 3.0 * 2.0 * q.B * q.H * (double)q.W * C;     // stats read + apply read + write
  }
  hipStream_t s = (hipStream_t)stream;
  prof_begin(s, PROF_NORM, 0.0, bytes);
  // exact statistics everywhere (the sliced form averages patch statistics: gn_fold_kernel): statistics per group, folded by the apply pass itself
  bool fold = gn_fold_in_apply() && groups <= kFoldGroups;
  for (int i = 0; i < n; ++i) fold = fold && G.p[i].patch == 0;
  if (fold) {
    for (int i = 0; i < n; ++i) G.p[i].gpart = G.p[i].part;      // (groups <= C: the per-group sums fit where the per-channel ones would go)
    if (gn_launch_stats(s, G, C, groups)) return 1;
    if (gn_launch_apply(s, G, silu, true, gamma, beta, groups, eps)) return 1;
    prof_end(s);
    return 0;
  }
  if (gn_launch_stats(s, G, C)) return 1;
  gn_prefix(G, 1, groups);
  hipLaunchKernelGGL(gn_fold_kernel, dim3(G.blk0[MX_MAX_SEGS]), dim3(kFoldThreads), 0, s, G, gamma, beta, groups, eps);
  MX_LAUNCH_CHECK();
  if (gn_launch_apply(s, G, silu)) return 1;
  prof_end(s);
  return 0;
}

extern "C" int mx_groupnorm_nhwc_cat(void* stream, const void* x, int C1, const void* x2, void* y, const float* gamma, const float* beta,
                                     int B, int H, int W, int C, int groups, float eps, int silu, int patch,
                                     void* workspace) {
  mx_gn_problem q;
  q.x = x; q.x2 = x2; q.y = y; q.B = B; q.H = H; q.W = W;
  return mx_groupnorm_nhwc_grouped(stream, &q, 1, C1, gamma, beta, C, groups, eps, silu, patch, workspace);
}

/* GroupNorm from the partial sums a producing launch left (mx_gemm_desc.gn_part_out): the fold reads them as "tiles" of `chunk` pixels; no statistics pass */
extern "C" int mx_groupnorm_nhwc_from_partials(void* stream, const void* x, void* y, const float* gamma, const float* beta, int B, int H, int W, int C, int groups,
                                               float eps, int silu, const float* part, int chunk, const float* add_bias, const float* add_rowbias, int ldrb,
                                               void* workspace) {
  using namespace mx;
  MX_CHECK(x && y && gamma && beta && part && workspace && B > 0 && H > 0 && W > 0, "groupnorm: null operand / empty problem");
  MX_CHECK(groups > 0 && C % groups == 0 && chunk > 0 && (H * W) % chunk == 0 && ((uintptr_t)part & 15) == 0, "groupnorm: bad groups / chunk / alignment");
  GnGroup G;
  G.n = 1;
  if (!gn_fill(G, 0, x, C, nullptr, y, (long)H * W * C, B, H, W, C, 0, (char*)workspace)) return 1;
  GnGroup F = G;                               // the fold's view: one "tile" per chunk of the producer, its sums where the statistics pass would have put them
  F.p[0].part = const_cast<float*>(part);
  F.p[0].g.tiles_y = H * W / chunk; F.p[0].g.tiles_x = 1; F.p[0].g.th = chunk; F.p[0].g.tw = 1;
  hipStream_t s = (hipStream_t)stream;
  prof_begin(s, PROF_NORM, 0.0, 2.0 * 2.0 * B * H * (double)W * C);
  gn_prefix(F, 1, groups);
  MX_CHECK(add_rowbias == nullptr || (add_bias != nullptr && ldrb >= C), "groupnorm: add_rowbias needs add_bias and ldrb >= C");
  hipLaunchKernelGGL(gn_fold_kernel, dim3(F.blk0[MX_MAX_SEGS]), dim3(kFoldThreads), 0, s, F, gamma, beta, groups, eps, add_bias, add_rowbias, ldrb);
  MX_LAUNCH_CHECK();
  if (gn_launch_apply(s, G, silu)) return 1;
  prof_end(s);
  return 0;
}

namespace mx {
// local part: stats -> per-(image, group) fp64 sums.  workspace: gn_workspace_exact(B, H, W, C, 0) bytes; sums: double [B][groups][2]
int launch_gn_pp_partial(hipStream_t s, const void* x, int C1, const void* x2, int B, int H, int W, int C, int groups, void* workspace, double* sums) {
  GnGroup G;
  G.n = 1;
  if (!gn_fill(G, 0, x, C1, x2, nullptr, 0, B, H, W, C, 0, (char*)workspace)) return 1;
  if (gn_launch_stats(s, G, C)) return 1;
  hipLaunchKernelGGL(gn_pp_sums_kernel, dim3(groups, B), dim3(64), 0, s, (const float*)G.p[0].part, sums, G.p[0].g, groups);
  MX_LAUNCH_CHECK();
  return 0;
}
// all_sums: double [world][B][groups][2]; y rows of image b start at y + b * y_img_elems; H = LOCAL rows, H_total = rows of the whole image
int launch_gn_pp_finish(hipStream_t s, const void* x, int C1, const void* x2, void* y, long y_img_elems, const float* gamma, const float* beta, const double* all_sums,
                        int world, int B, int H, int W, int C, int groups, int H_total, float eps, int silu, void* workspace,
                        const double* fresh_own, int own_rank) {
  GnGroup G;
  G.n = 1;
  if (!gn_fill(G, 0, x, C1, x2, y, y_img_elems, B, H, W, C, 0, (char*)workspace)) return 1;
  const double cnt = (double)H_total * W * (C / groups);
  hipLaunchKernelGGL(gn_pp_coef_kernel, dim3(groups, B), dim3(64), 0, s, all_sums, gamma, beta, G.p[0].coef, B, C, groups, world, cnt, eps, fresh_own, own_rank);
  return gn_launch_apply(s, G, silu);
}
}  // namespace mx

// ------------------------------------------------------------------------------------------
// LayerNorm over the last dim: one wave per row, values held in registers between the passes.
// Serves BasicTransformerBlock.norm1/2/3 (modules/transformer.py:191,239,266).
// ------------------------------------------------------------------------------------------
namespace mx {

template <int VPL>  // 16-byte chunks per lane (C <= 64*8*VPL)
__global__ __launch_bounds__(256) void layernorm_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, int M, int C, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int nch = C / 8;
  const bf16_t* xr = x + (long)row * C;
  float v[VPL][8];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int ch = lane + 64 * i;
    if (ch < nch) {
      const u32x4 u = *reinterpret_cast<const u32x4*>(xr + ch * 8);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[i][2 * e] = bf16lo_to_f32(u[e]);
        v[i][2 * e + 1] = bf16hi_to_f32(u[e]);
        sum += v[i][2 * e] + v[i][2 * e + 1];
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[i][e] = 0.f;
    }
  }
  const float mean = wave_sum(sum) / (float)C;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int ch = lane + 64 * i;
    if (ch < nch) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = v[i][e] - mean; sq += d * d; }
    }
  }
  const float rstd = rsqrtf(wave_sum(sq) / (float)C + eps);
  bf16_t* yr = y + (long)row * C;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int ch = lane + 64 * i;
    if (ch < nch) {
      const f32x4 one4 = {1.f, 1.f, 1.f, 1.f}, zero4 = {0.f, 0.f, 0.f, 0.f};     // gamma == nullptr: plain normalisation (affine folded elsewhere)
      const f32x4 g0 = gamma ? *reinterpret_cast<const f32x4*>(gamma + ch * 8) : one4;
      const f32x4 g1 = gamma ? *reinterpret_cast<const f32x4*>(gamma + ch * 8 + 4) : one4;
      const f32x4 b0 = gamma ? *reinterpret_cast<const f32x4*>(beta + ch * 8) : zero4;
      const f32x4 b1 = gamma ? *reinterpret_cast<const f32x4*>(beta + ch * 8 + 4) : zero4;
      float o[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        o[e] = (v[i][e] - mean) * rstd * g0[e] + b0[e];
        o[e + 4] = (v[i][e + 4] - mean) * rstd * g1[e] + b1[e];
      }
      u32x4 u = {pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7])};
      *reinterpret_cast<u32x4*>(yr + ch * 8) = u;
    }
  }
}

}  // namespace mx

extern "C" int mx_layernorm(void* stream, const void* x, void* y, const float* gamma, const float* beta,
                            int M, int C, float eps) {
  using namespace mx;
  MX_CHECK(x && y && ((gamma == nullptr) == (beta == nullptr)), "layernorm: null operand (gamma and beta are given together or not at all)");
  MX_CHECK(C % 8 == 0 && C <= 64 * 8 * 8, "layernorm: C must be a multiple of 8 and <= 4096");
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(cdiv(M, 4)), block(256);
  const int vpl = cdiv(C / 8, 64);
  const bf16_t* xp = (const bf16_t*)x; bf16_t* yp = (bf16_t*)y;
  if (vpl <= 1) hipLaunchKernelGGL((layernorm_kernel<1>), grid, block, 0, s, xp, yp, gamma, beta, M, C, eps);
  else if (vpl <= 2) hipLaunchKernelGGL((layernorm_kernel<2>), grid, block, 0, s, xp, yp, gamma, beta, M, C, eps);
  else if (vpl <= 3) hipLaunchKernelGGL((layernorm_kernel<3>), grid, block, 0, s, xp, yp, gamma, beta, M, C, eps);
  else if (vpl <= 4) hipLaunchKernelGGL((layernorm_kernel<4>), grid, block, 0, s, xp, yp, gamma, beta, M, C, eps);
  else hipLaunchKernelGGL((layernorm_kernel<8>), grid, block, 0, s, xp, yp, gamma, beta, M, C, eps);
  MX_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------
// AdaLN modulate for the MMDiT blocks: y = LN(x) * (1 + scale[b]) + shift[b]  (LayerNorm without affine, eps 1e-6),
// optionally a second output y2 with (scale2, shift2) sharing the normalisation (SD35AdaLayerNormZeroX).
// Serves norm1 / norm1_context / norm2 / norm2_context / norm_out of PatchJointTransformerBlock
// (modules/transformer.py:316-328, 359-360, 377-378; SD3Transformer.py:238).  scale/shift are fp32 rows of the one
// AdaLN projection GEMM of the step (row stride ldmod).
// ------------------------------------------------------------------------------------------
namespace mx {

// T5LayerNorm: y = x * rsqrt(mean(x^2) + eps) * w; one wave per row, the row stays in registers between the two passes
template <int VPL>
__global__ __launch_bounds__(256) void rmsnorm_rows_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, const float* __restrict__ w, int M, int C, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int nch = C / 8;
  const bf16_t* xr = x + (long)row * C;
  float v[VPL][8];
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int ch = lane + 64 * i;
    if (ch < nch) {
      const u32x4 u = *reinterpret_cast<const u32x4*>(xr + ch * 8);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[i][2 * e] = bf16lo_to_f32(u[e]); v[i][2 * e + 1] = bf16hi_to_f32(u[e]);
        sq += v[i][2 * e] * v[i][2 * e] + v[i][2 * e + 1] * v[i][2 * e + 1];
      }
    }
  }
  const float rstd = rsqrtf(wave_sum(sq) / (float)C + eps);
  bf16_t* yr = y + (long)row * C;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int ch = lane + 64 * i;
    if (ch < nch) {
      const f32x4 g0 = *reinterpret_cast<const f32x4*>(w + ch * 8), g1 = *reinterpret_cast<const f32x4*>(w + ch * 8 + 4);
      float o[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) { o[e] = v[i][e] * rstd * g0[e]; o[e + 4] = v[i][e + 4] * rstd * g1[e]; }
      *reinterpret_cast<u32x4*>(yr + ch * 8) = u32x4{pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7])};
    }
  }
}

// (sum, sum of squares) of every row: the one-slab statistics of the LayerNorm folded into its consumer GEMM (mx_gemm_desc.ln_stats)
__global__ __launch_bounds__(256) void row_stats_kernel(const bf16_t* __restrict__ x, int ldx, float* __restrict__ stats, int M, int C) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int nch = C / 8;
  const bf16_t* xr = x + (long)row * ldx;
  float s1 = 0.f, s2 = 0.f;
  for (int ch = lane; ch < nch; ch += 64) {
    const u32x4 u = *reinterpret_cast<const u32x4*>(xr + ch * 8);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float a = bf16lo_to_f32(u[e]), b = bf16hi_to_f32(u[e]);
      s1 += a + b;
      s2 += a * a + b * b;
    }
  }
  s1 = wave_sum(s1); s2 = wave_sum(s2);
  if (lane == 0) *reinterpret_cast<f32x2*>(stats + (long)row * 8) = f32x2{s1, s2};     // one slab, row pitch 4 (mxdenoise.h)
}

// rows of a mixed batch: group g holds rows [r0[g], r0[g + 1]) with rpb[g] rows per sample, its first sample is b0[g] (n == 0: one group)
struct RowGroups { int n; int r0[MX_MAX_SEGS + 1]; int rpb[MX_MAX_SEGS]; int b0[MX_MAX_SEGS]; };

template <int VPL>
__global__ __launch_bounds__(256) void layernorm_mod_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y,
                                                            bf16_t* __restrict__ y2, const float* __restrict__ scale,
                                                            const float* __restrict__ shift, const float* __restrict__ scale2,
                                                            const float* __restrict__ shift2, int ldmod, int M, int C,
                                                            int rows_per_batch, float eps, const RowGroups rg) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int nch = C / 8;
  const bf16_t* xr = x + (long)row * C;
  float v[VPL][8];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int ch = lane + 64 * i;
    if (ch < nch) {
      const u32x4 u = *reinterpret_cast<const u32x4*>(xr + ch * 8);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[i][2 * e] = bf16lo_to_f32(u[e]);
        v[i][2 * e + 1] = bf16hi_to_f32(u[e]);
        sum += v[i][2 * e] + v[i][2 * e + 1];
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[i][e] = 0.f;
    }
  }
  const float mean = wave_sum(sum) / (float)C;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int ch = lane + 64 * i;
    if (ch < nch) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = v[i][e] - mean; sq += d * d; }
    }
  }
  const float rstd = rsqrtf(wave_sum(sq) / (float)C + eps);
  int sample = row / rows_per_batch;
  if (rg.n > 0) {                              // mixed batch: the row's group, then its sample inside the group
    int g = 0;
#pragma unroll
    for (int k = 1; k < MX_MAX_SEGS; ++k) if (k < rg.n && row >= rg.r0[k]) g = k;
    sample = rg.b0[g] + (row - rg.r0[g]) / rg.rpb[g];
  }
  const long mrow = (long)sample * ldmod;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int ch = lane + 64 * i;
    if (ch < nch) {
      float n[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) n[e] = (v[i][e] - mean) * rstd;
      {
        const f32x4 s0 = *reinterpret_cast<const f32x4*>(scale + mrow + ch * 8);
        const f32x4 s1 = *reinterpret_cast<const f32x4*>(scale + mrow + ch * 8 + 4);
        const f32x4 h0 = *reinterpret_cast<const f32x4*>(shift + mrow + ch * 8);
        const f32x4 h1 = *reinterpret_cast<const f32x4*>(shift + mrow + ch * 8 + 4);
        float o[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) { o[e] = n[e] * (1.f + s0[e]) + h0[e]; o[e + 4] = n[e + 4] * (1.f + s1[e]) + h1[e]; }
        u32x4 u = {pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7])};
        *reinterpret_cast<u32x4*>(y + (long)row * C + ch * 8) = u;
      }
      if (y2) {
        const f32x4 s0 = *reinterpret_cast<const f32x4*>(scale2 + mrow + ch * 8);
        const f32x4 s1 = *reinterpret_cast<const f32x4*>(scale2 + mrow + ch * 8 + 4);
        const f32x4 h0 = *reinterpret_cast<const f32x4*>(shift2 + mrow + ch * 8);
        const f32x4 h1 = *reinterpret_cast<const f32x4*>(shift2 + mrow + ch * 8 + 4);
        float o[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) { o[e] = n[e] * (1.f + s0[e]) + h0[e]; o[e + 4] = n[e + 4] * (1.f + s1[e]) + h1[e]; }
        u32x4 u = {pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7])};
        *reinterpret_cast<u32x4*>(y2 + (long)row * C + ch * 8) = u;
      }
    }
  }
}

// RMSNorm over each 64-wide head of selected rows of a [rows, ld] bf16 matrix, in place:
//   x[r, 64*h : 64*h+64] *= rsqrt(mean(x^2) + eps) * w[h < heads_q ? wq : wk]
// 8 lanes x 16 bytes cover one head; a wave covers 8 heads of one row.  Serves norm_q / norm_k / norm_added_q /
// norm_added_k (attention.py:332-346, 377-388; diffusers RMSNorm(64, eps 1e-6)).
__global__ __launch_bounds__(256) void rmsnorm_heads_kernel(bf16_t* __restrict__ x, int ld, int nbatch, int rows_per_batch,
                                                            int batch_rows, int row_off, int heads_total, int heads_q,
                                                            const float* __restrict__ wq, const float* __restrict__ wk, float eps, float q_scale) {
  const int lane = threadIdx.x & 63;
  const int sub = lane & 7;       // 16-byte chunk inside the head
  const int hl = lane >> 3;       // head inside the wave's group of 8
  const int groups = (heads_total + 7) / 8;
  const long wid = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long total = (long)nbatch * rows_per_batch * groups;
  if (wid >= total) return;
  const int grp = (int)(wid % groups);
  const long r = wid / groups;
  const int b = (int)(r / rows_per_batch);
  const long row = (long)b * batch_rows + row_off + (r - (long)b * rows_per_batch);
  const int head = grp * 8 + hl;
  const bool ok = head < heads_total;
  bf16_t* p = x + row * ld + (ok ? head : 0) * 64 + sub * 8;
  u32x4 u = *reinterpret_cast<const u32x4*>(p);
  float v[8];
  float sq = 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    v[2 * e] = bf16lo_to_f32(u[e]); v[2 * e + 1] = bf16hi_to_f32(u[e]);
    sq += v[2 * e] * v[2 * e] + v[2 * e + 1] * v[2 * e + 1];
  }
  sq += __shfl_xor(sq, 1, 64); sq += __shfl_xor(sq, 2, 64); sq += __shfl_xor(sq, 4, 64);
  const float rs = rsqrtf(sq * (1.0f / 64.0f) + eps) * (head < heads_q ? q_scale : 1.0f);
  const float* w = (head < heads_q ? wq : wk) + sub * 8;
  u32x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = pack_bf16x2(v[2 * e] * rs * w[2 * e], v[2 * e + 1] * rs * w[2 * e + 1]);
  if (ok) *reinterpret_cast<u32x4*>(p) = o;
}

}  // namespace mx

static int launch_layernorm_mod(void* stream, const void* x, void* y, void* y2, const float* scale, const float* shift, const float* scale2,
                                const float* shift2, int ldmod, int M, int C, int rows_per_batch, float eps, const mx::RowGroups& rg) {
  using namespace mx;
  MX_CHECK(x && y && scale && shift, "layernorm_mod: null operand");
  MX_CHECK(!y2 || (scale2 && shift2), "layernorm_mod: second output needs scale2/shift2");
  MX_CHECK(C % 8 == 0 && C <= 64 * 8 * 8, "layernorm_mod: C must be a multiple of 8 and <= 4096");
  MX_CHECK(rows_per_batch > 0 && (rg.n > 0 || M % rows_per_batch == 0) && ldmod % 4 == 0, "layernorm_mod: bad batch geometry");
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(cdiv(M, 4)), block(256);
  const int vpl = cdiv(C / 8, 64);
  const bf16_t* xp = (const bf16_t*)x; bf16_t* yp = (bf16_t*)y; bf16_t* y2p = (bf16_t*)y2;
#define MX_LNMOD(V) hipLaunchKernelGGL((layernorm_mod_kernel<V>), grid, block, 0, s, xp, yp, y2p, scale, shift, scale2, shift2, ldmod, M, C, rows_per_batch, eps, rg)
  if (vpl <= 1) MX_LNMOD(1); else if (vpl <= 2) MX_LNMOD(2); else if (vpl <= 3) MX_LNMOD(3); else if (vpl <= 4) MX_LNMOD(4); else MX_LNMOD(8);
#undef MX_LNMOD
  MX_LAUNCH_CHECK();
  return 0;
}

extern "C" int mx_layernorm_mod(void* stream, const void* x, void* y, void* y2, const float* scale, const float* shift,
                                const float* scale2, const float* shift2, int ldmod, int M, int C, int rows_per_batch, float eps) {
  mx::RowGroups rg; rg.n = 0;
  return launch_layernorm_mod(stream, x, y, y2, scale, shift, scale2, shift2, ldmod, M, C, rows_per_batch, eps, rg);
}

/* Mixed batch: the rows are n groups one after the other, group g = batches[g] samples of rows_per_batch[g] rows; the modulation rows
 * (scale / shift, stride ldmod) are per sample in group order. */
extern "C" int mx_layernorm_mod_grouped(void* stream, const void* x, void* y, void* y2, const float* scale, const float* shift,
                                        const float* scale2, const float* shift2, int ldmod, int C, float eps, const int* batches,
                                        const int* rows_per_batch, int n) {
  MX_CHECK(batches && rows_per_batch && n >= 1 && n <= MX_MAX_SEGS, "layernorm_mod: grouped launch needs 1..MX_MAX_SEGS groups");
  mx::RowGroups rg; rg.n = n;
  int r = 0, b = 0;
  for (int g = 0; g < n; ++g) {
    MX_CHECK(batches[g] > 0 && rows_per_batch[g] > 0, "layernorm_mod: empty group");
    rg.r0[g] = r; rg.rpb[g] = rows_per_batch[g]; rg.b0[g] = b;
    r += batches[g] * rows_per_batch[g]; b += batches[g];
  }
  for (int g = n; g <= MX_MAX_SEGS; ++g) rg.r0[g] = r;
  for (int g = n; g < MX_MAX_SEGS; ++g) { rg.rpb[g] = 1; rg.b0[g] = 0; }
  return launch_layernorm_mod(stream, x, y, y2, scale, shift, scale2, shift2, ldmod, r, C, 1, eps, rg);
}

extern "C" int mx_rmsnorm_heads(void* stream, void* x, int ld, int nbatch, int rows_per_batch, int batch_rows, int row_off,
                                int heads_total, int heads_q, const float* wq, const float* wk, float eps, float q_scale) {
  using namespace mx;
  MX_CHECK(x && wq && wk, "rmsnorm_heads: null operand");
  MX_CHECK(ld % 8 == 0 && ld >= heads_total * 64, "rmsnorm_heads: bad row stride");
  MX_CHECK(nbatch > 0 && rows_per_batch > 0 && batch_rows >= row_off + rows_per_batch && row_off >= 0, "rmsnorm_heads: bad rows");
  const long waves = (long)nbatch * rows_per_batch * ((heads_total + 7) / 8);
  hipLaunchKernelGGL(rmsnorm_heads_kernel, dim3((unsigned)cdiv64(waves, 4)), dim3(256), 0, (hipStream_t)stream, (bf16_t*)x, ld,
                     nbatch, rows_per_batch, batch_rows, row_off, heads_total, heads_q, wq, wk, eps, q_scale);
  MX_LAUNCH_CHECK();
  return 0;
}

extern "C" int mx_row_stats(void* stream, const void* x, int ldx, float* stats, int M, int C) {
  using namespace mx;
  MX_CHECK(x && stats && M > 0, "row_stats: null operand");
  MX_CHECK(C % 8 == 0 && ldx >= C && ldx % 8 == 0 && (((uintptr_t)x & 15) | ((uintptr_t)stats & 7)) == 0, "row_stats: C and ldx must be multiples of 8, pointers aligned");
  hipLaunchKernelGGL(row_stats_kernel, dim3(cdiv(M, 4)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, ldx, stats, M, C);
  MX_LAUNCH_CHECK();
  return 0;
}

extern "C" int mx_rmsnorm(void* stream, const void* x, void* y, const float* w, int M, int C, float eps) {
  using namespace mx;
  MX_CHECK(x && y && w && M > 0, "rmsnorm: null operand");
  MX_CHECK(C % 8 == 0 && C <= 64 * 8 * 8 && ((((uintptr_t)x | (uintptr_t)y | (uintptr_t)w) & 15) == 0), "rmsnorm: C must be a multiple of 8 and <= 4096, pointers 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(cdiv(M, 4)), block(256);
  const int vpl = cdiv(C / 8, 64);
  const bf16_t* xp = (const bf16_t*)x; bf16_t* yp = (bf16_t*)y;
  if (vpl <= 1) hipLaunchKernelGGL((rmsnorm_rows_kernel<1>), grid, block, 0, s, xp, yp, w, M, C, eps);
  else if (vpl <= 2) hipLaunchKernelGGL((rmsnorm_rows_kernel<2>), grid, block, 0, s, xp, yp, w, M, C, eps);
  else if (vpl <= 4) hipLaunchKernelGGL((rmsnorm_rows_kernel<4>), grid, block, 0, s, xp, yp, w, M, C, eps);
  else hipLaunchKernelGGL((rmsnorm_rows_kernel<8>), grid, block, 0, s, xp, yp, w, M, C, eps);
  MX_LAUNCH_CHECK();
  return 0;
}
