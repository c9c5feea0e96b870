/* Plain-C restatement of the reference's native op (test infrastructure only; PARITY UNPINNED, see oracle/__init__.py).
 *
 * Follows /root/reference/sduss/model_executor/modules/kernels/norm_silu_concat.cu thread-for-thread in scalar loops:
 *   RowwiseMomentsCUDAKernel   :41-81   per (patch, group) mean and biased variance
 *   GetFullMeanAndRstd         :361-386 mean of patch means, rsqrt(mean of patch variances + eps)  (out of place here)
 *   NormSiluConcatCUDAKernel   :87-244  y = x*(rstd*gamma) + (beta - rstd*gamma*mean); interior + halo scatter
 *   MockNormSiluConcatCUDAKernel :248-358 halo scatter only
 * and norm_silu_concat.cpp:66-101 for the output allocation (zero-filled [N,C,H+2,W+2]).
 * fp32 throughout (the reference stores statistics in the tensor dtype, cpp:84-85).
 * Built by oracle/Makefile into oracle/_build/libgnhalo_ref.so and loaded with ctypes by tests/test_cpu.py.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

static void scatter(const float* y, float* out, const int* pidx, int N, int C, int H, int W) {
  const int W2 = W + 2, H2 = H + 2;
  memset(out, 0, sizeof(float) * (size_t)N * C * H2 * W2);
  for (int b = 0; b < N; ++b) {
    const int top = pidx[4 * b], left = pidx[4 * b + 1], bottom = pidx[4 * b + 2], right = pidx[4 * b + 3];
    for (int c = 0; c < C; ++c) {
      const float* src = y + ((size_t)b * C + c) * H * W;
#define OUT(n, r, q) out[(((size_t)(n) * C + c) * H2 + (r)) * W2 + (q)]
      for (int r = 0; r < H; ++r)
        for (int q = 0; q < W; ++q) OUT(b, r + 1, q + 1) = src[r * W + q];               /* cu:175 */
      for (int q = 0; q < W; ++q) {
        if (top != -1) OUT(top, H + 1, q + 1) = src[q];                                   /* cu:189-194 */
        if (bottom != -1) OUT(bottom, 0, q + 1) = src[(H - 1) * W + q];                   /* cu:195-200 */
      }
      for (int r = 0; r < H; ++r) {
        if (left != -1) {
          OUT(left, r + 1, W + 1) = src[r * W];                                           /* cu:208-209 */
          if (r == 0) OUT(left, 0, W + 1) = src[0];                                       /* cu:210-215 */
          if (r == H - 1) OUT(left, H + 1, W + 1) = src[r * W];                           /* cu:216-221 */
        }
        if (right != -1) {
          OUT(right, r + 1, 0) = src[r * W + W - 1];                                      /* cu:226-227 */
          if (r == 0) OUT(right, 0, 0) = src[W - 1];                                      /* cu:228-233 */
          if (r == H - 1) OUT(right, H + 1, 0) = src[r * W + W - 1];                      /* cu:234-239 */
        }
      }
#undef OUT
    }
  }
}

int gnhalo_mock(const float* x, float* out, const int* pidx, int N, int C, int H, int W) {
  scatter(x, out, pidx, N, C, H, W);
  return 0;
}

/* out: [N,C,H+2,W+2] if padding else [N,C,H,W] */
int gnhalo_groupnorm(const float* x, const float* gamma, const float* beta, float* out, int N, int C, int H, int W,
                     int cpg, double eps, int padding, const int* latent_offset, const int* patch_map, const int* pidx) {
  const int G = C / cpg, cnt = cpg * H * W;
  double* mean = (double*)malloc(sizeof(double) * N * G * 2);
  double* var = mean + (size_t)N * G;
  float* y = (float*)malloc(sizeof(float) * (size_t)N * C * H * W);
  if (!mean || !y) return 1;
  for (int i = 0; i < N * G; ++i) {                                                        /* cu:41-81 */
    const float* p = x + (size_t)i * cnt;
    double s = 0.0, q = 0.0;
    for (int j = 0; j < cnt; ++j) { s += p[j]; q += (double)p[j] * p[j]; }
    mean[i] = s / cnt;
    var[i] = q / cnt - mean[i] * mean[i];
  }
  for (int b = 0; b < N; ++b) {
    const int li = patch_map[b];                                                          /* 1-based, cu:372-374 */
    const int lo = latent_offset[li - 1], hi = latent_offset[li];
    for (int g = 0; g < G; ++g) {
      double m = 0.0, v = 0.0;
      for (int p = lo; p < hi; ++p) { m += mean[p * G + g]; v += var[p * G + g]; }       /* cu:377-382 */
      m /= (hi - lo);
      const double rstd = 1.0 / sqrt(v / (hi - lo) + eps);                                /* cu:383-384 */
      for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
        const double scale = rstd * (gamma ? gamma[c] : 1.0);                             /* cu:157 */
        const double shift = (beta ? beta[c] : 0.0) - scale * m;                          /* cu:158 */
        const float* src = x + ((size_t)b * C + c) * H * W;
        float* dst = y + ((size_t)b * C + c) * H * W;
        for (int j = 0; j < H * W; ++j) dst[j] = (float)(src[j] * scale + shift);         /* cu:163 */
      }
    }
  }
  if (padding) scatter(y, out, pidx, N, C, H, W);
  else memcpy(out, y, sizeof(float) * (size_t)N * C * H * W);
  free(mean); free(y);
  return 0;
}
