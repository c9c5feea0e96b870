// Shared helpers for the gfx950 kernels of mxdenoise.  CDNA4 only: wave = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>

namespace mx {

typedef unsigned short bf16_t;  // raw bf16 bits
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

constexpr int kWave = 64;

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }
__device__ __forceinline__ float bf16lo_to_f32(unsigned v) { return __uint_as_float(v << 16); }
__device__ __forceinline__ float bf16hi_to_f32(unsigned v) { return __uint_as_float(v & 0xffff0000u); }

// round-to-nearest-even via the hardware convert (keeps NaN a NaN, MI355X_MICROARCH "Correctness boundaries")
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(bf16_t, b);
}
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {     // one v_cvt_pk_bf16_f32 (round to nearest even, as the scalar cast)
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  const f32x2_t v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }
// exact (erf) GELU, as diffusers' GEGLU uses F.gelu default
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }

// erf by Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7, far below the bf16 output rounding): 1 rcp + 1 exp + 6 fma
// instead of the two-branch libm polynomial -- the GEGLU epilogue evaluates it for every FF1 output element.
__device__ __forceinline__ float erf_as(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float r = 1.0f - p * t * __expf(-ax * ax);
  return copysignf(r, x);
}
// tanh-approximated GELU (torch F.gelu(approximate="tanh")): 0.5 x (1 + tanh(sqrt(2/pi) (x + 0.044715 x^3))) = x * sigmoid(2u)
__device__ __forceinline__ float gelu_tanh_f(float x) {
  const float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
  return x * __builtin_amdgcn_rcpf(1.0f + __expf(-2.0f * u));    // (v_rcp_f32, 1 ulp: the IEEE division sequence is ~10 instructions per element)
}
__device__ __forceinline__ float quick_gelu_f(float x) { return x / (1.0f + __expf(-1.702f * x)); }
// GELU (erf form) by Abramowitz-Stegun 7.1.28: erf(z) = 1 - 1 / (1 + a1 z + ... + a6 z^6)^16 (z >= 0, |abs err| <= 3e-7), so
//   gelu(x) = x (1 - h) for x >= 0 and x h for x < 0 (= max(x, 0) - |x h|), with h = 0.5 / poly(|x| / sqrt 2)^16
// -- ONE transcendental (v_rcp) per element instead of the two of 7.1.26 (rcp + exp), and a plain polynomial that the two-element form
// runs on packed fp32 instructions (v_pk_fma_f32 / v_pk_mul_f32).  Measured against the float64 erf form: |abs err| <= 7.1e-7 over
// [-12, 12] (7.1.26: 4.6e-7); the GEGLU epilogue evaluates it for every FF1 output element and was VALU-bound on it (round 3,
// tools/exp/timeline_v4.py).
typedef float gelu_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ gelu_f32x2 gelu_fast2(const gelu_f32x2 x) {
  const gelu_f32x2 ax = __builtin_elementwise_abs(x) * 0.70710678118654752f;
  gelu_f32x2 p = ax * 0.0000430638f + 0.0002765672f;
  p = p * ax + 0.0001520143f;
  p = p * ax + 0.0092705272f;
  p = p * ax + 0.0422820123f;
  p = p * ax + 0.0705230784f;
  p = p * ax + 1.0f;
  p = p * p; p = p * p; p = p * p; p = p * p;              // ^16 (overflow -> inf -> h = 0: erf = 1)
  const gelu_f32x2 r = {__builtin_amdgcn_rcpf(p[0]), __builtin_amdgcn_rcpf(p[1])};
  const gelu_f32x2 xh = x * (r * 0.5f);          // same sign as x
  // x >= 0: x - x h;  x < 0: x h = -|x h|   ==   max(x, 0) - |x h|   (a v_max and a v_sub with the abs modifier: no compare / select)
  return gelu_f32x2{fmaxf(x[0], 0.f) - fabsf(xh[0]), fmaxf(x[1], 0.f) - fabsf(xh[1])};
}
__device__ __forceinline__ float gelu_fast(float x) { return gelu_fast2(gelu_f32x2{x, x})[0]; }
// tanh-form GELU, two elements: x sigmoid(2 u), u = sqrt(2 / pi) (x + 0.044715 x^3) -- the polynomial and the scaling on packed fp32
__device__ __forceinline__ gelu_f32x2 gelu_tanh2(const gelu_f32x2 x) {
  const gelu_f32x2 w = (x * x) * (0.044715f * 0.7978845608028654f) + 0.7978845608028654f;
  const gelu_f32x2 a = (x * w) * (-2.0f * 1.4426950408889634f);          // -2 u log2(e)
  const gelu_f32x2 d = gelu_f32x2{__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])} + 1.0f;
  return x * gelu_f32x2{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---- host side ----
void set_error(const std::string& msg);
int cu_count();   // compute units of the current device, whole XCD groups (multiple of 8); 256 when no device is visible (host-side planning)
#define MX_CHECK(cond, msg)                   \
  do {                                        \
    if (!(cond)) {                            \
      ::mx::set_error(std::string(msg));      \
      return 1;                               \
    }                                         \
  } while (0)
#define MX_HIP(expr)                                                                     \
  do {                                                                                   \
    hipError_t _e = (expr);                                                              \
    if (_e != hipSuccess) {                                                              \
      ::mx::set_error(std::string(#expr) + ": " + hipGetErrorString(_e));                \
      return 1;                                                                          \
    }                                                                                    \
  } while (0)
#define MX_LAUNCH_CHECK() MX_HIP(hipGetLastError())

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace mx

// ---- optional per-launch timing (bench.py roofline leg): hipEvents on the launch stream ----
namespace mx {
enum ProfKind { PROF_GEMM128 = 0, PROF_GEMM64 = 1, PROF_CONV128 = 2, PROF_CONV64 = 3, PROF_ATTN = 4, PROF_NORM = 5, PROF_GEMM_V2_160 = 6, PROF_CONV_V2_160 = 7, PROF_GEMM_V2_128 = 8, PROF_CONV_V2_128 = 9, PROF_GEMM_V3_256 = 10, PROF_ATTN_CROSS = 11, PROF_ATTN_TAIL = 12, PROF_KINDS = 13 };
bool prof_enabled();
void prof_begin(hipStream_t s, int kind, double flops, double bytes, int m = 0, int n = 0, int k = 0);
void prof_end(hipStream_t s);
}  // namespace mx
