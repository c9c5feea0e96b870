// Flash-style fused attention forward for gfx950, head_dim 64, bf16 in / fp32 softmax / bf16 out.
//
//   O[b, i, h, :] = softmax_j(Q[b,i,h,:] . K[b,j,h,:] * scale) V[b,j,h,:]
//
// Serves the reference's xformers.ops.memory_efficient_attention(q, k, v) call sites
// (sduss/model_executor/modules/attention.py:86, 172, 195, 214): no mask, no dropout, scale 1/sqrt(d).
//
// MI355X-first structure (cdna guide section 3 "An accumulator tile as the next MFMA's operand"):
//   * one wave = 32 query rows, 4 waves / workgroup, KV tiles of 64 keys staged in LDS and shared;
//   * S^T = K Q^T with v_mfma_f32_32x32x16_bf16 (K rows are the A operand, Q rows the B operand), so a
//     lane owns ONE query column and 2x16 key rows: row max / sum are in-lane plus one cross-half shuffle;
//   * the S^T accumulator, converted to bf16, IS the B operand of O^T += V^T P^T -- no LDS round trip for P;
//     V arrives already transposed (V^T[d][key], written by the QKV GEMM epilogue), so its A fragments are
//     plain 8-byte LDS reads in the permuted k order the accumulator layout dictates
//     (row = 16s + 8(j>>2) + 4h + (j&3));
//   * LDS images are XOR-swizzled so both the ds_read_b128 K reads and the ds_read_b64 V^T reads are
//     bank-conflict free (K: chunk ^= (row>>1)&7; V^T: 8-byte piece ^= ((row>>1)&7)<<1 | (row>>4)&1).
#include "common.h"
#include "../../include/mxdenoise.h"

namespace mx {

struct AttnArgs {
  const bf16_t* q; const bf16_t* k; const bf16_t* vt; bf16_t* o;
  int ldq, ldk, ldvt, ldo;
  long vt_bstride;
  int B, H, Lq, Lk;
  float scale_log2;  // scale * log2(e)
};

constexpr int KT = 64;  // keys per tile

__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(const AttnArgs p) {
  __shared__ __attribute__((aligned(16))) bf16_t sK[2][KT * 64];
  __shared__ __attribute__((aligned(16))) bf16_t sV[2][64 * KT];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int r = lane & 31;   // query column owned by this lane (and fragment row)
  const int hh = lane >> 5;  // half-wave
  const int head = blockIdx.y;
  const int b = blockIdx.z;
  const int q0 = blockIdx.x * 128 + wave * 32;

  // ---- Q fragments (B operand of S^T = K Q^T): Q[q0 + r][16*ks + 8*hh .. +7] ----
  bf16x8 qf[4];
  {
    int qi = q0 + r;
    if (qi > p.Lq - 1) qi = p.Lq - 1;
    const bf16_t* qp = p.q + ((long)b * p.Lq + qi) * p.ldq + head * 64 + hh * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qp + ks * 16);
  }

  // ---- staging: 512 16-byte chunks per tile per operand, 2 per thread ----
  const int srow = tid >> 3;  // + 32*i
  const int sch = tid & 7;
  const bf16_t* kbase = p.k + (long)b * p.Lk * p.ldk + head * 64 + sch * 8;
  const bf16_t* vbase = p.vt + (long)b * p.vt_bstride + ((long)head * 64) * p.ldvt + sch * 8;
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  u32x4 rk[2], rv[2];

  auto load_tile = [&](int kt) {
    const int key0 = kt * KT;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int key = key0 + srow + 32 * i;
      rk[i] = (key < p.Lk) ? *reinterpret_cast<const u32x4*>(kbase + (long)key * p.ldk) : zero4;
    }
    const int nvalid = p.Lk - (key0 + sch * 8);  // valid keys in this thread's V^T chunk
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int d = srow + 32 * i;
      u32x4 v = zero4;
      if (nvalid > 0) {
        v = *reinterpret_cast<const u32x4*>(vbase + (long)d * p.ldvt + key0);
        if (nvalid < 8) {  // zero the keys >= Lk so that 0 * garbage can never be NaN
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int lo = 2 * e, hi = 2 * e + 1;
            unsigned w = v[e];
            if (lo >= nvalid) w &= 0xffff0000u;
            if (hi >= nvalid) w &= 0x0000ffffu;
            v[e] = w;
          }
        }
      }
      rv[i] = v;
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = srow + 32 * i;
      *reinterpret_cast<u32x4*>(&sK[buf][row * 64 + ((sch ^ ((row >> 1) & 7)) * 8)]) = rk[i];
      // V^T: chunk swizzle + swap the two 8-byte halves on rows with bit 4 set
      u32x4 v = rv[i];
      if ((row >> 4) & 1) v = u32x4{v[2], v[3], v[0], v[1]};
      *reinterpret_cast<u32x4*>(&sV[buf][row * 64 + ((sch ^ ((row >> 1) & 7)) * 8)]) = v;
    }
  };

  f32x16 oacc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) oacc[i][e] = 0.f;
  float m_run = -INFINITY;  // running max of s*scale_log2 for query column r (identical in both halves)
  float l_run = 0.f;        // this half-wave's partial row sum

  const int ntiles = (p.Lk + KT - 1) / KT;
  const float c = p.scale_log2;
  const int vf = (((r >> 1) & 7) << 1) | ((r >> 4) & 1);  // V^T piece swizzle of rows db*32 + r

  load_tile(0);
  store_tile(0);
  __syncthreads();

  for (int kt = 0; kt < ntiles; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < ntiles) load_tile(kt + 1);

    // ---- S^T = K Q^T : two 32-key blocks ----
    f32x16 s[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int e = 0; e < 16; ++e) s[kb][e] = 0.f;
      const int row = kb * 32 + r;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int ch = (2 * ks + hh) ^ ((row >> 1) & 7);
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(&sK[buf][row * 64 + ch * 8]);
        s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s[kb], 0, 0, 0);
      }
    }
    // mask keys beyond Lk (last tile only)
    if (kt * KT + KT > p.Lk) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int key = kt * KT + kb * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
          if (key >= p.Lk) s[kb][e] = -INFINITY;
        }
    }
    // ---- online softmax (log2 domain) ----
    float mx_ = s[0][0];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int e = 0; e < 16; ++e) mx_ = fmaxf(mx_, s[kb][e]);
    mx_ = fmaxf(mx_, __shfl_xor(mx_, 32, 64));
    const float m_new = fmaxf(m_run, mx_ * c);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    m_run = m_new;
    float psum = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float pe = __builtin_amdgcn_exp2f(s[kb][e] * c - m_new);
        s[kb][e] = pe;
        psum += pe;
      }
    l_run = l_run * alpha + psum;
    if (!__all(alpha == 1.0f)) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) oacc[i][e] *= alpha;
    }
    // ---- O^T += V^T P^T : P fragments straight from the S^T accumulators ----
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        u32x4 pw;
#pragma unroll
        for (int e = 0; e < 4; ++e) pw[e] = pack_bf16x2(s[kb][8 * s2 + 2 * e], s[kb][8 * s2 + 2 * e + 1]);
        const bf16x8 pf = __builtin_bit_cast(bf16x8, pw);
        const int sidx = 2 * kb + s2;  // 16-key step inside the tile
#pragma unroll
        for (int db = 0; db < 2; ++db) {
          const int row = db * 32 + r;
          const int p0 = (4 * sidx + hh) ^ vf;
          const int p1 = (4 * sidx + 2 + hh) ^ vf;
          const u32x2 a0 = *reinterpret_cast<const u32x2*>(&sV[buf][row * 64 + p0 * 4]);
          const u32x2 a1 = *reinterpret_cast<const u32x2*>(&sV[buf][row * 64 + p1 * 4]);
          const u32x4 aw = {a0[0], a0[1], a1[0], a1[1]};
          oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, aw), pf, oacc[db], 0, 0, 0);
        }
      }
    }
    if (kt + 1 < ntiles) store_tile(buf ^ 1);
    __syncthreads();
  }

  // ---- normalise and store O[q][d] ----
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;
  const int qi = q0 + r;
  if (qi < p.Lq) {
    bf16_t* op = p.o + ((long)b * p.Lq + qi) * p.ldo + head * 64;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = db * 32 + 8 * g + 4 * hh;
        u32x2 o = {pack_bf16x2(oacc[db][4 * g] * inv, oacc[db][4 * g + 1] * inv),
                   pack_bf16x2(oacc[db][4 * g + 2] * inv, oacc[db][4 * g + 3] * inv)};
        *reinterpret_cast<u32x2*>(op + d) = o;
      }
  }
}

}  // namespace mx

extern "C" int mx_attention(void* stream, const void* q, int ldq, const void* k, int ldk, const void* vt, int ldvt,
                            int64_t vt_batch_stride, void* o, int ldo, int B, int H, int Lq, int Lk, float scale) {
  using namespace mx;
  MX_CHECK(q && k && vt && o, "attention: null operand");
  MX_CHECK(B > 0 && H > 0 && Lq > 0 && Lk > 0, "attention: empty problem");
  MX_CHECK(ldq % 8 == 0 && ldk % 8 == 0 && ldvt % 8 == 0 && ldo % 4 == 0, "attention: strides must be multiples of 8 elements");
  MX_CHECK(ldq >= H * 64 && ldk >= H * 64 && ldo >= H * 64, "attention: row stride smaller than H*64");
  MX_CHECK(ldvt >= ((Lk + 7) / 8) * 8, "attention: ldvt must cover Lk rounded up to 8");
  MX_CHECK(vt_batch_stride % 8 == 0 && vt_batch_stride >= (int64_t)H * 64 * ldvt, "attention: bad vt_batch_stride");
  AttnArgs a;
  a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.vt = (const bf16_t*)vt; a.o = (bf16_t*)o;
  a.vt_bstride = (long)vt_batch_stride;
  a.ldq = ldq; a.ldk = ldk; a.ldvt = ldvt; a.ldo = ldo; a.B = B; a.H = H; a.Lq = Lq; a.Lk = Lk;
  a.scale_log2 = scale * 1.4426950408889634f;
  dim3 grid(cdiv(Lq, 128), H, B);
  prof_begin((hipStream_t)stream, PROF_ATTN, 4.0 * B * H * (double)Lq * Lk * 64.0,
             2.0 * B * H * 64.0 * (2.0 * Lq + 2.0 * Lk), B * H, Lq, Lk);
  hipLaunchKernelGGL(attn_fwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, a);
  prof_end((hipStream_t)stream);
  MX_LAUNCH_CHECK();
  return 0;
}
