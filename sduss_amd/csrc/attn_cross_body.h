// The short-key cross-attention body (attention.hip holds the description and the stand-alone kernel attn_cross_kernel): ONE wave computes
// XK_QPW = 64 queries of one (batch, head) against Lk <= 96 keys held in registers.  Shared with the chained launch of attn_tail.hip, which runs the
// 77-key cross-attention of a transformer layer between two GEMM stages of the same launch on exactly this code (same bits).
#pragma once
#include "common.h"
#include "../../include/mxdenoise.h"

namespace mx {

struct AttnArgs {
  const bf16_t* q; const bf16_t* k; const bf16_t* vt; bf16_t* o;
  int ldq, ldk, ldvt, ldo;
  long vt_bstride;
  int B, H, Lq, Lk;
  float scale_log2;  // scale * log2(e)
  int xcd_map;       // 1: XCD-aware workgroup order (needs B*H % 8 == 0)
  // patch-parallel K / V^T (mx_attention_prescaled_chunked): keys come in `key_chunk`-long chunks gathered from the ranks;
  // chunk c of batch b starts at k + c * k_cstride + b * k_bstride (rows of ldk) and vt + c * vt_cstride + b * vt_bstride
  int key_chunk;     // 0: one contiguous key range per batch
  long k_bstride, k_cstride, vt_cstride;
  int causal;        // 1: key j counts for query i only when j <= i (text encoders; attn_fwd_kernel only)
  const float* bias; // additive score bias [H][Lq][ldb], already in the kernel's log2 domain (T5 relative position bias; attn_fwd_kernel only)
  int ldb;
  int xq_wpb;        // short-key kernel: waves that share the 32-query blocks of one (batch, head) (launch_attention_group deals them so that the launch is ONE round)
};

typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ unsigned pack2(float lo, float hi) {   // one v_cvt_pk_bf16_f32
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{lo, hi}, bf16x2_t));
}

// max over the two half-waves (lane l and l^32) without the LDS crossbar: v_permlane32_swap exchanges the upper half
// of a with the lower half of b.  Inline asm because the builtin folds its two results when both inputs are one value.
__device__ __forceinline__ float max_across_halves(float x) {
  float a = x, b = x;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return fmaxf(a, b);
}

constexpr int XK_MAXBLK = 3;                   // 32-key blocks (Lk <= 96)
constexpr int XK_QPW = 64;                     // queries per wave (two 32-query blocks)

// WT: O leaves as write-through (sc1) buffer stores (a chained launch hands it to other workgroups; the launcher keeps O below 2 GB)
// ---- the pieces: one fragment of the head's K / V^T (global -> register, masked), the queries of a 32-query block, one block's arithmetic.  The stand-alone
//      kernel's wave keeps the 24 fragments in registers (attn_cross_wave); the chained launch stages them in LDS for the eight waves that share a head
//      (attn_tail.hip) -- the MFMA operands are the same bits either way. ----
constexpr int XK_KFRAGS = XK_MAXBLK * 4;       // K fragments of a head: [32-key block][16-feature step]
constexpr int XK_VFRAGS = 2 * XK_MAXBLK * 2;   // V^T fragments: [16-key step][32-feature half]

// K fragment (A operand of S^T): K[32 kb + r][16 ks + 8 hh ..]; rows >= Lk are zero
// LK > 0: the key count is that compile-time constant (the launcher dispatches LK = 77, the text encoders' sequence length, when every problem has it): block and
// step counts, the masks of the ragged last block and the exponentials of its all-padding elements are then resolved at compile time -- the runtime form spends
// ~150 selects and ~140 moves per 32-query block on them, against ~210 essential vector instructions.  LK = 0: runtime p.Lk.
template <int LK = 0>
__device__ __forceinline__ bf16x8 xk_kfrag(const AttnArgs& p, const int b, const int head, const int kb, const int ks, const int lane) {
  const int r = lane & 31, hh = lane >> 5;
  const int Lk = LK > 0 ? LK : p.Lk;
  const int nblk = (Lk + 31) >> 5;
  const int key = kb * 32 + r;
  const bf16_t* kp = p.k + ((long)b * Lk + (key < Lk ? key : 0)) * p.ldk + head * 64 + hh * 8;
  return (kb < nblk && key < Lk) ? *reinterpret_cast<const bf16x8*>(kp + ks * 16) : __builtin_bit_cast(bf16x8, u32x4{0u, 0u, 0u, 0u});
}
// V^T fragment (A operand of O^T): vt[64 head + 32 db + r][16 st + 8 hh ..] (MX_VT_POS order); keys >= Lk zeroed, the pad of a V^T row may hold anything
template <int LK = 0>
__device__ __forceinline__ bf16x8 xk_vfrag(const AttnArgs& p, const int b, const int head, const int st, const int db, const int lane) {
  const int r = lane & 31, hh = lane >> 5;
  const int Lk = LK > 0 ? LK : p.Lk;
  const int nst = (Lk + 15) >> 4;
  u32x4 v = {0u, 0u, 0u, 0u};
  if (st < nst) {
    v = *reinterpret_cast<const u32x4*>(p.vt + (long)b * p.vt_bstride + ((long)head * 64 + db * 32 + r) * p.ldvt + st * 16 + hh * 8);
    // element e of the word is position 16 st + 8 hh + e = key 16 st + 4 hh + (e & 3) + 8 (e >> 2)   (MX_VT_POS swaps bits 2 and 3)
    const int kbase = st * 16 + 4 * hh;
    if (kbase + 12 > Lk) {                     // some element may be past the end (only in the last step)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int k0 = kbase + 2 * (e & 1) + 8 * (e >> 1);
        unsigned w = v[e];
        if (k0 >= Lk) w &= 0xffff0000u;
        if (k0 + 1 >= Lk) w &= 0x0000ffffu;
        v[e] = w;
      }
    }
  }
  return __builtin_bit_cast(bf16x8, v);
}
// the lane's part of the queries q0 .. q0 + 31 of (b, head): B operand of S^T
__device__ __forceinline__ void xk_load_q(const AttnArgs& p, const int b, const int head, const int q0, const int lane, bf16x8 (&qf)[4]) {
  const int r = lane & 31, hh = lane >> 5;
  int qi = q0 + r;
  if (qi > p.Lq - 1) qi = p.Lq - 1;
  const bf16_t* qp = p.q + ((long)b * p.Lq + qi) * p.ldq + head * 64 + hh * 8;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qp + ks * 16);
}

// One 32-query block: S^T = K Q^T, single-pass softmax, O^T = V^T P^T, O through the wave's 4-KB LDS patch as whole rows.  kf(kb, ks) / vf(st, db) hand out
// the fragments (registers or LDS).  WT: O leaves as write-through (sc1) buffer stores (a chained launch hands it to other workgroups; the launcher keeps
// O below 2 GB).
template <bool PRE, bool WT, int LK = 0, class KF, class VF>
__device__ __forceinline__ void xk_block(const AttnArgs& p, const int b, const int head, const int q0, const bf16x8 (&qf)[4], KF&& kf, VF&& vf, char* const patch,
                                         const int lane) {
#pragma clang fp contract(off)      // the stand-alone kernel and the chained launch must round identically (no multiply-add pair below is meant to fuse)
  const int r = lane & 31;
  const int hh = lane >> 5;
  const auto o_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.o, 0, WT ? (int)((long)p.B * p.Lq * p.ldo * 2) : 0, 0x00020000);
  const int Lk = LK > 0 ? LK : p.Lk;
  const int nblk = (Lk + 31) >> 5;             // <= XK_MAXBLK (launcher)
  const int nst = (Lk + 15) >> 4;
  const float c = PRE ? 1.0f : p.scale_log2;
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  // ---- S^T = K Q^T ----
  f32x16 s[XK_MAXBLK];
#pragma unroll
  for (int kb = 0; kb < XK_MAXBLK; ++kb) {
    s[kb] = zero16;
    if (kb < nblk) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf(kb, ks), qf[ks], s[kb], 0, 0, 0);
    }
  }
  // ---- single-pass softmax over the lane's keys (block kb element e = key 32 kb + (e & 3) + 8 (e >> 2) + 4 hh) ----
  float mx_ = -INFINITY;
#pragma unroll
  for (int kb = 0; kb < XK_MAXBLK; ++kb) {
    if (kb >= nblk) continue;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int key = kb * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
      if (LK > 0 && kb * 32 + (e & 3) + 8 * (e >> 2) >= LK) { s[kb][e] = -INFINITY; continue; }       // past the end for both half-waves: known at compile time
      float v = s[kb][e] * c;
      if ((kb + 1) * 32 > Lk && key >= Lk) v = -INFINITY;
      s[kb][e] = v;
      mx_ = fmaxf(mx_, v);
    }
  }
  mx_ = max_across_halves(mx_);
  float psum = 0.f;
#pragma unroll
  for (int kb = 0; kb < XK_MAXBLK; ++kb) {
    if (kb >= nblk) continue;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      if (LK > 0 && kb * 32 + (e & 3) + 8 * (e >> 2) >= LK) { s[kb][e] = 0.f; continue; }             // exp2(-inf) = 0 exactly: the same value, no instruction
      const float pe = __builtin_amdgcn_exp2f(s[kb][e] - mx_);
      s[kb][e] = pe;
      psum += pe;
    }
  }
  const float inv = 1.0f / (psum + __shfl_xor(psum, 32, 64));
  // ---- O^T = V^T P^T ----
  f32x16 oacc[2] = {zero16, zero16};
#pragma unroll
  for (int st = 0; st < 2 * XK_MAXBLK; ++st) {
    if (st >= nst) continue;
    const int kb = st >> 1, s2 = st & 1;
    u32x4 pw;
#pragma unroll
    for (int e = 0; e < 4; ++e) pw[e] = pack2(s[kb][8 * s2 + 2 * e], s[kb][8 * s2 + 2 * e + 1]);
    const bf16x8 pf = __builtin_bit_cast(bf16x8, pw);
#pragma unroll
    for (int db = 0; db < 2; ++db) oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf(st, db), pf, oacc[db], 0, 0, 0);
  }
  // ---- O[q][d]: lane (r, hh) holds d = 32 db + 8 g + 4 hh + {0..3} of query r.  Through the wave's LDS patch (rows = queries,
  //      128 B, 16-byte chunk ^= row & 7) so that the global stores are whole rows. ----
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const u32x2 o = {pack2(oacc[db][4 * g] * inv, oacc[db][4 * g + 1] * inv), pack2(oacc[db][4 * g + 2] * inv, oacc[db][4 * g + 3] * inv)};
      const int chunk = (4 * db + g) ^ (r & 7);
      *reinterpret_cast<u32x2*>(patch + r * 128 + chunk * 16 + hh * 8) = o;
    }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (lane >> 3) + 8 * i;
    const int ch = lane & 7;
    const u32x4 o = *reinterpret_cast<const u32x4*>(patch + row * 128 + ((ch ^ (row & 7)) * 16));
    const int qi = q0 + row;
    if (qi < p.Lq) {
      if constexpr (WT) __builtin_amdgcn_raw_buffer_store_b128(o, o_rsrc, (int)((((long)b * p.Lq + qi) * p.ldo + head * 64 + ch * 8) * 2), 0, 16);
      else *reinterpret_cast<u32x4*>(p.o + ((long)b * p.Lq + qi) * p.ldo + head * 64 + ch * 8) = o;
    }
  }
}

// the stand-alone kernel's wave: the head's fragments in registers, XK_QPW queries in 32-query blocks, the next block's queries fetched while one computes
// the wave's share: `nblocks` consecutive 32-query blocks from query q_wave0 (wave-uniform)
template <bool PRE, int LK = 0>
__device__ __forceinline__ void attn_cross_wave(const AttnArgs& p, const int b, const int head, const int q_wave0, const int nblocks, char* const patch, const int lane) {
  bf16x8 kf[XK_MAXBLK][4];
#pragma unroll
  for (int kb = 0; kb < XK_MAXBLK; ++kb)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) kf[kb][ks] = xk_kfrag<LK>(p, b, head, kb, ks, lane);
  bf16x8 vf[2 * XK_MAXBLK][2];
#pragma unroll
  for (int st = 0; st < 2 * XK_MAXBLK; ++st)
#pragma unroll
    for (int db = 0; db < 2; ++db) vf[st][db] = xk_vfrag<LK>(p, b, head, st, db, lane);
  bf16x8 qf[4], qn[4];
  xk_load_q(p, b, head, q_wave0, lane, qf);
  for (int blk = 0; blk < nblocks; ++blk) {
    const int q0 = q_wave0 + blk * 32;
    if (q0 >= p.Lq) break;                     // wave-uniform
    if (blk + 1 < nblocks) xk_load_q(p, b, head, q0 + 32, lane, qn);
    xk_block<PRE, false, LK>(p, b, head, q0, qf, [&](int kb, int ks) __attribute__((always_inline)) { return kf[kb][ks]; },
                         [&](int st, int db) __attribute__((always_inline)) { return vf[st][db]; }, patch, lane);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = qn[ks];
  }
}

}  // namespace mx
