// Flash-style fused attention forward for gfx950, head_dim 64, bf16 in / fp32 softmax / bf16 out.
//
//   O[b, i, h, :] = softmax_j(Q[b,i,h,:] . K[b,j,h,:] * scale) V[b,j,h,:]
//
// Serves the reference's xformers.ops.memory_efficient_attention(q, k, v) call sites
// (sduss/model_executor/modules/attention.py:86, 172, 195, 214) and the SD3 joint / dual attention
// (F.scaled_dot_product_attention, attention.py:324, 350, 394): no mask, no dropout, scale 1/sqrt(d).
//
// MI355X-first structure (cdna guide section 3 "An accumulator tile as the next MFMA's operand"):
//   * one wave = 32 query rows, 4 waves / workgroup, KV tiles of 64 keys staged in LDS and shared;
//   * S^T = K Q^T with v_mfma_f32_32x32x16_bf16 (K rows are the A operand, Q rows the B operand), so a
//     lane owns ONE query column and 2x16 key rows: row max / sum are in-lane plus one cross-half shuffle;
//   * the S^T accumulator, converted to bf16, IS the B operand of O^T += V^T P^T -- no LDS round trip for P;
//     V arrives already transposed (V^T[d][key], written by the QKV GEMM epilogue) AND with bits 2/3 of the key index
//     swapped (MX_VT_POS), which is exactly the k order the accumulator layout dictates
//     (element j of lane half h is key 16s + 8(j>>2) + 4h + (j&3)): a V^T A fragment is one aligned ds_read_b128;
//   * both LDS images share one XOR swizzle (16-byte chunk ^= (row>>1)&7 on 128-byte rows), so K and V^T reads use
//     the same per-lane offsets and are bank-conflict free;
//   * the kernel is VALU-bound at head_dim 64 (32 exp2 + ~90 other vector ops against 16 MFMAs per wave and tile), so the
//     loop carries no avoidable vector work: every LDS address is a loop-invariant per-lane offset whose buffer bit is
//     toggled by one XOR per tile, key/V^T masking exists only in a separate tail-tile path, P is packed with the
//     two-operand v_cvt_pk_bf16_f32, and the first MFMA of each S^T block takes a zero accumulator literal.
//   * PRE variant (mx_attention_prescaled; what the step plans use): Q arrives already multiplied by scale*log2(e) (the
//     producing GEMM epilogue / RMSNorm does it in fp32 before its single rounding), and the softmax reference m is
//     subtracted BY THE MATRIX CORE: a fifth k-step multiplies a constant K column of ones with a Q column holding -m, so
//     S' = s*c - m leaves the MFMA and p = exp2(S') needs no per-element vector op.  m is a lazy reference, not the
//     running max: it is set from the first tile and raised only when a tile's maximum exceeds it by more than 8 (then
//     that tile, O and l are rescaled -- exact, softmax is invariant to the reference; p <= 256 and l >= 1 always hold).
//     Per 64-key tile and lane this removes 32 v_fma, the alpha exp2 and 16 v_pk_mul for two extra MFMAs on a matrix pipe
//     that was 41 % busy while the vector ALU was 72 % busy (rocprofv3 SQ_ACTIVE_INST_VALU).
#include <algorithm>
#include <cstdlib>

#ifndef MX_AEXP
#define MX_AEXP 0   // ablation builds of attn_fwd_dma_kernel (tools/exp/build_attn_variants.sh): timing only
#endif

#include "common.h"
#include "../../include/mxdenoise.h"
#include "attn_cross_body.h"

namespace mx {


// Grouped launch (mx_attention_prescaled_grouped): up to MX_MAX_SEGS problems -- the resolutions of a mixed batch, each with its own
// sequence lengths, batch and operand bases -- in ONE launch.  A workgroup finds its problem from its index with two compares and runs the
// ordinary kernel body on it: `p` is a wave-uniform reference into the kernel-argument segment (scalar loads).  n == 1: an ordinary launch.
struct AttnGroup {
  AttnArgs g[MX_MAX_SEGS];
  int blk0[MX_MAX_SEGS + 1];           // first workgroup of every problem: a multiple of 8, so that (workgroup & 7) is the hardware XCD in EVERY
                                       // problem (advisor, round 3: after a problem with an odd workgroup count the later problems' XCD-aware map
                                       // was rotated and their K / V^T panels lost their L2)
  int nblk[MX_MAX_SEGS];               // workgroups of every problem; the padding workgroups behind them exit at once
  int n;
};

// workgroup -> (problem, query block, batch * H + head); ROWS = query rows per workgroup.  Consecutive workgroup ids are dealt round-robin to
// the 8 XCDs (one L2 each): with the plain order (query block fastest) the query blocks of one head land on all 8 XCDs and each L2 fetches that
// head's K / V^T (rocprofv3 FETCH_SIZE: 4.4x the algorithmic bytes at Lk = 4429).  When the number of (batch, head) pairs of a problem is a
// multiple of 8, XCD x instead owns the pairs == x (mod 8) and walks their query blocks consecutively (its workgroup count is then a multiple
// of 8 as well, so the next problem starts on XCD 0 again).
// ROWS == 0: the short-key kernel's deal -- (xq_wpb + 3) / 4 workgroups per (batch, head), see launch_attention_group
template <int ROWS>
__device__ __forceinline__ const AttnArgs& attn_locate(const AttnGroup& ga, int& qb, int& bh) {
  int lin = blockIdx.x;
  int s = 0;
#pragma unroll
  for (int i = 1; i < MX_MAX_SEGS; ++i) if (i < ga.n && lin >= ga.blk0[i]) s = i;
  const AttnArgs& p = ga.g[s];
  lin -= ga.blk0[s];
  if (lin >= ga.nblk[s]) { qb = -1; bh = 0; return p; }      // padding up to the next problem's 8-aligned start
  const int gx = ROWS > 0 ? (p.Lq + ROWS - 1) / (ROWS > 0 ? ROWS : 1) : (p.xq_wpb + 3) >> 2;
  if (p.xcd_map) {
    const int local = lin >> 3;
    qb = local % gx;
    bh = ((local / gx) << 3) + (lin & 7);
  } else {
    qb = lin % gx;
    bh = lin / gx;
  }
  return p;
}

constexpr int KT = 64;                 // keys per tile
constexpr int kBufBytes = 16384;       // one ring buffer: K tile (8 KB) then V^T tile (8 KB)

// EXTRA: the additive-bias and causal-mask forms of the text encoders, a separate instantiation (as run-time branches they cost the
// 60 cross-attention launches of a UNet step 3.4 us each: profiles r02_f vs r02_g)
template <bool PRE, bool EXTRA = false>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(const AttnGroup ga) {
  __shared__ __attribute__((aligned(16))) char smem[2 * kBufBytes];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int r = lane & 31;   // query column owned by this lane (and fragment row)
  const int hh = lane >> 5;  // half-wave
  // workgroup -> (query block, head, batch).  Consecutive workgroup ids are dealt round-robin to the 8 XCDs (one L2 each):
  // with the plain x-fastest order the query blocks of one head land on all 8 XCDs and each L2 fetches that head's K / V^T
  // (rocprofv3 FETCH_SIZE: 4.4x the algorithmic bytes at Lk = 4429).  When the number of (batch, head) pairs is a multiple
  // of 8, XCD x instead owns the pairs == x (mod 8) and walks their query blocks consecutively.
  int qb, bh;
  const AttnArgs& p = attn_locate<128>(ga, qb, bh);      // (problem of a grouped launch, query block, batch * H + head)
  if (qb < 0) return;
  const int head = bh % p.H;
  const int b = bh / p.H;
  const int q0 = qb * 128 + wave * 32;

  // ---- Q fragments (B operand of S^T = K Q^T): Q[q0 + r][16*ks + 8*hh .. +7] ----
  bf16x8 qf[4];
  {
    int qi = q0 + r;
    if (qi > p.Lq - 1) qi = p.Lq - 1;
    const bf16_t* qp = p.q + ((long)b * p.Lq + qi) * p.ldq + head * 64 + hh * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qp + ks * 16);
  }

  // ---- staging: 512 16-byte chunks per tile per operand, 2 per thread (rows srow, srow+32; chunk sch) ----
  const int srow = tid >> 3;
  const int sch = tid & 7;
  const bf16_t* kbase = p.k + (long)b * p.k_bstride + head * 64 + sch * 8;
  const bf16_t* vbase = p.vt + (long)b * p.vt_bstride + ((long)head * 64) * p.ldvt + sch * 8;
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  u32x4 rk[2], rv[2];
  // (row>>1)&7 is the same for rows srow and srow+32: one store offset, +4096 B for the second row
  unsigned st_off = (unsigned)(srow * 128 + ((sch ^ ((srow >> 1) & 7)) * 16));     // byte offset inside the K image of buffer 0

  auto load_tile = [&](int kt) {               // full tile: no predicates
    int key0 = kt * KT;
    const bf16_t* kb_ = kbase;
    const bf16_t* vb_ = vbase;
    if (p.key_chunk > 0) {                     // tiles never straddle a chunk (key_chunk % 64 == 0)
      const int ch = key0 / p.key_chunk;
      key0 -= ch * p.key_chunk;
      kb_ += ch * p.k_cstride;
      vb_ += ch * p.vt_cstride;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      rk[i] = *reinterpret_cast<const u32x4*>(kb_ + (long)(key0 + srow + 32 * i) * p.ldk);
      rv[i] = *reinterpret_cast<const u32x4*>(vb_ + (long)(srow + 32 * i) * p.ldvt + key0);
    }
  };
  auto load_tile_tail = [&](int kt) {          // last, partial tile: keys >= Lk are zero-filled (0 * garbage must not be NaN)
    const int key0 = kt * KT;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int key = key0 + srow + 32 * i;
      rk[i] = (key < p.Lk) ? *reinterpret_cast<const u32x4*>(kbase + (long)key * p.ldk) : zero4;
    }
    // chunk sch holds positions 8*sch .. +7 = keys ka .. ka+3 then ka+8 .. ka+11 (MX_VT_POS), ka = 16*(sch>>1) + 4*(sch&1)
    const int ka = key0 + 16 * (sch >> 1) + 4 * (sch & 1);
    const int nvalid = p.Lk - ka;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      u32x4 v = zero4;
      if (nvalid > 0) {
        v = *reinterpret_cast<const u32x4*>(vbase + (long)(srow + 32 * i) * p.ldvt + key0);
        if (nvalid < 12) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int kk = 2 * (e & 1) + 8 * (e >> 1);   // key offset of the word's low half
            unsigned w = v[e];
            if (kk >= nvalid) w &= 0xffff0000u;
            if (kk + 1 >= nvalid) w &= 0x0000ffffu;
            v[e] = w;
          }
        }
      }
      rv[i] = v;
    }
  };
  auto store_tile = [&](unsigned off) {        // off = st_off with the destination buffer's bit set
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      *reinterpret_cast<u32x4*>(smem + off + i * 4096) = rk[i];
      *reinterpret_cast<u32x4*>(smem + off + 8192 + i * 4096) = rv[i];
    }
  };

  // ---- per-lane LDS read offsets for buffer 0 (loop invariant; the buffer bit is XOR-toggled per tile) ----
  // row r, chunk 2*step + hh: the K fragment of k-step `step` and (at +8192) the V^T fragment of 16-key step `step`
  unsigned koff[4];
  {
    const int swz = (r >> 1) & 7;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) koff[ks] = (unsigned)(r * 128 + (((2 * ks + hh) ^ swz) * 16));
  }

  f32x16 oacc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) oacc[i][e] = 0.f;
  float m_run = -INFINITY;  // !PRE: running max of s*scale_log2 for query column r (identical in both halves)
  float l_run = 0.f;        // this half-wave's partial row sum
  // PRE: the reference k-step.  A operand: K column of ones = element k == 0 of the step (held by the hh == 0 lanes);
  // B operand: -m_ref (bf16-representable) in the same element of query column r.
  const unsigned one_lo = hh == 0 ? 0x3F80u : 0u;
  const bf16x8 kone = __builtin_bit_cast(bf16x8, u32x4{one_lo, 0u, 0u, 0u});
  u32x4 qm = {0u, 0u, 0u, 0u};
  float m_ref = 0.f;

  const int ntiles = (p.Lk + KT - 1) / KT;
  const bool ragged = (p.Lk % KT) != 0;
  const float c = p.scale_log2;
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  if (ntiles == 1 && ragged) load_tile_tail(0); else load_tile(0);
  store_tile(st_off);
  __syncthreads();

  for (int kt = 0; kt < ntiles; ++kt) {
    const bool has_next = kt + 1 < ntiles;
    if (has_next) {
      if (ragged && kt + 2 == ntiles) load_tile_tail(kt + 1); else load_tile(kt + 1);
    }

    // ---- S^T = K Q^T : two 32-key blocks.  All eight K fragments are requested before the first MFMA and the two
    //      accumulator chains alternate, so neither LDS latency nor the MFMA dependency sits between issues. ----
    bf16x8 fr[8];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) fr[kb * 4 + ks] = *reinterpret_cast<const bf16x8*>(smem + koff[ks] + kb * 4096);
    __builtin_amdgcn_sched_barrier(0);
    f32x16 s[2];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
        s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[kb * 4 + ks], qf[ks], ks == 0 ? zero16 : s[kb], 0, 0, 0);
    }
    if constexpr (PRE) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
        s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kone, __builtin_bit_cast(bf16x8, qm), s[kb], 0, 0, 0);
    }
    // V^T fragments (same registers): in flight while the softmax runs on the vector ALU
#pragma unroll
    for (int sidx = 0; sidx < 4; ++sidx)
#pragma unroll
      for (int db = 0; db < 2; ++db) fr[sidx * 2 + db] = *reinterpret_cast<const bf16x8*>(smem + koff[sidx] + 8192 + db * 4096);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (EXTRA) if (p.bias != nullptr) {   // additive bias, four consecutive keys per load; BEFORE the masks (the pad of a bias row may hold anything) (key = 32 kb + 8 g + 4 hh + {0..3})
      int qi = q0 + r;
      if (qi > p.Lq - 1) qi = p.Lq - 1;
      const float* brow = p.bias + ((long)head * p.Lq + qi) * p.ldb + kt * KT + 4 * hh;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 b4 = *reinterpret_cast<const f32x4*>(brow + kb * 32 + 8 * g);
#pragma unroll
          for (int e = 0; e < 4; ++e) s[kb][4 * g + e] += b4[e];
        }
    }
    if (ragged && !has_next) {          // mask keys beyond Lk (tail tile only)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int key = kt * KT + kb * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
          if (key >= p.Lk) s[kb][e] = -INFINITY;
        }
    }
    if constexpr (EXTRA) if (p.causal) {            // keys after the query never count (every tile: short sequences only)
      const int qi = q0 + r;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int key = kt * KT + kb * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
          if (key > qi) s[kb][e] = -INFINITY;
        }
    }
    // ---- online softmax (log2 domain) ----
    float mx_ = s[0][0];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int e = 0; e < 16; ++e) mx_ = fmaxf(mx_, s[kb][e]);
    mx_ = max_across_halves(mx_);
    float psum = 0.f;
    if constexpr (PRE) {
      // s already holds s*c - m_ref.  Move the reference only on the first tile or when a row ran more than 8 above it.
      const bool first = kt == 0;
      if (first || __any(mx_ > 8.0f)) {
        float delta = 0.f;
        if (first || mx_ > 8.0f) {
          const float nr = bf16lo_to_f32(pack2(m_ref + mx_, 0.f));        // new reference, bf16-representable
          delta = nr - m_ref;
          m_ref = nr;
        }
        qm[0] = hh == 0 ? (pack2(-m_ref, 0.f) & 0xffffu) : 0u;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int e = 0; e < 16; ++e) s[kb][e] -= delta;
        if (!first) {
          const float alpha = __builtin_amdgcn_exp2f(-delta);
          l_run *= alpha;
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) oacc[i][e] *= alpha;
        }
      }
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const float pe = __builtin_amdgcn_exp2f(s[kb][e]);
          s[kb][e] = pe;
          psum += pe;
        }
      l_run += psum;
    } else {
      const float m_new = fmaxf(m_run, mx_ * c);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      m_run = m_new;
      // (packed v_pk_fma_f32 / v_pk_add_f32 forms of this loop measured 0-3 % SLOWER in same-run A/B: scalar kept)
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
    }
    // ---- O^T += V^T P^T : P fragments straight from the S^T accumulators ----
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        u32x4 pw;
#pragma unroll
        for (int e = 0; e < 4; ++e) pw[e] = pack2(s[kb][8 * s2 + 2 * e], s[kb][8 * s2 + 2 * e + 1]);
        const bf16x8 pf = __builtin_bit_cast(bf16x8, pw);
        const int sidx = 2 * kb + s2;  // 16-key step inside the tile
#pragma unroll
        for (int db = 0; db < 2; ++db)
          oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[sidx * 2 + db], pf, oacc[db], 0, 0, 0);
      }
    }
    // the next tile lives in the other buffer: toggle the buffer bit of every offset
    st_off ^= kBufBytes;
    if (has_next) store_tile(st_off);
#pragma unroll
    for (int i = 0; i < 4; ++i) koff[i] ^= kBufBytes;
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
        u32x2 o = {pack2(oacc[db][4 * g] * inv, oacc[db][4 * g + 1] * inv),
                   pack2(oacc[db][4 * g + 2] * inv, oacc[db][4 * g + 3] * inv)};
        *reinterpret_cast<u32x2*>(op + d) = o;
      }
  }
}


// ----------------------------------------------------------------------------------------------------------------------
// The same kernel with K / V^T tiles staged by LDS-DMA two tiles ahead (round 2), for key counts that are whole tiles.
// attn_fwd_kernel fetches tile kt + 1 into registers while it computes tile kt and writes it to LDS at the end of the iteration:
// one tile of compute (~1.2k cycles per wave) has to cover an L2 / Infinity-Cache round trip under load, and 16 VGPRs hold the
// staged tile.  Here the tile goes L2 -> LDS directly (global_load_lds_dwordx4, swizzle on the source address) into a ring of
// three 16-KB buffers, issued right after the barrier of tile kt for tile kt + 2; one counted s_waitcnt vmcnt(4) + one raw
// s_barrier per tile.  The compute body is attn_fwd_kernel's.
// ----------------------------------------------------------------------------------------------------------------------
template <bool PRE>
__global__ __launch_bounds__(256, 2) void attn_fwd_dma_kernel(const AttnGroup ga) {
  #if (MX_AEXP & 32)
  __shared__ __attribute__((aligned(16))) char smem[3 * kBufBytes + 40 * 1024];
#elif (MX_AEXP & 64)
  __shared__ __attribute__((aligned(16))) char smem[3 * kBufBytes + 12 * 1024];
#else
  __shared__ __attribute__((aligned(16))) char smem[3 * kBufBytes];
#endif

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int r = lane & 31;   // query column owned by this lane (and fragment row)
  const int hh = lane >> 5;  // half-wave
  // workgroup -> (query block, head, batch).  Consecutive workgroup ids are dealt round-robin to the 8 XCDs (one L2 each):
  // with the plain x-fastest order the query blocks of one head land on all 8 XCDs and each L2 fetches that head's K / V^T
  // (rocprofv3 FETCH_SIZE: 4.4x the algorithmic bytes at Lk = 4429).  When the number of (batch, head) pairs is a multiple
  // of 8, XCD x instead owns the pairs == x (mod 8) and walks their query blocks consecutively.
  int qb, bh;
  const AttnArgs& p = attn_locate<128>(ga, qb, bh);      // (problem of a grouped launch, query block, batch * H + head)
  if (qb < 0) return;
  const int head = bh % p.H;
  const int b = bh / p.H;
  const int q0 = qb * 128 + wave * 32;

  // ---- Q fragments (B operand of S^T = K Q^T): Q[q0 + r][16*ks + 8*hh .. +7] ----
  bf16x8 qf[4];
  {
    int qi = q0 + r;
    if (qi > p.Lq - 1) qi = p.Lq - 1;
    const bf16_t* qp = p.q + ((long)b * p.Lq + qi) * p.ldq + head * 64 + hh * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qp + ks * 16);
  }

  // ---- staging by LDS-DMA into a three-buffer ring, two tiles ahead: 512 16-byte chunks per tile per operand, 2 per thread.
  //      Chunk c = i * 256 + tid lands at byte 16 c of the image (an LDS-DMA writes base + 16 * lane); it is row c >> 3, slot c & 7,
  //      and holds logical chunk slot ^ ((row >> 1) & 7) of that row: the swizzle is applied on the SOURCE address. ----
  // Addresses: a wave-uniform base that moves with the tile plus a per-lane 32-bit offset fixed for the whole kernel (as in attn_fwd64_kernel:
  // the per-tile 64-bit address arithmetic cost 6 % of the launch there).
  const char* kbase = (const char*)(p.k + (long)b * p.k_bstride + head * 64);
  const char* vbase = (const char*)(p.vt + (long)b * p.vt_bstride + ((long)head * 64) * p.ldvt);
  unsigned voffk[2], voffv[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = i * 256 + tid;
    const int row = c >> 3, ch = (c & 7) ^ ((row >> 1) & 7);
    voffk[i] = (unsigned)(row * p.ldk * 2 + ch * 16);
    voffv[i] = (unsigned)(row * p.ldvt * 2 + ch * 16);
  }
  const char* knext = kbase;
  const char* vnext = vbase;
  int chunk_left = p.key_chunk > 0 ? p.key_chunk : 0x7fffffff;
  int chunk_id = 0;
  const unsigned wave_img = (unsigned)__builtin_amdgcn_readfirstlane(wave) * 1024u;
  auto issue_tile = [&](int /*kt: tiles are issued in order*/, int buf) __attribute__((always_inline)) {
    char* img = smem + buf * kBufBytes + wave_img;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(knext + voffk[i]),
                                       (__attribute__((address_space(3))) void*)(img + i * 4096), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vnext + voffv[i]),
                                       (__attribute__((address_space(3))) void*)(img + 8192 + i * 4096), 16, 0, 0);
    }
    knext += (long)KT * p.ldk * 2;
    vnext += KT * 2;
    chunk_left -= KT;
    if (chunk_left == 0) {                     // tiles never straddle a chunk (key_chunk % 64 == 0)
      ++chunk_id;
      chunk_left = p.key_chunk;
      knext = kbase + (long)chunk_id * p.k_cstride * 2;
      vnext = vbase + (long)chunk_id * p.vt_cstride * 2;
    }
  };

  // ---- per-lane LDS read offsets for buffer 0 (loop invariant; the buffer bit is XOR-toggled per tile) ----
  // row r, chunk 2*step + hh: the K fragment of k-step `step` and (at +8192) the V^T fragment of 16-key step `step`
  unsigned koff[4];
  {
    const int swz = (r >> 1) & 7;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) koff[ks] = (unsigned)(r * 128 + (((2 * ks + hh) ^ swz) * 16));
  }

  f32x16 oacc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) oacc[i][e] = 0.f;
  float m_run = -INFINITY;  // !PRE: running max of s*scale_log2 for query column r (identical in both halves)
  float l_run = 0.f;        // this half-wave's partial row sum
  // PRE: the scores leave their MFMA chain as s - m_ref: the chain STARTS from the accumulator negm = (-m_ref, ... 16 times) of the lane's query column
  // (round 5; the guide's "row constants as the initial accumulator").  It was a fifth k-step (a K column of ones times -m_ref): two MFMAs per tile on the
  // scores' dependency chain; the splat costs 16 registers and is rewritten only when the reference moves.
  f32x16 negm = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  float m_ref = 0.f;

  const int ntiles = (p.Lk + KT - 1) / KT;
  const bool ragged = false;                   // the launcher sends ragged key counts to attn_fwd_kernel
  const float c = p.scale_log2;
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  issue_tile(0, 0);
  if (ntiles > 1) issue_tile(1, 1);
  int buf = 0;                                 // ring buffer of the tile being computed
  unsigned bofs = 0;                           // its byte offset

  for (int kt = 0; kt < ntiles; ++kt) {
    const bool has_next = kt + 1 < ntiles;
    // tile kt has landed (this thread's part; the next tile's four DMAs may stay in flight) ...
#if !(MX_AEXP & 2)
    if (has_next) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
#if !(MX_AEXP & 4)
    __builtin_amdgcn_s_barrier();              // ... for every thread, and every wave has left the buffer of tile kt - 1
#endif
    asm volatile("" ::: "memory");
#if !(MX_AEXP & 2)
    if (kt + 2 < ntiles) issue_tile(kt + 2, buf == 0 ? 2 : buf - 1);   // (kt + 2) % 3
#endif

    // ---- S^T = K Q^T : two 32-key blocks.  All eight K fragments are requested before the first MFMA and the two
    //      accumulator chains alternate, so neither LDS latency nor the MFMA dependency sits between issues. ----
    bf16x8 fr[8];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) fr[kb * 4 + ks] = *reinterpret_cast<const bf16x8*>(smem + bofs + koff[ks] + kb * 4096);
    __builtin_amdgcn_sched_barrier(0);
    f32x16 s[2];
#if (MX_AEXP & 8)
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int e = 0; e < 16; ++e) s[kb][e] = __builtin_bit_cast(float, __builtin_bit_cast(u32x4, fr[kb * 4 + (e & 3)])[e & 3] << 16) + oacc[kb][e];
#else
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
        s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[kb * 4 + ks], qf[ks], ks == 0 ? (PRE ? negm : zero16) : s[kb], 0, 0, 0);
    }
#endif
    // V^T fragments (same registers): in flight while the softmax runs on the vector ALU
#pragma unroll
    for (int sidx = 0; sidx < 4; ++sidx)
#pragma unroll
      for (int db = 0; db < 2; ++db) fr[sidx * 2 + db] = *reinterpret_cast<const bf16x8*>(smem + bofs + koff[sidx] + 8192 + db * 4096);
    __builtin_amdgcn_sched_barrier(0);
    if (ragged && !has_next) {          // mask keys beyond Lk (tail tile only)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int key = kt * KT + kb * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
          if (key >= p.Lk) s[kb][e] = -INFINITY;
        }
    }
    // ---- online softmax (log2 domain) ----
    auto col_max = [&]() __attribute__((always_inline)) {
      float mx_ = s[0][0];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int e = 0; e < 16; ++e) mx_ = fmaxf(mx_, s[kb][e]);
      return max_across_halves(mx_);
    };
    float psum = 0.f;
    if constexpr (PRE) {
      // s already holds s*c - m_ref, m_ref a lazy reference: set from tile 0, raised only when a row runs more than 8 above it.  After
      // tile 0 the exponentials run SPECULATIVELY and the sums are watched (as in attn_fwd64_kernel): a score > 8 above the reference
      // makes its p, hence the lane's sum, exceed 256; then (rarely) the tile is redone exactly.  The row maximum leaves the hot path.
      auto exps = [&]() __attribute__((always_inline)) {
        float ps = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int e = 0; e < 16; ++e) {
#if (MX_AEXP & 1)
            const float pe = s[kb][e];
#else
            const float pe = __builtin_amdgcn_exp2f(s[kb][e]);
#endif
            s[kb][e] = pe;
            ps += pe;
          }
        return ps;
      };
      auto move_reference = [&](bool first) __attribute__((always_inline)) {       // on exact scores in s
        const float mx_ = col_max();
        float delta = 0.f;
        if (first || mx_ > 8.0f) {
          const float nr = bf16lo_to_f32(pack2(m_ref + mx_, 0.f));        // new reference, bf16-representable
          delta = nr - m_ref;
          m_ref = nr;
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) negm[e] = -m_ref;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int e = 0; e < 16; ++e) s[kb][e] -= delta;
        if (!first) {
          const float alpha = __builtin_amdgcn_exp2f(-delta);
          l_run *= alpha;
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) oacc[i][e] *= alpha;
        }
      };
      if (kt == 0) {
        move_reference(true);
        psum = exps();
      } else {
        psum = exps();
        if (__any(psum > 256.0f)) {            // redo: scores again from the K fragments (the V^T fragments took their registers)
#pragma unroll
          for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) fr[kb * 4 + ks] = *reinterpret_cast<const bf16x8*>(smem + bofs + koff[ks] + kb * 4096);
#pragma unroll
          for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
              s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[kb * 4 + ks], qf[ks], ks == 0 ? negm : s[kb], 0, 0, 0);
          move_reference(false);
          psum = exps();
#pragma unroll
          for (int sidx = 0; sidx < 4; ++sidx)
#pragma unroll
            for (int db = 0; db < 2; ++db) fr[sidx * 2 + db] = *reinterpret_cast<const bf16x8*>(smem + bofs + koff[sidx] + 8192 + db * 4096);
        }
      }
      l_run += psum;
    } else {
      const float mx_ = col_max();
      const float m_new = fmaxf(m_run, mx_ * c);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      m_run = m_new;
      // (packed v_pk_fma_f32 / v_pk_add_f32 forms of this loop measured 0-3 % SLOWER in same-run A/B: scalar kept)
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
    }
    // ---- O^T += V^T P^T : P fragments straight from the S^T accumulators ----
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        u32x4 pw;
#pragma unroll
        for (int e = 0; e < 4; ++e) pw[e] = pack2(s[kb][8 * s2 + 2 * e], s[kb][8 * s2 + 2 * e + 1]);
        const bf16x8 pf = __builtin_bit_cast(bf16x8, pw);
        const int sidx = 2 * kb + s2;  // 16-key step inside the tile
#if (MX_AEXP & 16)
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
          for (int e = 0; e < 4; ++e) oacc[db][(4 * sidx + e) & 15] += __builtin_bit_cast(float, pw[e] ^ __builtin_bit_cast(u32x4, fr[sidx * 2 + db])[e]);
#else
#pragma unroll
        for (int db = 0; db < 2; ++db)
          oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[sidx * 2 + db], pf, oacc[db], 0, 0, 0);
#endif
      }
    }
    buf = buf == 2 ? 0 : buf + 1;
    bofs = (unsigned)buf * kBufBytes;
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
        u32x2 o = {pack2(oacc[db][4 * g] * inv, oacc[db][4 * g + 1] * inv),
                   pack2(oacc[db][4 * g + 2] * inv, oacc[db][4 * g + 3] * inv)};
        *reinterpret_cast<u32x2*>(op + d) = o;
      }
  }
}


// ----------------------------------------------------------------------------------------------------------------------
// 64 query rows per wave (round 2).  Ablation builds of attn_fwd_dma_kernel (tools/exp/build_attn_variants.sh, MX_AEXP) priced
// its parts at Lk = 4096: the four LDS-DMA pieces a wave issues per tile cost 19 % of the launch (an LDS-DMA instruction holds
// the wave's issue for 60-185 cycles, MI355X_MICROARCH.md), exp2 12 %, the S^T MFMAs 9 %, the barrier 4 %; a lone workgroup per CU
// runs at 57 % of three, i.e. one wave's tile is a serial chain (fragment reads -> S^T -> softmax -> O^T) that other waves must cover.
// Here a wave owns TWO 32-query blocks: the K / V^T fragments it reads from LDS and the DMA pieces it issues serve twice the MFMAs,
// and the two blocks are independent, so block 1's S^T MFMAs sit beside block 0's softmax and block 0's O^T MFMAs beside block 1's
// softmax in ONE instruction stream (sched_group_barrier pins the interleave).  A workgroup = 4 waves = 256 query rows, two
// workgroups per CU; O leaves through the (idle) K / V^T ring as whole 128-byte rows.
// ----------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void attn_fwd64_kernel(const AttnGroup ga) {
  __shared__ __attribute__((aligned(16))) char smem[3 * kBufBytes];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int r = lane & 31;   // query column owned by this lane (and fragment row)
  const int hh = lane >> 5;  // half-wave
  int qb, bh;
  const AttnArgs& p = attn_locate<256>(ga, qb, bh);      // (problem of a grouped launch, query block, batch * H + head)
  if (qb < 0) return;
  const int head = bh % p.H;
  const int b = bh / p.H;
  const int q0 = qb * 256 + wave * 64;

  // ---- Q fragments of the two blocks (B operand of S^T = K Q^T): Q[q0 + 32 blk + r][16*ks + 8*hh .. +7] ----
  bf16x8 qf[2][4];
#pragma unroll
  for (int blk = 0; blk < 2; ++blk) {
    int qi = q0 + blk * 32 + r;
    if (qi > p.Lq - 1) qi = p.Lq - 1;
    const bf16_t* qp = p.q + ((long)b * p.Lq + qi) * p.ldq + head * 64 + hh * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[blk][ks] = *reinterpret_cast<const bf16x8*>(qp + ks * 16);
  }

  // ---- LDS-DMA staging, three-buffer ring, two tiles ahead (layout of attn_fwd_dma_kernel).  Addresses are a wave-uniform
  //      base that moves with the tile plus a per-lane 32-bit offset fixed for the whole kernel: no vector work per tile. ----
  const char* kbase = (const char*)(p.k + (long)b * p.k_bstride + head * 64);
  const char* vbase = (const char*)(p.vt + (long)b * p.vt_bstride + ((long)head * 64) * p.ldvt);
  // A ragged last tile (rem = Lk % 64 keys) is fetched whole: its K rows >= Lk and its all-padding V^T chunks are redirected to
  // addresses inside the operands (row Lk - 1; chunk 0 of the V^T row), and masked after they leave LDS.
  const int rem = p.Lk % KT;
  unsigned voffk[2], voffv[2], toffk[2], toffv[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = i * 256 + tid;
    const int row = c >> 3, ch = (c & 7) ^ ((row >> 1) & 7);
    voffk[i] = (unsigned)(row * p.ldk * 2 + ch * 16);
    voffv[i] = (unsigned)(row * p.ldvt * 2 + ch * 16);
    toffk[i] = (unsigned)((row < rem ? row : rem - 1) * p.ldk * 2 + ch * 16);
    toffv[i] = (unsigned)(row * p.ldvt * 2 + ((ch >> 1) * 16 < rem ? ch * 16 : 0));    // chunk ch = positions 8 ch ..: 16-key group ch >> 1
  }
  // the next tile to issue: uniform pointers advanced per tile (a chunk boundary of the patch-parallel layout jumps them)
  const char* knext = kbase;
  const char* vnext = vbase;
  int chunk_left = p.key_chunk > 0 ? p.key_chunk : 0x7fffffff;
  int chunk_id = 0;
  const unsigned wave_img = (unsigned)__builtin_amdgcn_readfirstlane(wave) * 1024u;
  auto issue_tile = [&](int buf, bool tail) __attribute__((always_inline)) {
    char* img = smem + buf * kBufBytes + wave_img;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(knext + (tail ? toffk[i] : voffk[i])),
                                       (__attribute__((address_space(3))) void*)(img + i * 4096), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vnext + (tail ? toffv[i] : voffv[i])),
                                       (__attribute__((address_space(3))) void*)(img + 8192 + i * 4096), 16, 0, 0);
    }
    knext += (long)KT * p.ldk * 2;
    vnext += KT * 2;
    chunk_left -= KT;
    if (chunk_left == 0) {                     // tiles never straddle a chunk (key_chunk % 64 == 0)
      ++chunk_id;
      chunk_left = p.key_chunk;
      knext = kbase + (long)chunk_id * p.k_cstride * 2;
      vnext = vbase + (long)chunk_id * p.vt_cstride * 2;
    }
  };

  unsigned koff[4];                            // per-lane LDS read offsets inside a buffer (see attn_fwd_kernel)
  {
    const int swz = (r >> 1) & 7;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) koff[ks] = (unsigned)(r * 128 + (((2 * ks + hh) ^ swz) * 16));
  }

  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  f32x16 oacc[2][2] = {{zero16, zero16}, {zero16, zero16}};
  float l_run[2] = {0.f, 0.f};                 // this half-wave's partial row sums
  // the scores leave their MFMA chains as s - m_ref.  Block 0: the chain starts from negm0 = (-m_ref x 16) of the lane's query column (see attn_fwd_dma_kernel).
  // Block 1: the reference k-step (A operand = a K column of ones, B operand = -m_ref of query column r, bf16-representable): a splat for both blocks does not
  // fit the 256 registers of two waves per SIMD (measured with 11 spills: 332 -> 381 us at Lk 4096).
  f32x16 negm0 = zero16;
  const unsigned one_lo = hh == 0 ? 0x3F80u : 0u;
  const bf16x8 kone = __builtin_bit_cast(bf16x8, u32x4{one_lo, 0u, 0u, 0u});
  u32x4 qm1 = {0u, 0u, 0u, 0u};
  float m_ref[2] = {0.f, 0.f};
  auto set_ref = [&](int blk) __attribute__((always_inline)) {
    if (blk == 0) {
#pragma unroll
      for (int e = 0; e < 16; ++e) negm0[e] = -m_ref[0];
    } else {
      qm1[0] = hh == 0 ? (pack2(-m_ref[1], 0.f) & 0xffffu) : 0u;
    }
  };

  const int ntiles = (p.Lk + KT - 1) / KT;     // >= 3 (launcher)

  auto read_k = [&](bf16x8 (&fr)[8], unsigned bofs) __attribute__((always_inline)) {
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) fr[kb * 4 + ks] = *reinterpret_cast<const bf16x8*>(smem + bofs + koff[ks] + kb * 4096);
  };
  auto read_v = [&](bf16x8 (&fr)[8], unsigned bofs) __attribute__((always_inline)) {
#pragma unroll
    for (int sidx = 0; sidx < 4; ++sidx)
#pragma unroll
      for (int db = 0; db < 2; ++db) fr[sidx * 2 + db] = *reinterpret_cast<const bf16x8*>(smem + bofs + koff[sidx] + 8192 + db * 4096);
  };
  auto mask_scores = [&](f32x16 (&s)[2]) __attribute__((always_inline)) {        // ragged last tile: keys >= Lk never count
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int e = 0; e < 16; ++e)
        if (kb * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh >= rem) s[kb][e] = -INFINITY;
  };
  auto mask_v = [&](bf16x8 (&fr)[8]) __attribute__((always_inline)) {            // ... and their V^T entries (any bit pattern) are zero
#pragma unroll
    for (int sidx = 0; sidx < 4; ++sidx)
#pragma unroll
      for (int db = 0; db < 2; ++db) {
        u32x4 w = __builtin_bit_cast(u32x4, fr[sidx * 2 + db]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {          // word j = elements 2j, 2j + 1 = keys k0, k0 + 1
          const int k0 = 16 * sidx + 4 * hh + ((2 * j) & 3) + 8 * (j >> 1);
          if (k0 >= rem) w[j] &= 0xffff0000u;
          if (k0 + 1 >= rem) w[j] &= 0x0000ffffu;
        }
        fr[sidx * 2 + db] = __builtin_bit_cast(bf16x8, w);
      }
  };
  auto qk = [&](f32x16 (&s)[2], const bf16x8 (&fr)[8], int blk) __attribute__((always_inline)) {   // s = (K Q^T) - m_ref
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
        s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[kb * 4 + ks], qf[blk][ks], ks == 0 ? (blk == 0 ? negm0 : zero16) : s[kb], 0, 0, 0);
    if (blk != 0) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
        s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kone, __builtin_bit_cast(bf16x8, qm1), s[kb], 0, 0, 0);
    }
  };
  auto col_max = [&](const f32x16 (&s)[2]) __attribute__((always_inline)) {
    float mx_ = fmaxf(s[0][0], s[1][0]);
#pragma unroll
    for (int e = 1; e < 16; ++e) mx_ = fmaxf(mx_, fmaxf(s[0][e], s[1][e]));
    return max_across_halves(mx_);
  };
  // p = exp2(s) in place, packed to the four B fragments of O^T += V^T P^T; returns this half-wave's sum
  auto exp_pack = [&](f32x16 (&s)[2], u32x4 (&pw)[4]) __attribute__((always_inline)) {
    float ps = 0.f;                            // ONE chain: two would be SLP-packed into v_pk_add_f32, which costs more beside MFMAs
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const float p0 = __builtin_amdgcn_exp2f(s[0][e]);
      const float p1 = __builtin_amdgcn_exp2f(s[1][e]);
      s[0][e] = p0; s[1][e] = p1;
      ps += p0; ps += p1;
    }
#pragma unroll
    for (int sidx = 0; sidx < 4; ++sidx)
#pragma unroll
      for (int e = 0; e < 4; ++e) pw[sidx][e] = pack2(s[sidx >> 1][8 * (sidx & 1) + 2 * e], s[sidx >> 1][8 * (sidx & 1) + 2 * e + 1]);
    return ps;
  };
  // The softmax reference m_ref is lazy (attn_fwd_kernel): it only has to keep p = exp2(s - m_ref) in range.  The loop computes p
  // speculatively and looks at the sums: a score more than 8 above the reference makes its p, hence the lane's sum, exceed 256.
  // Then (rarely) this path redoes the block exactly: scores again from the K fragments, reference raised to the new maximum, p,
  // and O / l rescaled.  No false negatives; a false positive (32 scores averaging > 3 above the reference) costs time only.
  auto redo = [&](f32x16 (&s)[2], u32x4 (&pw)[4], const bf16x8 (&fr)[8], int blk, const bool tail) __attribute__((always_inline)) {
    qk(s, fr, blk);
    if (tail) mask_scores(s);
    const float mx_ = col_max(s);
    float delta = 0.f;
    if (mx_ > 8.0f) {
      const float nr = bf16lo_to_f32(pack2(m_ref[blk] + mx_, 0.f));          // new reference (rounded to bf16, as the former k-step form needed it: same values)
      delta = nr - m_ref[blk];
      m_ref[blk] = nr;
    }
    set_ref(blk);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int e = 0; e < 16; ++e) s[kb][e] -= delta;
    const float alpha = __builtin_amdgcn_exp2f(-delta);
    l_run[blk] *= alpha;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) oacc[blk][i][e] *= alpha;
    return exp_pack(s, pw);
  };

  issue_tile(0, false);
  issue_tile(1, false);
  // ---- the reference starts at the maximum of tile 0 (a whole tile; its scores are computed again by the loop) ----
  asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  {
    bf16x8 fr[8];
    read_k(fr, 0);
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
      f32x16 s[2];
      qk(s, fr, blk);
      m_ref[blk] = bf16lo_to_f32(pack2(col_max(s), 0.f));
      set_ref(blk);
    }
  }
  int buf = 0;
  unsigned bofs = 0;

  // one tile: fragments of buffer `bofs`; TAIL = the ragged last tile
  auto tile = [&](const bool TAIL) __attribute__((always_inline)) {
    bf16x8 fr[8];
    f32x16 s0[2], s1[2];
    u32x4 pw0[4], pw1[4];
    read_k(fr, bofs);
    qk(s0, fr, 0);                             // S^T of block 0
    if (TAIL) mask_scores(s0);
    __builtin_amdgcn_sched_barrier(0);
    qk(s1, fr, 1);                             // S^T of block 1 beside the exponentials of block 0
    float ps = exp_pack(s0, pw0);
    __builtin_amdgcn_sched_barrier(0);
    if (TAIL) mask_scores(s1);
    if (__any(ps > 256.0f)) {
      ps = redo(s0, pw0, fr, 0, TAIL);
    }
    l_run[0] += ps;
    read_v(fr, bofs);                          // V^T fragments (the K fragments are dead: same registers)
    if (TAIL) mask_v(fr);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int sidx = 0; sidx < 4; ++sidx)       // O^T of block 0 beside the exponentials of block 1
#pragma unroll
      for (int db = 0; db < 2; ++db)
        oacc[0][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[sidx * 2 + db], __builtin_bit_cast(bf16x8, pw0[sidx]), oacc[0][db], 0, 0, 0);
    ps = exp_pack(s1, pw1);
    __builtin_amdgcn_sched_barrier(0);
    if (__any(ps > 256.0f)) {
      read_k(fr, bofs);
      ps = redo(s1, pw1, fr, 1, TAIL);
      read_v(fr, bofs);
      if (TAIL) mask_v(fr);
    }
    l_run[1] += ps;
#pragma unroll
    for (int sidx = 0; sidx < 4; ++sidx)       // O^T of block 1
#pragma unroll
      for (int db = 0; db < 2; ++db)
        oacc[1][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[sidx * 2 + db], __builtin_bit_cast(bf16x8, pw1[sidx]), oacc[1][db], 0, 0, 0);
  };

  const int nfull = rem ? ntiles - 1 : ntiles;  // whole tiles
  for (int kt = 0; kt < nfull; ++kt) {
    if (kt > 0) {
      if (kt + 1 < ntiles) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();            // tile kt has landed for every thread; every wave has left the buffer of tile kt - 1
      asm volatile("" ::: "memory");
    }
    if (kt + 2 < ntiles) issue_tile(buf == 0 ? 2 : buf - 1, rem && kt + 3 == ntiles);   // tile kt + 2 into buffer (kt + 2) % 3
    tile(false);
    buf = buf == 2 ? 0 : buf + 1;
    bofs = (unsigned)buf * kBufBytes;
  }
  if (rem) {                                   // the ragged last tile (never tile 0)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    tile(true);
  }

  // ---- normalise; O[q][d] through the wave's 8 KB of the ring (rows = queries, 128 B, 16-byte chunk ^= row & 7) so that the
  //      global stores are whole rows.  Lane (r, hh) holds d = 32 db + 8 g + 4 hh + {0..3} of query 32 blk + r. ----
  __syncthreads();                             // every wave has left the last tile
  char* patch = smem + wave * 8192;
#pragma unroll
  for (int blk = 0; blk < 2; ++blk) {
    const float inv = 1.0f / (l_run[blk] + __shfl_xor(l_run[blk], 32, 64));
    const int row = blk * 32 + r;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const u32x2 o = {pack2(oacc[blk][db][4 * g] * inv, oacc[blk][db][4 * g + 1] * inv),
                         pack2(oacc[blk][db][4 * g + 2] * inv, oacc[blk][db][4 * g + 3] * inv)};
        const int chunk = (4 * db + g) ^ (row & 7);
        *reinterpret_cast<u32x2*>(patch + row * 128 + chunk * 16 + hh * 8) = o;
      }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int row = (lane >> 3) + 8 * i;
    const int ch = lane & 7;
    const u32x4 o = *reinterpret_cast<const u32x4*>(patch + row * 128 + ((ch ^ (row & 7)) * 16));
    const int qi = q0 + row;
    if (qi < p.Lq) *reinterpret_cast<u32x4*>(p.o + ((long)b * p.Lq + qi) * p.ldo + head * 64 + ch * 8) = o;
  }
}


// ----------------------------------------------------------------------------------------------------------------------
// Cross-attention with a SHORT key sequence (Lk <= 96: the 77 text tokens of SDXL; PatchCrossAttention.forward,
// modules/attention.py:59-110).  Half of the attention launches of a UNet step (70 of 140) are this shape, and the general
// kernel above spends them on machinery they do not need: two 64-key LDS tiles (the second 13/64 valid), a barrier per
// tile, the online-softmax rescale, and 8-byte output stores.  Here
//   * every wave works alone (no LDS tiles, no barrier): the head's K rows and V^T rows live in REGISTERS as ready-made MFMA
//     fragments (<= 12 + 12 fragments, loaded once per wave from L2) and are reused for all of the wave's query blocks;
//   * keys are padded to the next multiple of 32 for S^T = K Q^T (3 blocks for 77 keys) and of 16 for O^T += V^T P^T (5 steps);
//   * single-pass softmax: all scores of a query are in the lane (<= 48 per half-wave), so one max, one exp2 pass, one sum;
//   * the output block is transposed through a wave-private 4-KB LDS patch, so O is written as full 128-byte rows (8 lanes x
//     16 B; the per-CU store rate of full lines is 2.7x that of 8-byte pieces: profiles/r02_a_*), and the next query block's
//     Q rows are fetched while the current one computes.
// The launch is bound by its HBM traffic (Q read + O write); K / V^T stay in L2.
// ----------------------------------------------------------------------------------------------------------------------
template <bool PRE, int LK = 0>
__global__ __launch_bounds__(256, 2) void attn_cross_kernel(const AttnGroup ga) {
  __shared__ __attribute__((aligned(16))) char smem[4 * 4096];       // one 32 x 64 bf16 patch per wave
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  int qb, bh;
  const AttnArgs& p = attn_locate<0>(ga, qb, bh);      // (problem of a grouped launch, workgroup of the pair, batch * H + head)
  if (qb < 0) return;
  // the pair's 32-query blocks are dealt to its xq_wpb waves, the first `rem` taking one more: every wave of the launch works on base or base + 1 blocks
  const int w = __builtin_amdgcn_readfirstlane(qb * 4 + wave);
  if (w >= p.xq_wpb) return;                   // (waves work alone: no barrier in this kernel)
  const int nb = (p.Lq + 31) >> 5;
  const int base = nb / p.xq_wpb, rem = nb - base * p.xq_wpb;
  const int first = w * base + (w < rem ? w : rem);
  attn_cross_wave<PRE, LK>(p, bh / p.H, bh % p.H, first * 32, base + (w < rem ? 1 : 0), smem + wave * 4096, lane);
}

}  // namespace mx

// validate one problem and fill its argument block
static int fill_attention_args(mx::AttnArgs& a, const void* q, int ldq, const void* k, int ldk, const void* vt, int ldvt, int64_t vt_batch_stride, void* o,
                               int ldo, int B, int H, int Lq, int Lk, float scale, bool pre, int key_chunk, int64_t k_bstride, int64_t k_cstride,
                               int64_t vt_cstride, bool causal, const float* bias, int ldb) {
  using namespace mx;
  MX_CHECK(q && k && vt && o, "attention: null operand");
  MX_CHECK(B > 0 && H > 0 && Lq > 0 && Lk > 0, "attention: empty problem");
  MX_CHECK((((uintptr_t)q | (uintptr_t)k | (uintptr_t)vt | (uintptr_t)o) & 15) == 0, "attention: operand pointers must be 16-byte aligned");
  MX_CHECK(ldq % 8 == 0 && ldk % 8 == 0 && ldvt % 8 == 0 && ldo % 4 == 0, "attention: strides must be multiples of 8 elements");
  MX_CHECK(ldq >= H * 64 && ldk >= H * 64 && ldo >= H * 64, "attention: row stride smaller than H*64");
  if (key_chunk > 0) {
    MX_CHECK(key_chunk % 64 == 0 && Lk % key_chunk == 0, "attention: key_chunk must be a multiple of 64 and divide Lk");
    MX_CHECK(ldvt >= key_chunk && k_bstride % 8 == 0 && k_cstride % 8 == 0 && vt_cstride % 8 == 0, "attention: bad chunk strides");
  } else {
    MX_CHECK(ldvt >= MX_VT_LD(Lk), "attention: ldvt must cover MX_VT_LD(Lk) (keys are stored in MX_VT_POS order)");
  }
  MX_CHECK(vt_batch_stride % 8 == 0 && vt_batch_stride >= (int64_t)H * 64 * ldvt, "attention: bad vt_batch_stride");
  a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.vt = (const bf16_t*)vt; a.o = (bf16_t*)o;
  a.vt_bstride = (long)vt_batch_stride;
  a.ldq = ldq; a.ldk = ldk; a.ldvt = ldvt; a.ldo = ldo; a.B = B; a.H = H; a.Lq = Lq; a.Lk = Lk;
  a.scale_log2 = scale * 1.4426950408889634f;
  a.causal = causal ? 1 : 0;
  a.bias = bias; a.ldb = ldb;
  if (bias) MX_CHECK(pre && key_chunk == 0 && ldb % 4 == 0 && ldb >= (Lk + KT - 1) / KT * KT && ((uintptr_t)bias & 15) == 0,
                     "attention: bias form needs prescaled q, ldb a multiple of 4 covering whole 64-key tiles, 16-byte alignment");
  if (causal) MX_CHECK(Lq == Lk && key_chunk == 0 && Lk <= 4096, "attention: causal form is for Lq == Lk <= 4096, one key range");
  a.key_chunk = key_chunk; a.k_bstride = key_chunk > 0 ? (long)k_bstride : (long)Lk * ldk; a.k_cstride = (long)k_cstride; a.vt_cstride = (long)vt_cstride;
  a.xcd_map = ((B * H) % 8 == 0) ? 1 : 0;
  a.xq_wpb = 0;
  return 0;
}

// One launch over ga.n problems (ga.g filled).  The kernel is chosen for the launch as a whole -- by its longest query sequence -- and must
// be able to serve every problem; otherwise the general register-staged kernel takes them all.
static int launch_attention_group(void* stream, mx::AttnGroup& ga, bool pre, bool force_cross = false) {
  using namespace mx;
  const int n = ga.n;
  int maxLq = 0;
  bool all_short = true, all_long_k = true, whole_tiles = true, plain = true, o8 = true;
  double flops = 0, bytes = 0;
  for (int i = 0; i < n; ++i) {
    const AttnArgs& a = ga.g[i];
    maxLq = std::max(maxLq, a.Lq);
    all_short = all_short && a.Lk <= 32 * XK_MAXBLK;
    all_long_k = all_long_k && a.Lk > 2 * KT;
    whole_tiles = whole_tiles && a.Lk % KT == 0 && a.Lk >= 3 * KT;
    plain = plain && a.key_chunk == 0 && !a.causal && !a.bias;
    o8 = o8 && a.ldo % 8 == 0;
    flops += 4.0 * a.B * a.H * (double)a.Lq * a.Lk * 64.0;
    bytes += 2.0 * a.B * a.H * 64.0 * (2.0 * a.Lq + 2.0 * a.Lk);
  }
  bool extra = false;
  for (int i = 0; i < n; ++i) extra = extra || ga.g[i].causal || ga.g[i].bias;
  MX_CHECK(!extra || n == 1, "attention: the causal / bias forms are not grouped");
  // (at Lq 1024 the general kernel is 7 % faster than the short-key one: both are latency-bound)
  enum { K_GENERAL, K_CROSS, K_W64, K_DMA } kind = K_GENERAL;
  if (force_cross) MX_CHECK(all_short && plain && o8, "attention: the short-key kernel needs Lk <= 96, no mask / bias / key chunks and ldo % 8 == 0");
  bool all77 = true;                           // the text encoders' 77 keys: the short-key kernel has a compile-time form for them (attn_cross_body.h, LK), which beats the
  for (int i = 0; i < n; ++i) all77 = all77 && ga.g[i].Lk == 77;      // general kernel at every query length (round 5: 15.7 vs 18.8 us at B8 H20 Lq 1024, profiles/r05_l_cross77_ab.txt)
  if (all_short && (maxLq >= 2048 || force_cross || (all77 && pre)) && plain && o8) kind = K_CROSS;          // short key sequence: every wave keeps K / V^T in registers
  else if (extra) kind = K_GENERAL;                                       // the masked / biased forms live in the register-staged kernel
  else if (pre && all_long_k && maxLq >= 2048 && o8) kind = K_W64;        // 64 query rows per wave (a tie with the 32-row kernels at Lq 1024)
  else if (whole_tiles) kind = K_DMA;                                     // whole tiles: LDS-DMA staging two tiles ahead
  const int rows = kind == K_W64 ? 256 : 128;
  if (kind == K_CROSS) {
    // The short-key kernel is latency-bound: a wave pays ~3 us to fetch its head's 24 fragments and ~2.5 us per 32-query block, two workgroups per CU.  With a fixed
    // 64 queries per wave the step's shapes were 1.25 rounds (B8 H20 Lq1024: 640 workgroups on 512 slots) or 2.5 (Lq4096).  Deal the blocks instead: every wave of
    // a (batch, head) takes `per` or `per - 1` consecutive blocks, `per` chosen to minimise rounds x (3 + 2.5 per) -- one round whenever the pairs fit the chip.
    const long slots = 2L * cu_count();
    int best_per = 1; double best_t = 0;
    int max_nb = 1;
    for (int i = 0; i < n; ++i) max_nb = std::max(max_nb, cdiv(ga.g[i].Lq, 32));
    for (int per = 1; per <= max_nb; ++per) {
      long wgs = 0;
      for (int i = 0; i < n; ++i) wgs += (long)cdiv(cdiv(cdiv(ga.g[i].Lq, 32), per), 4) * ga.g[i].H * ga.g[i].B;
      const double t = (double)cdiv64(wgs, slots) * (3.0 + 2.5 * per);
      if (best_t == 0 || t < best_t - 1e-9) { best_t = t; best_per = per; }
      if (wgs <= slots) break;                 // one round: a larger share only lengthens it
    }
    static const bool fixed64 = [] { const char* e = getenv("MX_XQ_FIXED"); return e && e[0] == '1'; }();      // A/B: the former 64 queries per wave
    if (fixed64) best_per = 2;
    for (int i = 0; i < n; ++i) ga.g[i].xq_wpb = cdiv(cdiv(ga.g[i].Lq, 32), best_per);
  }
  long blocks = 0;
  for (int i = 0; i < n; ++i) {
    blocks = (blocks + 7) & ~7L;
    ga.blk0[i] = (int)blocks;
    ga.nblk[i] = (kind == K_CROSS ? cdiv(ga.g[i].xq_wpb, 4) : cdiv(ga.g[i].Lq, rows)) * ga.g[i].H * ga.g[i].B;
    blocks += ga.nblk[i];
  }
  for (int i = n; i < MX_MAX_SEGS; ++i) ga.nblk[i] = 0;
  for (int i = n; i <= MX_MAX_SEGS; ++i) ga.blk0[i] = (int)blocks;
  MX_CHECK(blocks < 2147483647L, "attention: too many workgroups");
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)blocks), block(256);
  const AttnArgs& a0 = ga.g[0];
  prof_begin(st, kind == K_CROSS ? PROF_ATTN_CROSS : PROF_ATTN, flops, bytes, n == 1 ? a0.B * a0.H : n, maxLq, a0.Lk);
  switch (kind) {
    case K_CROSS: {
      if (pre && all77) hipLaunchKernelGGL((attn_cross_kernel<true, 77>), grid, block, 0, st, ga);
      else if (pre) hipLaunchKernelGGL(attn_cross_kernel<true>, grid, block, 0, st, ga);
      else hipLaunchKernelGGL(attn_cross_kernel<false>, grid, block, 0, st, ga);
      break;
    }
    case K_W64: hipLaunchKernelGGL(attn_fwd64_kernel, grid, block, 0, st, ga); break;
    case K_DMA:
      if (pre) hipLaunchKernelGGL(attn_fwd_dma_kernel<true>, grid, block, 0, st, ga); else hipLaunchKernelGGL(attn_fwd_dma_kernel<false>, grid, block, 0, st, ga);
      break;
    default:
      if (extra) { if (pre) hipLaunchKernelGGL((attn_fwd_kernel<true, true>), grid, block, 0, st, ga); else hipLaunchKernelGGL((attn_fwd_kernel<false, true>), grid, block, 0, st, ga); }
      else if (pre) hipLaunchKernelGGL(attn_fwd_kernel<true>, grid, block, 0, st, ga);
      else hipLaunchKernelGGL(attn_fwd_kernel<false>, grid, block, 0, st, ga);
  }
  prof_end(st);
  MX_LAUNCH_CHECK();
  return 0;
}

static int launch_attention(void* stream, const void* q, int ldq, const void* k, int ldk, const void* vt, int ldvt,
                            int64_t vt_batch_stride, void* o, int ldo, int B, int H, int Lq, int Lk, float scale, bool pre,
                            int key_chunk = 0, int64_t k_bstride = 0, int64_t k_cstride = 0, int64_t vt_cstride = 0, bool causal = false,
                            const float* bias = nullptr, int ldb = 0) {
  mx::AttnGroup ga;
  ga.n = 1;
  if (fill_attention_args(ga.g[0], q, ldq, k, ldk, vt, ldvt, vt_batch_stride, o, ldo, B, H, Lq, Lk, scale, pre, key_chunk, k_bstride, k_cstride, vt_cstride,
                          causal, bias, ldb)) return 1;
  return launch_attention_group(stream, ga, pre);
}

/* Grouped form of mx_attention_prescaled: the problems of all resolutions of a mixed batch in ONE launch (the reference regroups the patches of
 * every latent per resolution and calls the attention once per resolution: modules/attention.py:152-203). */
extern "C" int mx_attention_prescaled_grouped(void* stream, const mx_attn_problem* probs, int n, int ldq, int ldk, int ldo, int H) {
  MX_CHECK(probs != nullptr && n >= 1 && n <= MX_MAX_SEGS, "attention: grouped launch needs 1..MX_MAX_SEGS problems");
  mx::AttnGroup ga;
  ga.n = n;
  for (int i = 0; i < n; ++i) {
    const mx_attn_problem& p = probs[i];
    if (fill_attention_args(ga.g[i], p.q, ldq, p.k, ldk, p.vt, p.ldvt, p.vt_batch_stride, p.o, ldo, p.B, H, p.Lq, p.Lk, 1.0f, true, 0, 0, 0, 0, false, nullptr, 0))
      return 1;
  }
  return launch_attention_group(stream, ga, true);
}

extern "C" int mx_attention(void* stream, const void* q, int ldq, const void* k, int ldk, const void* vt, int ldvt,
                            int64_t vt_batch_stride, void* o, int ldo, int B, int H, int Lq, int Lk, float scale) {
  return launch_attention(stream, q, ldq, k, ldk, vt, ldvt, vt_batch_stride, o, ldo, B, H, Lq, Lk, scale, false);
}

extern "C" int mx_attention_prescaled(void* stream, const void* q, int ldq, const void* k, int ldk, const void* vt, int ldvt,
                                      int64_t vt_batch_stride, void* o, int ldo, int B, int H, int Lq, int Lk) {
  return launch_attention(stream, q, ldq, k, ldk, vt, ldvt, vt_batch_stride, o, ldo, B, H, Lq, Lk, 1.0f, true);
}

/* mx_attention_prescaled through the short-key kernel whatever Lq: the separate-launch form of stage 2 of mx_attn_tail (attn_tail.hip) */
extern "C" int mx_attention_cross_prescaled(void* stream, const void* q, int ldq, const void* k, int ldk, const void* vt, int ldvt,
                                            int64_t vt_batch_stride, void* o, int ldo, int B, int H, int Lq, int Lk) {
  mx::AttnGroup ga;
  ga.n = 1;
  if (fill_attention_args(ga.g[0], q, ldq, k, ldk, vt, ldvt, vt_batch_stride, o, ldo, B, H, Lq, Lk, 1.0f, true, 0, 0, 0, 0, false, nullptr, 0)) return 1;
  return launch_attention_group(stream, ga, true, /*force_cross=*/true);
}

/* causal self-attention of a short sequence (the CLIP text encoders: 77 tokens), q prescaled as for mx_attention_prescaled */
extern "C" int mx_attention_prescaled_causal(void* stream, const void* q, int ldq, const void* k, int ldk, const void* vt, int ldvt,
                                             int64_t vt_batch_stride, void* o, int ldo, int B, int H, int L) {
  return launch_attention(stream, q, ldq, k, ldk, vt, ldvt, vt_batch_stride, o, ldo, B, H, L, L, 1.0f, true, 0, 0, 0, 0, true);
}

/* additive score bias (T5 relative position bias): softmax(q k^T + bias[h]) v with q AND bias already in the log2 domain (both multiplied by
 * log2(e) by their producers); bias fp32 [H][Lq][ldb], ldb >= Lk rounded up to 64 */
extern "C" int mx_attention_prescaled_bias(void* stream, const void* q, int ldq, const void* k, int ldk, const void* vt, int ldvt,
                                           int64_t vt_batch_stride, void* o, int ldo, int B, int H, int Lq, int Lk, const float* bias, int ldb) {
  MX_CHECK(bias != nullptr, "attention: null bias");
  return launch_attention(stream, q, ldq, k, ldk, vt, ldvt, vt_batch_stride, o, ldo, B, H, Lq, Lk, 1.0f, true, 0, 0, 0, 0, false, bias, ldb);
}

/* patch-parallel form (mx_unet_forward_pp): K rows and V^T columns of the `world` ranks arrive rank-major from the all-gather.
 * keys [c * key_chunk, (c + 1) * key_chunk) of batch b: K rows at k + c * k_chunk_stride + b * k_batch_stride (row stride ldk),
 * V^T at vt + c * vt_chunk_stride + b * vt_batch_stride + (h * 64 + d) * ldvt + MX_VT_POS(key - c * key_chunk). */
extern "C" int mx_attention_prescaled_chunked(void* stream, const void* q, int ldq, const void* k, int ldk, const void* vt, int ldvt,
                                              int64_t vt_batch_stride, void* o, int ldo, int B, int H, int Lq, int Lk, int key_chunk,
                                              int64_t k_batch_stride, int64_t k_chunk_stride, int64_t vt_chunk_stride) {
  return launch_attention(stream, q, ldq, k, ldk, vt, ldvt, vt_batch_stride, o, ldo, B, H, Lq, Lk, 1.0f, true, key_chunk,
                          k_batch_stride, k_chunk_stride, vt_chunk_stride);
}
