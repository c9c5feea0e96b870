// Shared argument block and fused epilogue of the gfx950 GEMM / implicit-GEMM kernels (gemm_bf16.hip: generic
// 128-row tile; gemm_bf16_v2.hip: 256-row pipelined tile).  See gemm_bf16.hip for the orientation: the accumulator
// block acc[i][j] of v_mfma_f32_16x16x32_bf16 holds, per lane, features n = n_wave0 + 16 i + 4 (lane>>4) + {0..3}
// of token m = m_wave0 + 16 j + (lane & 15).
#pragma once
#include "common.h"
#include "../../include/mxdenoise.h"

namespace mx {

// one problem of a grouped launch (mx_gemm_seg + the first m-tile index of the problem in the launch's tile list)
constexpr int kMaxSegs = MX_MAX_SEGS;
struct GemmSeg {
  const bf16_t* a; const bf16_t* a2; void* c; const bf16_t* residual; bf16_t* vt; const float* rowbias; const float* gate;
  const float* ln_stats; float* stats_out;
  int M, tile0, rows_per_batch, ldvt;
  int B, Hin, Win, Hout, Wout;
  int a_batch_rows, a_row_off, c_batch_rows, c_row_off;
};

struct GemmArgs {
  const bf16_t* a;
  const bf16_t* w;
  void* c;
  const float* bias;
  const float* rowbias;
  const bf16_t* residual;
  bf16_t* vt;
  int M, N, K;
  int lda, ldc, ldr, ldrb;
  int rows_per_batch;
  int flags;
  int seg, period, ldvt;
  int B, Hin, Win, Cin, Hout, Wout, stride, up, corner_patch;
  int a_batch_rows, a_row_off, c_batch_rows, c_row_off;
  const float* gate;
  int ldg;
  int xcd_map;   // 1: XCD-aware workgroup -> tile map (gemm_tile_of_block)
  float out_scale;   // != 0: (acc + bias) * out_scale (QKV: q segments only)
  const float* rms_wq;   // MX_EPI_RMSNORM: per-head RMSNorm weights of the q / k segments
  const float* rms_wk;
  float rms_eps;
  int stagger_ticks;   // experiment (gemm_bf16_v3.hip)
  int vhalo;           // conv: images stored with one halo row above and below (patch-parallel)
  const bf16_t* a2;    // GEMM: columns [k_split, K) of the A operand come from a2 (row stride lda2); nullptr = one source
  int lda2, k_split;
  const float* ln_stats;   // folded LayerNorm (mxdenoise.h): (sum, sum of squares) per row and slab; nullptr = off
  const float* ln_colsum;
  int ln_slabs;
  float ln_eps;
  float* stats_out;        // row statistics of the stored values, one slab per wave column panel; nullptr = off
  // finalised row statistics (mxdenoise.h, ln_final): consumer side (the 256 x 256 kernel) / producer side (256-row tiles) + its panel tickets
  const float* ln_final;
  float* ln_final_out;
  unsigned* ln_final_cnt;
  int ln_final_slabs;      // slabs the producer's own launch writes per row (N / wave panel)
  float* gn_part;          // GroupNorm partial sums per 64 output rows and channel (mxdenoise.h gn_part_out); nullptr = off
  // split-K (128-row tiles, small launches: gemm_bf16_v2.hip): the K tiles of an output tile are dealt to `splitk` workgroups; each leaves its
  // fp32 partial tile in sk_ws and takes a ticket from sk_cnt[tile]; the last one sums the partials in slice order and runs the epilogue
  int splitk;
  float* sk_ws;
  unsigned* sk_cnt;
  // grouped launch (mx_gemm_desc.segs): nseg problems along M, prob[i].tile0 = first m-tile of problem i, mt_total = all m-tiles; nseg == 0: M above
  int nseg, mt_total;
  GemmSeg prob[kMaxSegs];
};

// Grouped launch: m-tile `tm` of the whole launch -> its problem.  `q` (a copy of the kernel argument) becomes that problem -- rows, per-batch
// structure, operand bases -- and tm the tile index inside it.  Everything here is wave-uniform (scalar registers / scalar loads from the
// kernel-argument segment); the kernel body that follows is the ordinary one.  Returns the problem index.
__device__ __forceinline__ int gemm_select_seg(GemmArgs& q, const GemmArgs& p, int& tm) {
  if (p.nseg <= 0) return 0;
  int s = 0;
#pragma unroll
  for (int i = 1; i < kMaxSegs; ++i) if (i < p.nseg && tm >= p.prob[i].tile0) s = i;
  const GemmSeg& g = p.prob[s];
  tm -= g.tile0;
  q.a = g.a; q.a2 = g.a2; q.c = g.c; q.residual = g.residual; q.vt = g.vt; q.rowbias = g.rowbias; q.gate = g.gate;
  q.ln_stats = g.ln_stats; q.stats_out = g.stats_out;
  q.M = g.M; q.rows_per_batch = g.rows_per_batch; q.ldvt = g.ldvt;
  q.B = g.B; q.Hin = g.Hin; q.Win = g.Win; q.Hout = g.Hout; q.Wout = g.Wout;
  q.a_batch_rows = g.a_batch_rows; q.a_row_off = g.a_row_off; q.c_batch_rows = g.c_batch_rows; q.c_row_off = g.c_row_off;
  return s;
}
// m-tiles of the whole launch for tiles of `rows` rows
__device__ __forceinline__ int gemm_m_tiles(const GemmArgs& p, int rows) { return p.nseg > 0 ? p.mt_total : (p.M + rows - 1) / rows; }

// a * s + c on four lanes' worth of values as explicit fused multiply-adds (the functions below switch the compiler's own contraction off)
__device__ __forceinline__ f32x4 fma4(const f32x4 a, const float s, const f32x4 c) {
  return f32x4{fmaf(a[0], s, c[0]), fmaf(a[1], s, c[1]), fmaf(a[2], s, c[2]), fmaf(a[3], s, c[3])};
}

// sum over the four lanes of a token (lane bits 4 and 5) without the LDS crossbar: v_permlane16_swap / v_permlane32_swap exchange
// 16-lane rows / wave halves between two registers (inline asm: the builtins fold their two results when both inputs are one value)
__device__ __forceinline__ float sum_over_fq(float x) {
  float a = x, b = x;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  a += b; b = a;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return a + b;
}

// Folded LayerNorm in the 256 / 128-row kernels, accumulator side.  Before the K loop the accumulators start at -mean_m * colsum_n, so
// the loop leaves x W'^T - mean colsum and the epilogue only multiplies by rstd_m (returned; 1 without the fold) before the bias: the
// statistics and column sums are fetched while the first operand tiles are in flight, and the epilogue carries no extra vectors.
// (First version: everything in the epilogue -- eight serial round trips to the statistics per 256 x 256 tile and 60 spilled registers
// cost 37 us per launch.)  The four lanes of a token split the slabs; the slab loop is the OUTER loop so that each round has MI
// independent loads in flight.
template <int NI, int MI>
__device__ __forceinline__ void gemm_ln_init(const GemmArgs& p, f32x4 (&acc)[NI][MI], const int m_wave0, const int wave_n0, const int fr,
                                             const int fq, float (&rstd)[MI]) {
#pragma clang fp contract(off)      // (round 5) every fused multiply-add below is written out: two compilations of this function -- the stand-alone kernels and the chained
                                    // launch of attn_tail.hip -- must round identically, and the contraction the compiler picks depends on the code around it
  // statistics layout: row m holds its slabs side by side, ln_stats[(m * pitch + slab) * 2 + {0, 1}], pitch = slabs rounded up to 4: lane
  // (token, fq) reads slabs 4 fq .. 4 fq + 3 (of every group of 16) as two 16-byte loads -- one round trip for up to 16 slabs
  const int slabs = p.ln_slabs;
  const int pitch = (slabs + 3) & ~3;
  float s1[MI], s2[MI];
#pragma unroll
  for (int j = 0; j < MI; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
  for (int g = 4 * fq; g < slabs; g += 16) {
    constexpr int JC = MI > 4 ? 4 : MI;        // token blocks per round (registers: 8 per block in flight)
#pragma unroll
    for (int j0 = 0; j0 < MI; j0 += JC) {
      f32x4 va[JC], vb[JC];
#pragma unroll
      for (int j = 0; j < JC; ++j) {
        const int m = m_wave0 + (j0 + j) * 16 + fr;
        const float* src = p.ln_stats + ((long)(m < p.M ? m : p.M - 1) * pitch + g) * 2;
        va[j] = *reinterpret_cast<const f32x4*>(src);
        vb[j] = *reinterpret_cast<const f32x4*>(src + 4);
      }
#pragma unroll
      for (int j = 0; j < JC; ++j) {           // entries past the last slab were never written
        s1[j0 + j] += va[j][0] + (g + 1 < slabs ? va[j][2] : 0.f) + (g + 2 < slabs ? vb[j][0] : 0.f) + (g + 3 < slabs ? vb[j][2] : 0.f);
        s2[j0 + j] += va[j][1] + (g + 1 < slabs ? va[j][3] : 0.f) + (g + 2 < slabs ? vb[j][1] : 0.f) + (g + 3 < slabs ? vb[j][3] : 0.f);
      }
    }
  }
  const float inv = 1.0f / (float)p.K;
  float nmean[MI];
#pragma unroll
  for (int j = 0; j < MI; ++j) {
    const float a = sum_over_fq(s1[j]), b = sum_over_fq(s2[j]);
    const float mean = a * inv;
    const float var = fmaxf(fmaf(-mean, mean, b * inv), 0.f);
    rstd[j] = rsqrtf(var + p.ln_eps);
    nmean[j] = -mean;
  }
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int n = wave_n0 + i * 16 + fq * 4;
    const f32x4 cs = n < p.N ? *reinterpret_cast<const f32x4*>(p.ln_colsum + n) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < MI; ++j) acc[i][j] = cs * nmean[j];
  }
}

// one row: (rstd, rstd * mean) of row m from the slabs of partial sums.  Called by all 64 lanes; the four lanes of a token
// (lane & 15 equal) split the slabs and combine by shuffles.
__device__ __forceinline__ void gemm_ln_row(const GemmArgs& p, const int m, const int fq, float& rstd, float& rm) {
#pragma clang fp contract(off)
  const int mc = m < p.M ? m : p.M - 1;
  const int pitch = (p.ln_slabs + 3) & ~3;
  float s1 = 0.f, s2 = 0.f;
  for (int s = fq; s < p.ln_slabs; s += 4) {
    const f32x2 v = *reinterpret_cast<const f32x2*>(p.ln_stats + ((long)mc * pitch + s) * 2);
    s1 += v[0]; s2 += v[1];
  }
  s1 = sum_over_fq(s1); s2 = sum_over_fq(s2);
  const float inv = 1.0f / (float)p.K;
  const float mean = s1 * inv;
  const float var = fmaxf(fmaf(-mean, mean, s2 * inv), 0.f);
  rstd = rsqrtf(var + p.ln_eps);
  rm = rstd * mean;
}

// ---- GroupNorm partial sums from the producing launch (mxdenoise.h gn_part_out).  Called AFTER the epilogue of a 256-row tile whose epilogue is
// bias (+ per-sample row bias) only: the value a token stores is acc + c with c constant per channel and sample, so the launch leaves the sums of the
// ACCUMULATORS and the GroupNorm's fold adds c back in closed form (sum + 64 c, sum of squares + 2 c sum + 64 c^2, in fp64) -- no load here, nothing
// in front of the epilogue's own loads and stores (a first form summed acc + c before the epilogue: its loads and stores sat in front of the epilogue's
// and cost the conv 7 us per launch).  Lane (token fr, fq) holds features 16 i + 4 fq + {0..3}: sum over its MI token blocks, then over the 16 lanes of
// its DPP row; the stores overlap the drain of the tile's own. ----
__device__ __forceinline__ float row16_sum(float x) {
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true));    // quad_perm [1, 0, 3, 2]
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, true));    // quad_perm [2, 3, 0, 1]
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xF, 0xF, true));   // row_half_mirror
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x140, 0xF, 0xF, true));   // row_mirror
  return x;
}
template <int NI, int MI>
__device__ __forceinline__ void gemm_gn_partials(const GemmArgs& p, const f32x4 (&acc)[NI][MI], const int m_wave0, const int wave_n0, const int fr, const int fq) {
  static_assert(MI == 4, "one 64-token chunk per wave");
  if (m_wave0 >= p.M) return;                  // (M % 64 == 0: a chunk is whole or absent)
  float* dst = p.gn_part + ((long)(m_wave0 >> 6) * p.N + wave_n0 + fq * 4) * 2;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    f32x4 s = {0.f, 0.f, 0.f, 0.f}, q = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < MI; ++j) { s += acc[i][j]; q += acc[i][j] * acc[i][j]; }
#pragma unroll
    for (int e = 0; e < 4; ++e) { s[e] = row16_sum(s[e]); q[e] = row16_sum(q[e]); }
    if (fr == 0) {
      *reinterpret_cast<f32x4*>(dst + i * 32) = f32x4{s[0], q[0], s[1], q[1]};
      *reinterpret_cast<f32x4*>(dst + i * 32 + 4) = f32x4{s[2], q[2], s[3], q[3]};
    }
  }
}

// ---- finalised row statistics (mxdenoise.h, ln_final).  PRODUCER side, after the epilogue of a 256-row tile whose launch has ln_final_out:
// the slabs of this tile's rows went out as write-through (sc1) stores; the workgroup drains them, takes a ticket of its 256-row panel, and the
// LAST of the panel's N / BN workgroups reads the panel's slabs back (sc1 loads), adds them in slab order and leaves (mean, rstd) per row.  No
// fence (an L2 write-back of the whole tile's output) and no acquire: the same hand-off as split-K below.  Every thread of the workgroup calls. ----
__device__ __forceinline__ void gemm_ln_finalize(const GemmArgs& p, const int tm, const int nt, volatile int* ticket_word) {
#pragma clang fp contract(off)
  const int tid = threadIdx.x;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) *ticket_word = (int)__hip_atomic_fetch_add(p.ln_final_cnt + tm, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (*ticket_word != nt - 1) return;
  if (tid == 0) __hip_atomic_store(p.ln_final_cnt + tm, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // zero again for the next launch
  const int slabs = p.ln_final_slabs;
  const int pitch = (slabs + 3) & ~3;
  const int m = tm * 256 + (tid >> 1);         // two threads per row: slabs [0, half) and [half, slabs), each in slab order
  const int half = (slabs + 1) >> 1;
  const int s0 = (tid & 1) ? half : 0, s1 = (tid & 1) ? slabs : half;
  float a = 0.f, q = 0.f;
  if (m < p.M) {
    const unsigned long long* src = reinterpret_cast<const unsigned long long*>(p.stats_out + (long)m * pitch * 2);
    for (int s = s0; s < s1; ++s) {
      const unsigned long long v = __hip_atomic_load(src + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      a += __uint_as_float((unsigned)v); q += __uint_as_float((unsigned)(v >> 32));
    }
  }
  const float a2 = __shfl_xor(a, 1, 64), q2 = __shfl_xor(q, 1, 64);
  if (!(tid & 1) && m < p.M) {
    const float inv = 1.0f / (float)p.N;
    const float mean = (a + a2) * inv;
    const float var = fmaxf(fmaf(-mean, mean, (q + q2) * inv), 0.f);
    *reinterpret_cast<f32x2*>(p.ln_final_out + (long)m * 2) = f32x2{mean, rsqrtf(var + p.ln_eps)};
  }
}

// CONSUMER side (the persistent 256 x 256 kernel): the tile's accumulators start at -mean_m * colsum_n from 8 bytes per row; rstd for the epilogue.
// KEEP: rstd stays in registers across the K loop; otherwise the epilogue fetches it again, one token block ahead (gemm_epilogue_regs
// RSTD_LOAD: the QKV instantiation has no registers to spare)
template <int NI, int MI, bool KEEP>
__device__ __forceinline__ void gemm_ln_init_final(const GemmArgs& p, f32x4 (&acc)[NI][MI], const int m_wave0, const int wave_n0, const int fr,
                                                   const int fq, float (&rstd)[MI]) {
#pragma clang fp contract(off)
  f32x2 st[MI];
#pragma unroll
  for (int j = 0; j < MI; ++j) {
    const int m = m_wave0 + j * 16 + fr;
    st[j] = *reinterpret_cast<const f32x2*>(p.ln_final + (long)(m < p.M ? m : p.M - 1) * 2);
  }
  f32x4 cs[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) cs[i] = *reinterpret_cast<const f32x4*>(p.ln_colsum + wave_n0 + i * 16 + fq * 4);
  if constexpr (KEEP) {
#pragma unroll
    for (int j = 0; j < MI; ++j) rstd[j] = st[j][1];
  }
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < MI; ++j) acc[i][j] = cs[i] * (-st[j][0]);
}

// ---- split-K hand-off (cdna guide, "Projection GEMM at M = 256" item 2 and Guideline 16): partial tiles travel through memory as write-through
// (sc1) 16-byte stores and are read back with sc1 loads, so neither a release fence (an L2 write-back) nor an acquire is needed; the counter is an
// agent-scope atomic. ----
// Both directions go through buffer instructions with aux = 16 (sc1) -- __builtin_amdgcn_raw_buffer_store_b128 / _load_b128: the compiler
// counts and waits for them itself.  (Two hand-written forms failed first: an inline-asm global_store_dwordx4 whose data registers the compiler
// overwrote before the store had read them -- guide section 5.7 item 1, stores -- and inline-asm loads whose destinations were copied before
// any hand-placed s_waitcnt.)
typedef unsigned int u32x4_sk __attribute__((ext_vector_type(4)));
// After the K loop of a split launch.  Returns true in the workgroup that must run the epilogue (acc then holds the sum over all slices, added in
// slice order whichever workgroup arrives last: the result does not depend on timing).  `ticket_word`: one LDS word nobody else uses any more.
template <int NI, int MI>
__device__ __forceinline__ bool splitk_combine(const GemmArgs& p, f32x4 (&acc)[NI][MI], const int tile, const int slice, const int tile_elems,
                                               volatile int* ticket_word) {
  const int tid = threadIdx.x;
  // one descriptor over this TILE's slabs (wave-uniform base: tile comes from blockIdx); per-lane byte offsets in voffset
  const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(p.sk_ws + (long)tile * p.splitk * tile_elems, 0, p.splitk * tile_elems * 4, 0x00020000);
  const int lane_off = tid * 16;
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < MI; ++j)
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_sk, acc[i][j]), rsrc, (slice * tile_elems + (i * MI + j) * 2048) * 4 + lane_off, 0, 16);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // every storing wave drains its stores (and its past-the-end DMAs) ...
  __syncthreads();                                       // ... before ONE lane signals for the workgroup
  if (tid == 0) *ticket_word = (int)__hip_atomic_fetch_add(p.sk_cnt + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  const int ticket = *ticket_word;
  if (ticket != p.splitk - 1) return false;
  if (tid == 0) __hip_atomic_store(p.sk_cnt + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // the counter is zero again for the next launch
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < MI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int s = 0; s < p.splitk; ++s) {                  // a whole partial tile in flight at a time (NI * MI 16-byte pieces per lane)
    f32x4 v[NI][MI];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < MI; ++j)
        v[i][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (s * tile_elems + (i * MI + j) * 2048) * 4 + lane_off, 0, 16));
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < MI; ++j) acc[i][j] += v[i][j];
  }
  return true;
}

// row of the A operand / of the output for logical row m (joint-sequence remap, see mxdenoise.h)
__device__ __forceinline__ long gemm_in_row(const GemmArgs& p, int m) {
  if (p.a_batch_rows <= 0) return m;
  const int b = m / p.rows_per_batch;
  return (long)b * p.a_batch_rows + p.a_row_off + (m - b * p.rows_per_batch);
}
__device__ __forceinline__ long gemm_out_row(const GemmArgs& p, int m, int bidx) {
  if (p.c_batch_rows <= 0) return m;
  return (long)bidx * p.c_batch_rows + p.c_row_off + (m - bidx * p.rows_per_batch);
}

// Register-layout epilogue of the generic 128-row kernel (gemm_bf16.hip); the 256-row kernels use gemm_epilogue_regs below.
template <int NI, int MI, int BN>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& p, f32x4 (&acc)[NI][MI], const int m_wave0,
                                              const int wave_n0, const int fr, const int fq) {
  const int flags = p.flags;
  const bool geglu = (flags & MX_EPI_GEGLU) != 0;
  const bool qkv = (flags & MX_EPI_QKV) != 0;
  // QKV: the wave's feature range lies inside one segment (seg % 64 == 0)
  int seg_idx = 0, seg_grp = 0, seg_pos = 0;
  bool to_vt = false;
  if (qkv) {
    seg_idx = wave_n0 / p.seg;
    seg_grp = seg_idx / p.period;
    seg_pos = seg_idx - seg_grp * p.period;
    to_vt = (seg_pos == p.period - 1);
  }

  // MX_EPI_RMSNORM: the wave's 64 features are one head (BN = 128).  A token's head is spread over the NI = 4 blocks of a
  // lane and the 4 lanes fr, fr+16, fr+32, fr+48.
  const bool rms = qkv && (flags & MX_EPI_RMSNORM) && !to_vt;
  const bool ln = p.ln_stats != nullptr;
#pragma unroll
  for (int j = 0; j < MI; ++j) {
    const int m = m_wave0 + j * 16 + fr;
    float rms_mul = 1.0f;
    float ln_rstd = 1.0f, ln_rm = 0.f;
    if (ln) gemm_ln_row(p, m, fq, ln_rstd, ln_rm);   // before the row mask: the shuffles need every lane
    if (rms) {                                   // before the row mask: the shuffles need every lane
      float ss = 0.f;
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int n = wave_n0 + i * 16 + fq * 4;
#pragma unroll
        for (int q = 0; q < 4; ++q) { const float t = acc[i][j][q] + (p.bias ? p.bias[n + q] : 0.f); ss += t * t; }
      }
      ss = sum_over_fq(ss);
      rms_mul = rsqrtf(ss * (1.0f / 64.0f) + p.rms_eps) * ((seg_pos == 0 && p.out_scale != 0.f) ? p.out_scale : 1.0f);
    }
    if (m >= p.M) continue;
    const int bidx = (p.rows_per_batch > 0) ? (m / p.rows_per_batch) : 0;
    const long orow = gemm_out_row(p, m, bidx);
    const long rrow = (flags & MX_EPI_RES_BCAST) ? (long)(m - bidx * p.rows_per_batch) : orow;
#pragma unroll
    for (int i = 0; i < (NI); ++i) {
      if (geglu && i >= NI / 2) continue;
      const int n = wave_n0 + i * 16 + fq * 4;  // packed feature index of v[0]
      if (n >= p.N) continue;
      float v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = acc[i][j][q];
      if (ln) {
        const f32x4 c4 = *reinterpret_cast<const f32x4*>(p.ln_colsum + n);
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = v[q] * ln_rstd - ln_rm * c4[q];
      }
      if (p.bias) {
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(p.bias + n);
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] += b4[q];
      }
      if (rms) {
        const float* w = (seg_pos == 0 ? p.rms_wq : p.rms_wk) + ((n - seg_idx * p.seg) & 63);
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] *= rms_mul * w[q];
      } else if (p.out_scale != 0.f && (!qkv || seg_pos == 0)) {
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] *= p.out_scale;
      }
      if (geglu) {
        float g[4];
        const int ng = n + (NI / 2) * 16;  // gate blocks follow the hidden blocks inside the wave tile
#pragma unroll
        for (int q = 0; q < 4; ++q) g[q] = acc[i + NI / 2][j][q];
        if (ln) {
          const f32x4 c4 = *reinterpret_cast<const f32x4*>(p.ln_colsum + ng);
#pragma unroll
          for (int q = 0; q < 4; ++q) g[q] = g[q] * ln_rstd - ln_rm * c4[q];
        }
        if (p.bias) {
          const f32x4 b4 = *reinterpret_cast<const f32x4*>(p.bias + ng);
#pragma unroll
          for (int q = 0; q < 4; ++q) g[q] += b4[q];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = v[q] * ((flags & MX_EPI_GEGLU_TANH) ? gelu_tanh_f(g[q]) : gelu_fast(g[q]));
        const int nout = wave_n0 / 2 + i * 16 + fq * 4;
        u32x2 o = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
        *reinterpret_cast<u32x2*>(reinterpret_cast<bf16_t*>(p.c) + (long)m * p.ldc + nout) = o;
        continue;
      }
      if (p.rowbias) {
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(p.rowbias + (long)bidx * p.ldrb + n);
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] += b4[q];
      }
      if (p.gate) {
        const f32x4 g4 = *reinterpret_cast<const f32x4*>(p.gate + (long)bidx * p.ldg + n);
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] *= g4[q];
      }
      if (p.residual) {
        const u32x2 r = *reinterpret_cast<const u32x2*>(p.residual + rrow * p.ldr + n);
        v[0] += bf16lo_to_f32(r[0]); v[1] += bf16hi_to_f32(r[0]);
        v[2] += bf16lo_to_f32(r[1]); v[3] += bf16hi_to_f32(r[1]);
      }
      if (flags & MX_EPI_SILU) {
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = silu_f(v[q]);
      }
      if (flags & MX_EPI_GELU_TANH) {
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = gelu_tanh_f(v[q]);
      }
      if (flags & (MX_EPI_GELU | MX_EPI_QUICK_GELU)) {
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = (flags & MX_EPI_GELU) ? gelu_fast(v[q]) : quick_gelu_f(v[q]);
      }
      if (qkv) {
        const int nin = n - seg_idx * p.seg;  // position inside the segment
        if (to_vt) {
          const int key0 = (p.c_batch_rows > 0 ? p.c_row_off : 0) + m - bidx * p.rows_per_batch;
          const int key = MX_VT_POS(key0);      // attention's V^T key order (mxdenoise.h)
          const int nv = p.N / p.period;
          bf16_t* dst = p.vt + ((long)bidx * nv + (long)seg_grp * p.seg + nin) * p.ldvt + key;
#pragma unroll
          for (int q = 0; q < 4; ++q) dst[(long)q * p.ldvt] = f32_to_bf16(v[q]);
        } else {
          const int ccol = seg_grp * (p.period - 1) * p.seg + seg_pos * p.seg + nin;
          u32x2 o = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
          *reinterpret_cast<u32x2*>(reinterpret_cast<bf16_t*>(p.c) + orow * p.ldc + ccol) = o;
        }
        continue;
      }
      if (flags & MX_EPI_OUT_F32) {
        f32x4 o = {v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.c) + orow * p.ldc + n) = o;
      } else {
        u32x2 o = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
        *reinterpret_cast<u32x2*>(reinterpret_cast<bf16_t*>(p.c) + orow * p.ldc + n) = o;
      }
    }
  }
}


// XCD-aware workgroup -> tile map of the one-tile-per-workgroup kernels.  Consecutive workgroup ids are dealt round-robin to
// the 8 XCDs, each with its own 4 MB L2 and 32 CUs.  With the plain map (m fastest) the 32 tiles an XCD runs at one time
// are 16-32 different token panels x 1-2 feature panels, so its L2 takes in ~18 operand tiles per K step and every token
// panel is also fetched by other XCDs (rocprofv3 FETCH_SIZE: 3x the algorithmic bytes).  Here XCD x owns the token panels
// m = 8 i + x and walks them in bands of MB panels x all feature panels, features slowest inside a band: the 32 concurrent
// tiles form a ~4 x 8 block (12 operand tiles per K step) and a token panel enters exactly one L2.
// Needs mt % 8 == 0 (else the plain map).  Returns (m tile, n tile) of workgroup b.
__device__ __forceinline__ void gemm_tile_of_block(const int b, const int mt, const int nt, const int xcd_map, int& tm, int& tn) {
  constexpr int MB = 4;
  if (!xcd_map || (mt & 7) != 0) { tm = b % mt; tn = b / mt; return; }
  const int xcd = b & 7;
  const int local = b >> 3;                   // index inside the XCD's sequence
  const int mloc = mt >> 3;                   // token panels owned by one XCD
  const int band_tiles = MB * nt;
  const int band = local / band_tiles;
  const int r = local - band * band_tiles;
  const int rows = (mloc - band * MB) < MB ? (mloc - band * MB) : MB;   // the last band may be short
  tn = r / rows;
  tm = ((band * MB + (r - tn * rows)) << 3) + xcd;
}

// ----------------------------------------------------------------------------------------------------------------------
// Register-exchange epilogue of the 256-row kernels (round 2; replaces the LDS-staged epilogue of round 1, see git history).
//
// The staged epilogue above costs ~15 us per 256 x 256 tile (8 slabs x {ds_write, barrier, ds_read, store}: 30 % of a
// K = 1536 launch and > 40 % at the SDXL depths K = 640 / 1280, profiles/r01_e_gemm_component_removal.txt) with the
// matrix pipe idle, and it occupies the LDS ring, so a persistent workgroup cannot keep its operand stream running.
// Here the transpose happens in registers instead.  In the accumulator layout the four lane rows fq = lane >> 4 of a
// 16-token block hold features 4 fq .. 4 fq + 3 of each 16-feature block.  For a PAIR of adjacent blocks (32 features)
// two half-wave exchanges per register,
//     v_permlane32_swap a, b   (lanes 32-63 of a <-> lanes 0-31 of b)
//     v_permlane16_swap a, b   (odd 16-lane rows of a <-> even rows of b),
// leave lane row q with the eight consecutive features 8 q .. 8 q + 7 of the pair: [a, b] of that lane.  A token's 32
// features are then 4 lanes x 16 bytes = 64 contiguous bytes per store instruction (two instructions complete a
// 128-byte line), and the residual is read the same way.  No LDS, no barrier: every wave streams its own rows, waves
// that finish early move on, and the LDS ring keeps receiving the next tile's operands.
//   * order of operations unchanged: bias -> (RMSNorm | out_scale) -> GEGLU | row bias -> gate (accumulator layout),
//     exchange in fp32, residual -> SiLU / GELU-tanh -> one rounding -> store (row layout);
//   * an odd block count (BN = 160: five blocks per wave) stores its last block from the accumulator layout (8 bytes per
//     lane, 32 contiguous bytes per token);
//   * MX_EPI_QKV: waves whose features fall in a V segment write V^T directly (2-byte stores along the key axis).
// ----------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void lane_exchange_pair(float (&a)[4], float (&b)[4]) {
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    unsigned x = __float_as_uint(a[d]), y = __float_as_uint(b[d]);
    auto r = __builtin_amdgcn_permlane32_swap(x, y, false, false);
    auto t = __builtin_amdgcn_permlane16_swap(r[0], r[1], false, false);
    a[d] = __uint_as_float(t[0]);
    b[d] = __uint_as_float(t[1]);
  }
}

// neighbour-lane exchange (lane ^ 1) of a 16-byte value: DPP quad_perm [1, 0, 3, 2]
__device__ __forceinline__ u32x4 lane_xor1(const u32x4 x) {
  u32x4 r;
#pragma unroll
  for (int d = 0; d < 4; ++d) r[d] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)x[d], 0xB1, 0xF, 0xF, true);
  return r;
}

// VPF: per-sample vectors are loaded one token block ahead (costs 8 * NIO registers; off in the persistent 256 x 256 kernel)
// STATS: can write row statistics (stats_out) and apply the folded LayerNorm's rstd (ln_rstd_a); off in the 256 x 256 kernels, whose
// register budget is spent (pick_tile keeps launches with ln_stats / stats_out away from them)
// FEAT: which of the runtime-selected epilogue features are COMPILED IN (the launcher picks the smallest instantiation that serves the
// launch's flags).  Round 3 (tools/exp/timeline_v4.py): with every feature behind a runtime branch the 256 x 256 kernel was 164 KB of code
// -- 2.5 instruction caches -- and a plain bias-only epilogue took 14 us per tile of branching through it.
constexpr int EPI_F_QKV = 1;                  // MX_EPI_QKV (segmented output, V^T, RMSNorm of the heads)
constexpr int EPI_F_ACT = 2;                  // any of MX_EPI_SILU / GELU / GELU_TANH / QUICK_GELU (and GEGLU_TANH in the gated epilogue), chosen at run time
constexpr int EPI_F_ALL = 3;
constexpr int EPI_F_TANH = 4;                 // MX_EPI_GELU_TANH alone (the MMDiT feed-forward): only that activation is compiled in
__host__ __device__ inline int gemm_epi_features(int flags) {
  const int acts = flags & (MX_EPI_SILU | MX_EPI_GELU | MX_EPI_GELU_TANH | MX_EPI_QUICK_GELU | MX_EPI_GEGLU_TANH);
  return ((flags & MX_EPI_QKV) ? EPI_F_QKV : 0) | (acts == 0 ? 0 : acts == MX_EPI_GELU_TANH ? EPI_F_TANH : EPI_F_ACT);
}
// EMIT: stats_out is compiled in (STATS alone: only the folded LayerNorm's rstd -- the 256 x 256 kernel's LN instantiations)
// RSTD_LOAD: the folded LayerNorm's rstd is fetched from p.ln_final one token block ahead instead of arriving in ln_rstd_a (two registers
// instead of MI: the QKV instantiation of the 256 x 256 kernel)
// (Tried in round 4 and dropped: the WHOLE fold in the epilogue -- acc * rstd + (bias - rstd * mean * colsum) from one 8-byte load per token block, nothing in
// front of the K loop.  +7.9 us per GEGLU launch against +4 us for the accumulator start: the block-ahead load does not cover an L2 round trip.)
// WT: the output tile and its row statistics are stored WRITE-THROUGH (sc1: buffer stores with aux 16 over a descriptor of C, 8-byte agent-scope atomic
// stores for the statistics) -- a chained launch (attn_tail.hip) hands them to other workgroups of the same launch (cdna guide, Guideline 16 R1); the
// values stored are the same
template <int NI, int MI, bool GEGLU, bool VEC = true, bool VPF = true, bool STATS = true, int FEAT = EPI_F_ALL, bool EMIT = STATS, bool RSTD_LOAD = false, bool WT = false>
__device__ __forceinline__ void gemm_epilogue_regs(const GemmArgs& p, f32x4 (&acc)[NI][MI], const int m_wave0, const int wave_n0,
                                                   const int fr, const int fq, const float (&ln_rstd_a)[MI]) {
#pragma clang fp contract(off)      // (round 5) fused multiply-adds are written out (fma4 / fmaf): every instantiation and every translation unit rounds alike
  constexpr int NIO = GEGLU ? NI / 2 : NI;     // output blocks per wave
  constexpr int NP = NIO / 2;                  // exchanged pairs
  constexpr bool ODD = (NIO & 1) != 0;         // a last block stored from the accumulator layout
  // residual loads run this many token blocks ahead.  Stores and loads share vmcnt on gfx9 and retire in issue order, so a residual load issued
  // AFTER a block's stores cannot be consumed before those stores are acknowledged: the 256-row x 160/128 kernels (MI = 4) therefore request
  // the whole residual tile before their first store (40 registers); the 256 x 256 kernel (MI = 8) has no room for that and stays 2 ahead
  constexpr int DEPTH = MI <= 4 ? MI : 2;
  static_assert(!GEGLU || NI % 4 == 0, "GEGLU: hidden and gate halves must be whole pairs");
  const int flags = p.flags;
  constexpr bool ACT = (FEAT & EPI_F_ACT) != 0;
  constexpr bool TANH = (FEAT & (EPI_F_ACT | EPI_F_TANH)) != 0;
  const bool qkv = (FEAT & EPI_F_QKV) != 0 && (flags & MX_EPI_QKV) != 0;
  int seg_idx = 0, seg_grp = 0, seg_pos = 0;
  bool to_vt = false;
  if (qkv) {                                   // the wave's feature range lies inside one segment (seg % (16 NI) == 0)
    seg_idx = wave_n0 / p.seg;
    seg_grp = seg_idx / p.period;
    seg_pos = seg_idx - seg_grp * p.period;
    to_vt = seg_pos == p.period - 1;
  }
  const bool rms = qkv && (flags & MX_EPI_RMSNORM) && !to_vt;
  const float w_scale = (!rms && p.out_scale != 0.f && (!qkv || seg_pos == 0)) ? p.out_scale : 1.0f;
  // first output column of the wave
  const int col0 = GEGLU ? wave_n0 / 2 : qkv ? seg_grp * (p.period - 1) * p.seg + seg_pos * p.seg + (wave_n0 - seg_idx * p.seg) : wave_n0;
  const bool has_res = !GEGLU && p.residual != nullptr;
  const bool has_rb = VEC && !GEGLU && p.rowbias != nullptr, has_gate = VEC && !GEGLU && p.gate != nullptr;

  f32x4 bias_r[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    bias_r[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (p.bias && (GEGLU || i < NIO)) bias_r[i] = *reinterpret_cast<const f32x4*>(p.bias + wave_n0 + i * 16 + fq * 4);
  }
  f32x4 rmsw_r[NI];                            // RMSNorm weight of the lane's features (one 64-wide head per wave)
  if (rms) {
    const float* w = seg_pos == 0 ? p.rms_wq : p.rms_wk;
#pragma unroll
    for (int i = 0; i < NI; ++i) rmsw_r[i] = *reinterpret_cast<const f32x4*>(w + ((wave_n0 - seg_idx * p.seg + i * 16 + fq * 4) & 63));
  }
  // folded LayerNorm: the accumulators arrive as x W'^T - mean_m colsum_n (gemm_ln_init) and ln_rstd_a holds rstd_m (1 without the fold,
  // and acc * 1 + bias is acc + bias exactly)
  const bool stats = EMIT && !GEGLU && !qkv && p.stats_out != nullptr;

  // ---- V^T waves of the fused q | k | v projection: keys along the lanes, 2-byte stores (see gemm_epilogue); nothing else applies to them
  //      (no RMSNorm, no q scale, no residual), so they take their own short path here -- with their own token walk, so that neither this
  //      path's state nor the row arrays below are alive in the other (round 3: held together they spilled, and every reload of a spilled
  //      address waits on vmcnt = on all the stores issued before it) ----
  if constexpr ((FEAT & EPI_F_QKV) != 0) {
    if (qkv && to_vt) {
      const int rpb_v = p.rows_per_batch > 0 ? p.rows_per_batch : 0x7fffffff;
      const int m_first = m_wave0 + fr;
      int bv = p.rows_per_batch > 0 ? m_first / p.rows_per_batch : 0;
      int rv = m_first - bv * (p.rows_per_batch > 0 ? p.rows_per_batch : 0);
      const int key_off = p.c_batch_rows > 0 ? p.c_row_off : 0;
      const int nv = p.N / p.period;
      const long feat0 = (long)seg_grp * p.seg + (wave_n0 - seg_idx * p.seg) + fq * 4;     // V feature of v[0][0]
      auto rstd_at = [&](int j) __attribute__((always_inline)) -> float {
        const int m = m_first + 16 * j;
        return p.ln_final[(long)(m < p.M ? m : p.M - 1) * 2 + 1];
      };
      float rs_next = RSTD_LOAD ? rstd_at(0) : 1.0f;
#pragma unroll
      for (int j = 0; j < MI; ++j) {
        const float rs_cur = rs_next;
        if (RSTD_LOAD && j + 1 < MI) rs_next = rstd_at(j + 1);
        if (m_first + 16 * j < p.M) {
          const int key = MX_VT_POS(key_off + rv);
          bf16_t* dst = p.vt + ((long)bv * nv + feat0) * p.ldvt + key;
          const float rs = RSTD_LOAD ? rs_cur : STATS ? ln_rstd_a[j] : 1.0f;
#pragma unroll
          for (int i = 0; i < NIO; ++i) {
            const f32x4 t = STATS ? fma4(acc[i][j], rs, bias_r[i]) : acc[i][j] + bias_r[i];
#pragma unroll
            for (int q = 0; q < 4; ++q) dst[(long)(i * 16 + q) * p.ldvt] = f32_to_bf16(t[q]);
          }
        }
        rv += 16;
        if (rv >= rpb_v) { rv -= rpb_v; ++bv; }
      }
      return;
    }
  }
  // ---- rows.  Lane (fr, fq) holds token m_j = m_wave0 + 16 j + fr of token block j; its lane pair (fr ^ 1) holds tokens A_j = m_j & ~1 and
  //      B_j = A_j + 1.  A token is (batch b, row r inside the batch): m = b * rows_per_batch + r, and its output row is b * c_batch_rows +
  //      c_row_off + r (joint-sequence remap; without it the row is m).  ONE integer division per lane finds (b, r) of A_0; every later
  //      token is 16 rows further, a compare-and-wrap (the launcher keeps rows_per_batch == 0 or >= 16: pick_tile).  Round 3: the
  //      divisions, 64-bit multiplies and selects of the per-block form were most of the epilogue's 4 000 instructions per wave.
  //      Tokens past M take the row of token M - 1: loads stay in bounds and need no branch, stores are masked by m < M. ----
  const bool odd_lane = (fr & 1) != 0;
  const int rpb = p.rows_per_batch > 0 ? p.rows_per_batch : 0x7fffffff;          // no batches: one batch holds every row
  const int cbr = p.c_batch_rows > 0 ? p.c_batch_rows : (p.rows_per_batch > 0 ? p.rows_per_batch : 0);
  const int cro = p.c_batch_rows > 0 ? p.c_row_off : 0;
  const int b_last = p.rows_per_batch > 0 ? (p.M - 1) / p.rows_per_batch : 0;     // (wave-uniform: scalar unit)
  const int r_last = p.M - 1 - b_last * (p.rows_per_batch > 0 ? p.rows_per_batch : 0);
  const int row_last = b_last * cbr + cro + r_last;
  const int mA0 = (m_wave0 + fr) & ~1;
  int row_a[MI], row_b[MI];                    // output rows of A_j, B_j
  int bo_walk = 0, ro_walk = 0;                // batch / row in batch of the lane's OWN token at the block load_batch_vectors is called for next
  {
    int bA = p.rows_per_batch > 0 ? mA0 / p.rows_per_batch : 0;
    int rA = mA0 - bA * (p.rows_per_batch > 0 ? p.rows_per_batch : 0);
#pragma unroll
    for (int j = 0; j < MI; ++j) {
      int bB = bA, rB = rA + 1;
      if (rB >= rpb) { rB = 0; ++bB; }
      const int mA = mA0 + 16 * j;
      row_a[j] = mA < p.M ? bA * cbr + cro + rA : row_last;
      row_b[j] = mA + 1 < p.M ? bB * cbr + cro + rB : row_last;
      if constexpr (VEC) { if (j == 0) { bo_walk = odd_lane ? bB : bA; ro_walk = odd_lane ? rB : rA; } }
      rA += 16;
      if (rA >= rpb) { rA -= rpb; ++bA; }
    }
  }
  auto token = [&](int j) __attribute__((always_inline)) { return m_wave0 + j * 16 + fr; };
  auto row_own = [&](int j) __attribute__((always_inline)) { return odd_lane ? row_b[j] : row_a[j]; };
  // byte addresses: base + row * (leading dimension in bytes) -- one v_mad_u64_u32 per access
  const char* const res_base = reinterpret_cast<const char*>(p.residual);
  char* const c_base = reinterpret_cast<char*>(p.c);
  const unsigned ldr_b = (unsigned)p.ldr * 2u, ldc_b = (unsigned)p.ldc * 2u;
  auto res_at = [&](int row, int col) __attribute__((always_inline)) { return res_base + (unsigned long long)(unsigned)row * ldr_b + (unsigned)(col * 2); };
  auto c_at = [&](int row, int col) __attribute__((always_inline)) { return c_base + (unsigned long long)(unsigned)row * ldc_b + (unsigned)(col * 2); };
  // stores of the output tile: plain, or write-through through a buffer descriptor of C (WT: the launcher keeps C below 2 GB)
  const auto c_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.c, 0, WT ? (int)((unsigned)p.M * ldc_b) : 0, 0x00020000);
  auto store_c16 = [&](int row, int col, const u32x4 v) __attribute__((always_inline)) {
    if constexpr (WT) __builtin_amdgcn_raw_buffer_store_b128(v, c_rsrc, (int)((unsigned)row * ldc_b + (unsigned)(col * 2)), 0, 16);
    else *reinterpret_cast<u32x4*>(c_at(row, col)) = v;
  };
  auto store_c8 = [&](int row, int col, const u32x2 v) __attribute__((always_inline)) {
    if constexpr (WT) __builtin_amdgcn_raw_buffer_store_b64(v, c_rsrc, (int)((unsigned)row * ldc_b + (unsigned)(col * 2)), 0, 16);
    else *reinterpret_cast<u32x2*>(c_at(row, col)) = v;
  };

  auto load_res = [&](int j, int pr) __attribute__((always_inline)) -> u32x4 {
    if (!has_res) return u32x4{0u, 0u, 0u, 0u};
    return *reinterpret_cast<const u32x4*>(res_at(row_own(j), col0 + pr * 32 + fq * 8));
  };
  auto load_res_odd = [&](int j) __attribute__((always_inline)) -> u32x2 {
    if (!has_res) return u32x2{0u, 0u};
    return *reinterpret_cast<const u32x2*>(res_at(row_own(j), col0 + NP * 32 + fq * 4));
  };
  // per-sample vectors (row bias of the time embedding, AdaLN gate) of the lane's token: one token block ahead
  f32x4 rb_r[VEC ? NIO : 1], gate_r[VEC ? NIO : 1];
  auto load_batch_vectors = [&](int j) __attribute__((always_inline)) {
    if constexpr (VEC) {                       // (called for j = 0, 1, 2, ... in order: the walk advances one block per call)
      (void)j;
      const int bidx = bo_walk < b_last ? bo_walk : b_last;
      ro_walk += 16;
      if (ro_walk >= rpb) { ro_walk -= rpb; ++bo_walk; }
#pragma unroll
      for (int i = 0; i < NIO; ++i) {
        const int n = wave_n0 + i * 16 + fq * 4;
        if (has_rb) rb_r[i] = *reinterpret_cast<const f32x4*>(p.rowbias + (long)bidx * p.ldrb + n);
        if (has_gate) gate_r[i] = *reinterpret_cast<const f32x4*>(p.gate + (long)bidx * p.ldg + n);
      }
    }
  };
  if constexpr (VPF) { if (has_rb || has_gate) load_batch_vectors(0); }
  // FULL-LINE form (bf16 output, two pairs at a time).  After the pair exchange the two lanes of a lane pair (tokens T, T + 1)
  // swap one pair each, so that instruction A writes token T -- the even lane its first pair, the odd lane token T's second
  // pair: 8 lanes x 16 B = 128 contiguous bytes per token -- and instruction B writes token T + 1.  Measured 2.7x the per-CU
  // store rate of 64-byte row pieces (profiles/r02_a_dma_stream_and_store_microbench.txt).  The residual is read the same way.
  constexpr int NG = NP / 2;                   // groups of two pairs
  constexpr bool fullmode = NG > 0;     // (fp32 output is served by the generic kernel: pick_tile)
  const int col_full = col0 + (odd_lane ? 32 : 0) + fq * 8;     // this lane's column in group 0 (group g: + 64 g)
  // slot 2g of a group: (token A = even token of the lane pair, this lane's column pair); slot 2g + 1: (token B = odd token, same pair)
  auto load_res_full = [&](int j, int g, int which) __attribute__((always_inline)) -> u32x4 {
    if (!has_res) return u32x4{0u, 0u, 0u, 0u};
    return *reinterpret_cast<const u32x4*>(res_at(which ? row_b[j] : row_a[j], col_full + 64 * g));
  };
  auto load_res_any = [&](int j, int pr) __attribute__((always_inline)) -> u32x4 {
    if (pr < 2 * NG) return load_res_full(j, pr >> 1, pr & 1);
    return load_res(j, pr);
  };
  u32x4 res_r[DEPTH][NP > 0 ? NP : 1];
  u32x2 res_o[DEPTH];
  {
#pragma unroll
    for (int d = 0; d < DEPTH && d < MI; ++d) {
#pragma unroll
      for (int pr = 0; pr < NP; ++pr) res_r[d][pr] = load_res_any(d, pr);
      if constexpr (ODD) res_o[d] = load_res_odd(d);
    }
  }

  auto rstd_at = [&](int j) __attribute__((always_inline)) -> float {
    const int m = token(j);
    return p.ln_final[(long)(m < p.M ? m : p.M - 1) * 2 + 1];
  };
  float rstd_next = RSTD_LOAD ? rstd_at(0) : 1.0f;
#pragma unroll
  for (int j = 0; j < MI; ++j) {
    const int m = token(j);
    // ---- accumulator layout: bias, RMSNorm / scale, GEGLU, per-sample vectors ----
    float v[NIO][4];
    float rms_mul = 1.0f;
    if constexpr (!VPF) { if (has_rb || has_gate) load_batch_vectors(j); }
    const float rstd_cur = rstd_next;
    if (RSTD_LOAD && j + 1 < MI) rstd_next = rstd_at(j + 1);
    const float ln_rstd = RSTD_LOAD ? rstd_cur : STATS ? ln_rstd_a[j] : 1.0f;
    float st1 = 0.f, st2 = 0.f;                // stats_out: this lane's part of the token's sums
    if (rms) {                                 // every lane takes part in the shuffles (masking happens at the store)
      float ss = 0.f;
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) { const float t = acc[i][j][q] + bias_r[i][q]; ss = fmaf(t, t, ss); }
      ss = sum_over_fq(ss);
      rms_mul = rsqrtf(fmaf(ss, 1.0f / 64.0f, p.rms_eps)) * ((seg_pos == 0 && p.out_scale != 0.f) ? p.out_scale : 1.0f);
    }
#pragma unroll
    for (int i = 0; i < NIO; ++i) {
      f32x4 t = STATS ? fma4(acc[i][j], ln_rstd, bias_r[i]) : acc[i][j] + bias_r[i];
      if constexpr (GEGLU) {
        const f32x4 g = STATS ? fma4(acc[i + NI / 2][j], ln_rstd, bias_r[i + NI / 2]) : acc[i + NI / 2][j] + bias_r[i + NI / 2];
        if (ACT && (flags & MX_EPI_GEGLU_TANH)) {  // (GEGLU: ACT = the tanh form is compiled in)
#pragma unroll
          for (int q = 0; q < 4; ++q) t[q] = t[q] * gelu_tanh_f(g[q]);
        } else {                                   // two elements per packed instruction
          const gelu_f32x2 g01 = gelu_fast2(gelu_f32x2{g[0], g[1]}), g23 = gelu_fast2(gelu_f32x2{g[2], g[3]});
          t[0] *= g01[0]; t[1] *= g01[1]; t[2] *= g23[0]; t[3] *= g23[1];
        }
      } else {
        if (rms) {
#pragma unroll
          for (int q = 0; q < 4; ++q) t[q] *= rms_mul * rmsw_r[i][q];
        } else {
          t *= w_scale;
        }
        if constexpr (VEC) {
          if (has_rb) t += rb_r[i];
          if (has_gate) t *= gate_r[i];
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) v[i][q] = t[q];
    }
    if constexpr (VPF) { if (j + 1 < MI && (has_rb || has_gate)) load_batch_vectors(j + 1); }
    // ---- pairs of blocks: exchange, then residual -> activation -> store in the row layout ----
    auto finish8 = [&](float (&o)[8], const u32x4 rr) __attribute__((always_inline)) {
      if (has_res) {
#pragma unroll
        for (int q = 0; q < 4; ++q) { o[2 * q] += bf16lo_to_f32(rr[q]); o[2 * q + 1] += bf16hi_to_f32(rr[q]); }
      }
      if (ACT && (flags & MX_EPI_SILU)) {
#pragma unroll
        for (int q = 0; q < 8; ++q) o[q] = silu_f(o[q]);
      }
      if (TANH && (flags & MX_EPI_GELU_TANH)) {       // two elements per packed instruction
#pragma unroll
        for (int q = 0; q < 8; q += 2) { const gelu_f32x2 t2 = gelu_tanh2(gelu_f32x2{o[q], o[q + 1]}); o[q] = t2[0]; o[q + 1] = t2[1]; }
      }
      if (ACT && (flags & (MX_EPI_GELU | MX_EPI_QUICK_GELU))) {
#pragma unroll
        for (int q = 0; q < 8; ++q) o[q] = (flags & MX_EPI_GELU) ? gelu_fast(o[q]) : quick_gelu_f(o[q]);
      }
      if (stats) {
#pragma unroll
        for (int q = 0; q < 8; ++q) { st1 += o[q]; st2 = fmaf(o[q], o[q], st2); }
      }
    };
    if constexpr (fullmode) {
      const bool ok_a = (m & ~1) < p.M, ok_b = (m | 1) < p.M;
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        lane_exchange_pair(v[4 * g], v[4 * g + 1]);
        lane_exchange_pair(v[4 * g + 2], v[4 * g + 3]);
        float oa[8], ob[8];                      // (own token, pair 2g), (own token, pair 2g + 1)
#pragma unroll
        for (int q = 0; q < 4; ++q) { oa[q] = v[4 * g][q]; oa[4 + q] = v[4 * g + 1][q]; ob[q] = v[4 * g + 2][q]; ob[4 + q] = v[4 * g + 3][q]; }
        const u32x4 ra = res_r[j % DEPTH][2 * g], rb = res_r[j % DEPTH][2 * g + 1];     // (token A, my pair), (token B, my pair)
        if (j + DEPTH < MI) { res_r[j % DEPTH][2 * g] = load_res_full(j + DEPTH, g, 0); res_r[j % DEPTH][2 * g + 1] = load_res_full(j + DEPTH, g, 1); }
        u32x4 res_a = ra, res_b = rb;
        if (has_res) {
          // even lane: ra = (T, pair 2g) is its own; it lacks (T, pair 2g + 1) = the odd lane's ra.  Odd lane: rb = (T + 1, pair 2g + 1)
          // is its own; it lacks (T + 1, pair 2g) = the even lane's rb.
          const u32x4 nb_ra = lane_xor1(ra), nb_rb = lane_xor1(rb);
          res_a = odd_lane ? nb_rb : ra;
          res_b = odd_lane ? rb : nb_ra;
        }
        finish8(oa, res_a);
        finish8(ob, res_b);
        const u32x4 wa = {pack_bf16x2(oa[0], oa[1]), pack_bf16x2(oa[2], oa[3]), pack_bf16x2(oa[4], oa[5]), pack_bf16x2(oa[6], oa[7])};
        const u32x4 wb = {pack_bf16x2(ob[0], ob[1]), pack_bf16x2(ob[2], ob[3]), pack_bf16x2(ob[4], ob[5]), pack_bf16x2(ob[6], ob[7])};
        // even lane gives away its second pair and receives the odd token's first pair; the odd lane the other way round
        // (two unconditional neighbour reads, each consumed by one select: the DPP move folds into the v_cndmask)
        const u32x4 nb_b = lane_xor1(wb), nb_a = lane_xor1(wa);
        const u32x4 st_a = odd_lane ? nb_b : wa;     // token A (even): even lane pair 2g, odd lane pair 2g + 1 (the even lane's wb)
        const u32x4 st_b = odd_lane ? wb : nb_a;     // token B (odd): even lane pair 2g (the odd lane's wa), odd lane pair 2g + 1
        if (ok_a) store_c16(row_a[j], col_full + 64 * g, st_a);
        if (ok_b) store_c16(row_b[j], col_full + 64 * g, st_b);
      }
    }
#pragma unroll
    for (int pr = 2 * NG; pr < NP; ++pr) {
      lane_exchange_pair(v[2 * pr], v[2 * pr + 1]);
      float o[8];
#pragma unroll
      for (int q = 0; q < 4; ++q) { o[q] = v[2 * pr][q]; o[4 + q] = v[2 * pr + 1][q]; }
      const u32x4 rr = res_r[j % DEPTH][pr];
      if (j + DEPTH < MI) res_r[j % DEPTH][pr] = load_res(j + DEPTH, pr);
      finish8(o, rr);
      if (m < p.M) {
        const u32x4 w = {pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7])};
        store_c16(row_own(j), col0 + pr * 32 + fq * 8, w);
      }
    }
    // ---- odd last block: straight from the accumulator layout ----
    if constexpr (ODD) {
      float o[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) o[q] = v[NIO - 1][q];
      const u32x2 rr = res_o[j % DEPTH];
      if (j + DEPTH < MI) res_o[j % DEPTH] = load_res_odd(j + DEPTH);
      if (has_res) { o[0] += bf16lo_to_f32(rr[0]); o[1] += bf16hi_to_f32(rr[0]); o[2] += bf16lo_to_f32(rr[1]); o[3] += bf16hi_to_f32(rr[1]); }
      if (ACT && (flags & MX_EPI_SILU)) {
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q] = silu_f(o[q]);
      }
      if (TANH && (flags & MX_EPI_GELU_TANH)) {
#pragma unroll
        for (int q = 0; q < 4; q += 2) { const gelu_f32x2 t2 = gelu_tanh2(gelu_f32x2{o[q], o[q + 1]}); o[q] = t2[0]; o[q + 1] = t2[1]; }
      }
      if (ACT && (flags & (MX_EPI_GELU | MX_EPI_QUICK_GELU))) {
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q] = (flags & MX_EPI_GELU) ? gelu_fast(o[q]) : quick_gelu_f(o[q]);
      }
      if (stats) {
#pragma unroll
        for (int q = 0; q < 4; ++q) { st1 += o[q]; st2 = fmaf(o[q], o[q], st2); }
      }
      if (m < p.M) store_c8(row_own(j), col0 + NP * 32 + fq * 4, u32x2{pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])});
    }
    if (stats) {                               // the token's four lanes hold disjoint columns of the wave's panel: slab = panel index
      st1 = sum_over_fq(st1); st2 = sum_over_fq(st2);      // (round 5: permlane swaps instead of four trips through the LDS crossbar per token block; same sums, same order)
      if (fq == 0 && m < p.M) {
        const int slab = wave_n0 / (16 * NI), pitch = (p.N / (16 * NI) + 3) & ~3;
        float* dst = p.stats_out + ((long)m * pitch + slab) * 2;
        if (WT || p.ln_final_out != nullptr) {  // read back inside this launch (gemm_ln_finalize by the panel's last workgroup; a chained launch's next stage): write-through (sc1)
          const unsigned long long bits = (unsigned long long)__float_as_uint(st1) | ((unsigned long long)__float_as_uint(st2) << 32);
          __hip_atomic_store(reinterpret_cast<unsigned long long*>(dst), bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
          *reinterpret_cast<f32x2*>(dst) = f32x2{st1, st2};
        }
      }
    }
  }
}

}  // namespace mx
