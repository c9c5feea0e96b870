// The ATTENTION TAIL of a BasicTransformerBlock as ONE launch (round 5):
//
//     stage 0   y  = attn1.to_out(ao) + y            (+ row statistics of y for the folded norm2)          transformer.py:204-236, attention.py:204-232
//     stage 1   q2 = attn2.to_q(norm2(y)) * scale                                                          attention.py:73-79
//     stage 2   ao2 = softmax(q2 K^T) V   over the 77 text keys (K / V^T hoisted, per sample)             attention.py:80-96
//     stage 3   y  = attn2.to_out(ao2) + y           (+ finalised row statistics for the folded norm3)     attention.py:97-110, transformer.py:239-262
//
// As four launches each stage is one round of 256 tiles (256 x 160, one per CU) and pays its own entry (descriptor + first operand fetch: ~3.6 us),
// store drain and dispatch gap (~4.5 us) on a 20-us K loop, and the cross-attention launch sits on the small-launch floor (profiles/r04_timeline_v4_final.txt,
// r04_n_cross_attn_bench.txt).  Here the four stages are work items of one persistent launch.  A stage-s item of a 256-row PANEL needs stage s - 1 of the SAME
// panel only (a row of to_q reads the whole row of y; a head's queries their own rows), so a panel's nt items per stage are taken by nt workgroups that
// hand the panel's rows to each other through memory inside the launch while the other panels run beside them:
//
//   * work queue: workgroup b serves queue b & 7 (the XCD the hardware deals it to: panels 8 i + q live in ONE L2).  A queue's items are numbered
//     round by round (G panels at a time), stage-major inside a round, so that an item's dependencies always carry SMALLER tickets: whoever holds a
//     ticket waits only for items already taken by running (or finished) workgroups -- no residency or placement assumption, any number of
//     workgroups (even one) drains a queue; XCD affinity is for speed only (cdna guide, Guideline 16 / "Contract");
//   * hand-off (Guideline 16, R1 with the acquire kept): every handed-off byte is stored WRITE-THROUGH (sc1) by the GEMM epilogue / attention body
//     (gemm_args.h WT, attn_cross_body.h WT), every storing wave drains (s_waitcnt vmcnt(0)), the workgroup barriers, ONE lane adds to the panel's
//     done counter (agent scope); the consumer's lane 0 polls that counter relaxed (sc1 load + s_sleep, bounded), ONE agent-scope acquire invalidates the
//     CU's L1, vmcnt(0), workgroup barrier, then plain loads (the LDS-DMA operand stream included).  No address is rewritten after another XCD may have
//     read it earlier in the launch (the two statistics buffers and ao / ao2 are distinct: mx_attn_tail_supported), so stale L2 lines cannot arise under
//     any placement;
//   * arithmetic: the GEMM stages run gemm_v5_tile (gemm_v5_body.h) on the instantiation the separate launches take, the attention stage
//     xk_block (attn_cross_body.h): same tiles, same order of summation, same epilogues -- the launch's results equal the four launches' BIT FOR BIT
//     (tests/test_attn_tail_gpu.py; mx_attention_cross_prescaled is the separate form of stage 2);
//   * the counters are left zero by the last workgroup to leave (every launch finds and leaves them zero); a poll that does not end within ~2^22
//     sleeps sets the error word and the workgroup abandons its waits (mx_attn_tail_status) instead of hanging the device.
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "common.h"
#include "../../include/mxdenoise.h"
#include "gemm_args.h"
#include "gemm_v5_body.h"
#include "attn_cross_body.h"

namespace mx {

int gemm_prepare_for_chain(void* stream, const mx_gemm_desc* d, GemmArgs& a, int& bn, int& rows, int& splitk);

constexpr int kTailBN = 160;
constexpr int kTailStages = 4;
// sync words (unsigned): queue tickets on lines of their own, then exit counter, error word, then done[panel][stage]
constexpr int kTailQStride = 32;
constexpr int kTailExit = 8 * kTailQStride;
constexpr int kTailErr = kTailExit + 1;
constexpr int kTailDone = kTailExit + 32;
constexpr unsigned kTailSpinLimit = 1u << 22;

struct TailArgs {
  GemmArgs g[3];          // stage 0, 1, 3
  AttnArgs x;             // stage 2 over the whole problem
  unsigned* sync;
  int mt, nt, G;          // 256-row panels, tiles (= work items) per panel and stage, panels per round of a queue
  int L;                  // tokens per sample (a panel lies inside one sample: L % 256 == 0)
};

 static_assert(sizeof(TailArgs) <= 4096, "kernel arguments are limited to 4 KB");

typedef __attribute__((address_space(3))) char lds_char;

// Stage 2 of a panel: the 77-key cross-attention of the panel's 256 rows for the heads == tile (mod nt) -- at most three.  The workgroup's eight waves take one
// 32-query block of the panel each, for every one of those heads.  The heads' K / V^T fragments (24 x 1 KB per head, in the MFMA operand layout, masked:
// xk_kfrag / xk_vfrag) are fetched ONCE per workgroup -- each wave a ninth of them -- and staged in LDS; the queries of all the wave's blocks are requested
// at the same time: one memory round trip per item, then register / LDS work only.  A function of its own (not inlined): inside the kernel body the register
// allocator, holding the GEMM stage's loop invariants, spilled the query fragments to scratch and this stage took 19 us per item.
// (First forms: a (head, 64-query) unit per wave and turn as the stand-alone kernel's waves do, 15.9 us per item; a head per turn shared by the eight waves
// through the CU's L1, 14.1 us: both pay a fragment fetch per head and wave, and nothing hides it.)
__device__ __attribute__((noinline)) void tail_xattn(const bf16_t* q, const bf16_t* k, const bf16_t* vt, bf16_t* o, int ldq, int ldk, int ldvt, int ldo, long vt_bstride,
                                                     int B, int H, int Lq, int Lk, int panel, int tile, int nt, int L, unsigned lds_base) {
  AttnArgs x;
  x.q = q; x.k = k; x.vt = vt; x.o = o; x.ldq = ldq; x.ldk = ldk; x.ldvt = ldvt; x.ldo = ldo; x.vt_bstride = vt_bstride; x.B = B; x.H = H; x.Lq = Lq; x.Lk = Lk;
  x.scale_log2 = 1.0f; x.xcd_map = 0; x.key_chunk = 0; x.k_bstride = 0; x.k_cstride = 0; x.vt_cstride = 0; x.causal = 0; x.bias = nullptr; x.ldb = 0;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  constexpr int NF = XK_KFRAGS + XK_VFRAGS;
  lds_char* const frags = reinterpret_cast<lds_char*>((uintptr_t)lds_base);
  char* const patch = (char*)(frags + 3 * NF * 1024 + wave * 4096);       // (generic pointer: xk_block's patch accesses become flat_ accesses of LDS)
  const int row0 = panel * 256;
  const int b = row0 / L;
  const int q0 = row0 - b * L + wave * 32;
  const int nh = (H - tile + nt - 1) / nt;                  // heads of this item (<= 3: tail_prepare)
  bf16x8 qf[3][4];
#pragma unroll
  for (int sl = 0; sl < 3; ++sl) if (sl < nh) xk_load_q(x, b, tile + sl * nt, q0, lane, qf[sl]);
  // the wave's ninth of the fragments: ALL requested before the first is written to LDS (a load-then-store loop serialised nine memory round trips:
  // 7.7 us per head)
  constexpr int NMY = 3 * NF / 8;
  static_assert(3 * NF % 8 == 0, "the fragments of three heads are dealt to eight waves");
  bf16x8 mine[NMY];
#pragma unroll
  for (int j = 0; j < NMY; ++j) {
    const int i = wave + 8 * j;
    if (i < nh * NF) {
      const int sl = i / NF, f = i - sl * NF, head = tile + sl * nt;
      mine[j] = f < XK_KFRAGS ? xk_kfrag(x, b, head, f >> 2, f & 3, lane) : xk_vfrag(x, b, head, (f - XK_KFRAGS) >> 1, (f - XK_KFRAGS) & 1, lane);
    }
  }
#pragma unroll
  for (int j = 0; j < NMY; ++j) {
    const int i = wave + 8 * j;
    if (i < nh * NF) *reinterpret_cast<__attribute__((address_space(3))) bf16x8*>(frags + (i * 64 + lane) * 16) = mine[j];
  }
  __syncthreads();
#pragma unroll
  for (int sl = 0; sl < 3; ++sl) {
    if (sl < nh) {
      const lds_char* fb = frags + (sl * NF * 64 + lane) * 16;
      // all of the head's fragments up front (24 LDS reads in flight), then the block on registers
      bf16x8 kf[XK_MAXBLK][4], vf[2 * XK_MAXBLK][2];
#pragma unroll
      for (int kb = 0; kb < XK_MAXBLK; ++kb)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) kf[kb][ks] = *reinterpret_cast<const __attribute__((address_space(3))) bf16x8*>(fb + (kb * 4 + ks) * 1024);
#pragma unroll
      for (int st = 0; st < 2 * XK_MAXBLK; ++st)
#pragma unroll
        for (int db = 0; db < 2; ++db) vf[st][db] = *reinterpret_cast<const __attribute__((address_space(3))) bf16x8*>(fb + (XK_KFRAGS + st * 2 + db) * 1024);
      xk_block<true, true>(x, b, tile + sl * nt, q0, qf[sl], [&](int kb, int ks) __attribute__((always_inline)) { return kf[kb][ks]; },
                           [&](int st, int db) __attribute__((always_inline)) { return vf[st][db]; }, patch, lane);
    }
  }
}

#ifdef MX_TAIL_STAMPS   // diagnostic build (tools/exp/tail_timeline.py): wall-clock stamps (100 MHz) per workgroup and work item
__device__ unsigned long long g_tail_stamps[256 * 16 * 6];
#define TAIL_STAMP(slot, val) do { if (tid == 0 && blockIdx.x < 256 && n_item < 16) g_tail_stamps[(blockIdx.x * 16 + n_item) * 6 + (slot)] = (val); } while (0)
#define TAIL_NOW() __builtin_amdgcn_s_memrealtime()
extern "C" int mx_debug_tail_stamps(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tail_stamps), sizeof(g_tail_stamps)); }
#else
#define TAIL_STAMP(slot, val) do {} while (0)
#define TAIL_NOW() 0ull
#endif

__global__ __launch_bounds__(512, 2) void attn_tail_kernel(const TailArgs t) {
  constexpr int STAGE_ELEMS = (256 + kTailBN) * BK5;
  __shared__ __attribute__((aligned(16))) bf16_t smem[NSTAGE5 * STAGE_ELEMS];
  __shared__ int s_ctl[4];      // [0] ticket, [1] abandon flag
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int queue = blockIdx.x & 7;
  const int np_q = t.mt > queue ? (t.mt - queue + 7) >> 3 : 0;           // panels queue, queue + 8, ...
  const int per_stage = t.G * t.nt;                                      // items of one stage in one round
  const int per_round = per_stage * kTailStages;
  const int total = ((np_q + t.G - 1) / t.G) * per_round;
  unsigned* const qticket = t.sync + queue * kTailQStride;
  unsigned* const done = t.sync + kTailDone;
  if (tid == 0) s_ctl[1] = 0;
  bool abandoned = false;
  [[maybe_unused]] int n_item = 0;
  for (;;) {
    __syncthreads();                            // (the control words of the previous item have been read by everyone)
    if (tid == 0) s_ctl[0] = (int)__hip_atomic_fetch_add(qticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int ticket = __builtin_amdgcn_readfirstlane(s_ctl[0]);
    if (ticket >= total) break;
    const int round = ticket / per_round;
    const int in_round = ticket - round * per_round;
    const int stage = in_round / per_stage;
    const int idx = in_round - stage * per_stage;
    const int pl = round * t.G + idx / t.nt;
    const int tile = idx % t.nt;
    if (pl >= np_q) continue;                   // (the last round of a queue may be short)
    const int panel = queue + 8 * pl;
    TAIL_STAMP(0, ((unsigned long long)panel << 16) | (unsigned long long)(stage << 8) | (unsigned long long)tile);
    TAIL_STAMP(1, TAIL_NOW());
    if (stage > 0 && !abandoned) {
      // ---- wait for stage - 1 of this panel: ONE lane polls relaxed, ONE acquire, then the workgroup ----
      if (tid == 0) {
        const unsigned* flag = done + panel * kTailStages + (stage - 1);
        unsigned spins = 0;
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)t.nt) {
          __builtin_amdgcn_s_sleep(8);
          if (++spins > kTailSpinLimit) {
            __hip_atomic_store(t.sync + kTailErr, (unsigned)(0x80000000u | (stage << 24) | panel), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_ctl[1] = 1;
            break;
          }
        }
      }
      if (wave == 0) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __syncthreads();
      abandoned = s_ctl[1] != 0;
    }
    TAIL_STAMP(2, TAIL_NOW());
    if (!abandoned) {
      if (stage == 2) {
        tail_xattn(t.x.q, t.x.k, t.x.vt, t.x.o, t.x.ldq, t.x.ldk, t.x.ldvt, t.x.ldo, t.x.vt_bstride, t.x.B, t.x.H, t.x.Lq, t.x.Lk, panel, tile, t.nt, t.L,
                   (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem);
      } else {
        const GemmArgs& pk = t.g[stage == 3 ? 2 : stage];     // (ONE dynamically indexed read of the kernel-argument segment: a select of three references loads all three)
        gemm_v5_tile<kTailBN, 4, false, 0, false, false, true>(pk, panel, tile, smem);
      }
    }
    // ---- publish: every storing wave drains, the workgroup barriers, one lane signals ----
    TAIL_STAMP(3, TAIL_NOW());
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    TAIL_STAMP(4, TAIL_NOW());
    if (tid == 0) __hip_atomic_fetch_add(done + panel * kTailStages + stage, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    TAIL_STAMP(5, TAIL_NOW());
    ++n_item;
  }
  // ---- the last workgroup to leave zeroes the counters (everyone else is past its last poll) ----
  __syncthreads();                              // (every wave has read its last ticket)
  if (tid == 0) s_ctl[0] = (int)__hip_atomic_fetch_add(t.sync + kTailExit, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (s_ctl[0] == (int)gridDim.x - 1) {
    for (int i = tid; i < 8; i += 512) __hip_atomic_store(t.sync + i * kTailQStride, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int i = tid; i < t.mt * kTailStages; i += 512) __hip_atomic_store(done + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid == 0) __hip_atomic_store(t.sync + kTailExit, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// ---- host side ----
// The step plans take the chained launch only when asked to (MX_ATTN_TAIL=1): measured on MI355X it is a tie per layer against the four launches in
// isolation and 1.2 ms per SDXL step SLOWER inside the step (profiles/r05_c_attn_tail_*.txt) -- the per-panel waits and drains cost what
// the three kernel boundaries cost, and the write-through hand-offs take the hidden state out of the L2s for the launches that follow.
static bool tail_enabled() {
  static const bool on = [] { const char* e = getenv("MX_ATTN_TAIL"); return e && e[0] == '1'; }();
  return on;
}

// fills the kernel arguments; `why` (optional) receives the reason when the launch cannot be served.  Returns 0 when it can.
static int tail_prepare(void* stream, const mx_attn_tail_desc* d, TailArgs& t, std::string* why) {
  auto no = [&](const char* m) { if (why) *why = m; return 1; };
  if (!d) return no("null descriptor");
  const mx_gemm_desc* gd[3] = {&d->out1, &d->to_q, &d->out2};
  const int M = d->out1.M, C = d->out1.N;
  if (M <= 0 || C <= 0 || C % kTailBN != 0 || C % 64 != 0) return no("the width must be a multiple of 160 and of 64");
  if (d->heads * 64 != C || d->B <= 0 || d->L <= 0 || d->L % 256 != 0 || (long)d->B * d->L != M) return no("heads * 64 == C, M == B * L and L % 256 == 0 are required");
  if (d->ctx_len <= 0 || d->ctx_len > 32 * XK_MAXBLK) return no("the short-key attention serves at most 96 keys");
  if (d->heads > 3 * (C / kTailBN)) return no("an attention item stages at most three heads");
  if (!d->k || !d->vt || !d->sync || d->ldk < C || d->ldvt < MX_VT_LD(d->ctx_len) || d->ldvt % 8 != 0 || d->ldk % 8 != 0) return no("bad K / V^T / sync operands");
  for (int i = 0; i < 3; ++i) {
    const mx_gemm_desc& g = *gd[i];
    if (g.M != M || g.N != C || g.K != C || g.n_segs != 0 || g.flags != 0 || g.rowbias || g.gate || g.a2 || g.a_batch_rows > 0 || g.c_batch_rows > 0 || g.gn_part_out ||
        g.ln_final || g.splitk > 1)
      return no("every stage is a plain C x C linear over the same rows (bias, residual, statistics, folded LayerNorm by slabs, output scale only)");
    if ((long)M * g.ldc * 2 >= (1L << 31) || g.ldc % 8 != 0) return no("an output exceeds the 2 GB reach of the write-through stores");
  }
  // the chain: y -> to_q -> q2 -> attention -> ao2 -> to_out -> y
  if (d->to_q.a != d->out1.c || d->to_q.lda != d->out1.ldc) return no("to_q must read what out1 writes");
  if (d->out2.a == d->out1.a || d->out2.a == d->out1.c || d->to_q.c == d->out1.c || d->to_q.c == d->out2.a || d->to_q.c == d->out1.a)
    return no("ao, y, q2 and ao2 must be four different buffers");
  if (!d->out1.stats_out || d->to_q.ln_stats != d->out1.stats_out || !d->to_q.ln_colsum) return no("to_q takes its folded LayerNorm from out1's row statistics");
  if (d->out1.ln_final_out || d->to_q.stats_out) return no("out1 leaves slab statistics only; to_q none");
  if (d->out2.stats_out && d->out2.stats_out == d->out1.stats_out) return no("the two statistics buffers must differ (a rewritten slab could be read stale)");
  if (d->out2.ln_stats || d->out1.ln_stats) return no("the output projections take no folded LayerNorm");
  int bn = 0, rows = 0, sk = 0;
  for (int i = 0; i < 3; ++i) {
    if (gemm_prepare_for_chain(stream, gd[i], t.g[i], bn, rows, sk)) return no(mx_last_error());
    if (bn != kTailBN || rows != 256 || sk > 1) return no("a stage does not take the 256 x 160 tile (the chained launch runs that kernel's tiles)");
  }
  if (d->to_q.ln_slabs != mx_gemm_stats_slabs(&d->out1)) return no("to_q.ln_slabs must be the slab count out1 writes");
  std::memset(&t.x, 0, sizeof(t.x));
  t.x.q = (const bf16_t*)d->to_q.c; t.x.ldq = d->to_q.ldc; t.x.k = (const bf16_t*)d->k; t.x.ldk = d->ldk; t.x.vt = (const bf16_t*)d->vt; t.x.ldvt = d->ldvt;
  t.x.vt_bstride = d->vt_batch_stride; t.x.o = (bf16_t*)const_cast<void*>(d->out2.a); t.x.ldo = d->out2.lda;
  t.x.B = d->B; t.x.H = d->heads; t.x.Lq = d->L; t.x.Lk = d->ctx_len; t.x.scale_log2 = 1.0f;
  if ((long)M * t.x.ldo * 2 >= (1L << 31) || t.x.ldo % 8 != 0 || t.x.ldq % 8 != 0) return no("the attention output exceeds the 2 GB reach of the write-through stores");
  t.sync = d->sync;
  t.mt = cdiv(M, 256); t.nt = C / kTailBN;
  t.G = std::max(1, (cu_count() / 8) / t.nt);
  t.L = d->L;
  return 0;
}

}  // namespace mx

extern "C" size_t mx_attn_tail_sync_bytes(int M) {
  if (M <= 0) return 0;
  return ((size_t)(mx::kTailDone + mx::cdiv(M, 256) * mx::kTailStages) * sizeof(unsigned) + 255) & ~(size_t)255;
}

extern "C" int mx_attn_tail_preferred(void) { return mx::tail_enabled() ? 1 : 0; }

extern "C" int mx_attn_tail_supported(const mx_attn_tail_desc* d) {
  mx::TailArgs t;
  static const bool dbg = getenv("MX_ATTN_TAIL_DEBUG") != nullptr;
  std::string why;
  const bool ok = mx::tail_prepare(nullptr, d, t, dbg ? &why : nullptr) == 0;
  if (dbg && !ok) fprintf(stderr, "[mx attn_tail] not served: %s\n", why.c_str());
  return ok;
}

extern "C" int mx_attn_tail(void* stream, const mx_attn_tail_desc* d) {
  using namespace mx;
  TailArgs t;
  std::string why;
  if (tail_prepare(stream, d, t, &why)) { set_error("attn_tail: " + why); return 1; }
  hipStream_t s = (hipStream_t)stream;
  const double M = d->out1.M, C = d->out1.N;
  prof_begin(s, PROF_ATTN_TAIL, 3 * 2.0 * M * C * C + 4.0 * M * C * d->ctx_len, 2.0 * (6 * M * C + 3 * C * C), (int)M, (int)C, (int)C);
  hipLaunchKernelGGL(attn_tail_kernel, dim3(cu_count()), dim3(512), 0, s, t);
  prof_end(s);
  MX_LAUNCH_CHECK();
  return 0;
}

/* the error word of a sync buffer (0 = every wait of every launch ended); synchronises the stream */
extern "C" int mx_attn_tail_status(void* stream, const unsigned* sync, unsigned* word) {
  MX_CHECK(sync && word, "attn_tail_status: null operand");
  MX_HIP(hipMemcpyAsync(word, sync + mx::kTailErr, sizeof(unsigned), hipMemcpyDeviceToHost, (hipStream_t)stream));
  MX_HIP(hipStreamSynchronize((hipStream_t)stream));
  return 0;
}
