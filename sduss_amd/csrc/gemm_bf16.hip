// bf16 MFMA GEMM and implicit-GEMM 3x3 convolution for gfx950 (MI355X, CDNA4).
//
//   C[M, N] = X[M, K] * W[N, K]^T  (+ bias, + per-sample row bias, + residual, SiLU | GEGLU | QKV split)
//
// Orientation: the WEIGHT tile is the MFMA "A" operand and the ACTIVATION tile the "B" operand, so
// the accumulator of v_mfma_f32_16x16x32_bf16 (col = lane&15, row = 4*(lane>>4)+reg) holds, per lane,
// four CONSECUTIVE output features n of ONE token m: the epilogue stores 8 bytes per lane row-major
// and fuses bias / residual / activation without a transpose; the transposed V^T store that the
// attention kernel wants is the lane-contiguous direction (MI355X-first choice: no LDS round trip).
//
// Tile: 128 tokens x BN features (BN = 128 or 64) x BK = 64, 256 threads = 4 waves as 2(m) x 2(n);
// LDS rows are 128 B with the 16-byte chunk index XOR-swizzled by ((row>>1)&7), which makes the
// ds_read_b128 fragment reads of the 16x16x32 operand conflict-free (cdna guide T2, lane groups of
// ds_read_b128).  Register-prefetch double buffering, one barrier per K tile.
//
// Implicit GEMM conv (CONV=true): K = 9*Cin ordered tap-major, activations NHWC; the loader turns an
// output pixel + tap into a source pixel (stride 1/2, fused nearest x2 upsample, zero padding, and the
// reference's sliced-mode halo-corner rule -- see mxdenoise.h) and reads 16 B of channels from it.
//
// Reference call sites this serves: F.linear / F.conv2d issued by sduss/model_executor/modules/
// resnet.py:106,132,163 and attention.py:73-96,148-151,220 (through un-vendored diffusers/torch).
#include <algorithm>
#include <cstdlib>

#include <mutex>
#include <unordered_map>

#include "common.h"
#include "../../include/mxdenoise.h"
#include "gemm_args.h"

namespace mx {

constexpr int BM = 128;
constexpr int BK = 64;

__device__ __forceinline__ int swz(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }

template <int BN, bool CONV>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const GemmArgs pk) {
  constexpr int NI = BN / 32;  // 16-wide feature blocks per wave (wave covers BN/2 features)
  constexpr int MI = 4;        // 16-wide token blocks per wave (wave covers 64 tokens)
  constexpr int WROWS = BN / 32;  // W rows staged per thread
  __shared__ __attribute__((aligned(16))) bf16_t sX[2][BM * BK];
  __shared__ __attribute__((aligned(16))) bf16_t sW[2][BN * BK];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1;  // token half
  const int wn = wave & 1;   // feature half
  GemmArgs p = pk;              // grouped launch (gemm_args.h): blockIdx.x counts the m-tiles of all problems; p becomes this tile's problem
  int tm = blockIdx.x;
  gemm_select_seg(p, pk, tm);
  const int m0 = tm * BM;
  const int n0 = blockIdx.y * BN;
  const int nk = p.K / BK;

  // ---- staging assignment: thread -> (row = (tid>>3) + 32*i, chunk = tid&7) ----
  const int srow = tid >> 3;
  const int sch = tid & 7;

  // activation row descriptors
  int xoff[4];           // GEMM: element offset of the row start (or -1 when the row is out of range)
  int xoff2[4];          // the same inside a2 (dual-source A operand)
  int cb[4], cy[4], cx[4];  // CONV: batch, centre y/x in virtual-input coordinates (cb = -1: invalid row)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + srow + 32 * i;
    if constexpr (!CONV) {
      xoff[i] = (m < p.M) ? (int)(gemm_in_row(p, m) * p.lda) + sch * 8 : -1;
      xoff2[i] = (p.a2 != nullptr && m < p.M) ? m * p.lda2 + sch * 8 : 0;
    } else {
      if (m < p.M) {
        const int hw = p.Hout * p.Wout;
        const int b = m / hw;
        const int r = m - b * hw;
        const int oy = r / p.Wout;
        cb[i] = b;
        cy[i] = oy * p.stride;
        cx[i] = (r - oy * p.Wout) * p.stride;
      } else {
        cb[i] = -1; cy[i] = 0; cx[i] = 0;
      }
    }
  }
  int woff[WROWS];
#pragma unroll
  for (int i = 0; i < WROWS; ++i) {
    const int n = n0 + srow + 32 * i;
    woff[i] = (n < p.N) ? n * p.K + sch * 8 : -1;
  }

  u32x4 rx[4], rw[WROWS];
  const u32x4 zero4 = {0u, 0u, 0u, 0u};

  auto load_tile = [&](int kt) {
    const int k0 = kt * BK;
    if constexpr (!CONV) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (p.a2 != nullptr && k0 >= p.k_split) rx[i] = (xoff[i] >= 0) ? *reinterpret_cast<const u32x4*>(p.a2 + xoff2[i] + (k0 - p.k_split)) : zero4;
        else rx[i] = (xoff[i] >= 0) ? *reinterpret_cast<const u32x4*>(p.a + xoff[i] + k0) : zero4;
      }
    } else {
      const int tap = k0 / p.Cin;
      const int c0 = k0 - tap * p.Cin;
      const int dy = tap / 3 - 1;
      const int dx = tap - (tap / 3) * 3 - 1;
      const int Hv = p.Hin << p.up, Wv = p.Win << p.up;
      const int P = p.corner_patch;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int iy = cy[i] + dy;
        const int ix = cx[i] + dx;
        if (P > 0 && dy != 0 && dx != 0) {
          // halo-corner rule of the reference's sliced path (norm_silu_concat.cu:210-221, 228-239)
          const bool cross_r = ((iy + P) / P) != ((cy[i] + P) / P);
          const bool cross_c = ((ix + P) / P) != ((cx[i] + P) / P);
          if (cross_r && cross_c) iy = cy[i];
        }
        const bool ok = (cb[i] >= 0) && (iy >= -p.vhalo) && (iy < Hv + p.vhalo) && (ix >= 0) && (ix < Wv);
        if (ok) {
          const long src = (((long)cb[i] * (p.Hin + 2 * p.vhalo) + (iy >> p.up) + p.vhalo) * p.Win + (ix >> p.up)) * p.Cin + c0 + sch * 8;
          rx[i] = *reinterpret_cast<const u32x4*>(p.a + src);
        } else {
          rx[i] = zero4;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < WROWS; ++i)
      rw[i] = (woff[i] >= 0) ? *reinterpret_cast<const u32x4*>(p.w + woff[i] + k0) : zero4;
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = srow + 32 * i;
      *reinterpret_cast<u32x4*>(&sX[buf][row * BK + swz(row, sch) * 8]) = rx[i];
    }
#pragma unroll
    for (int i = 0; i < WROWS; ++i) {
      const int row = srow + 32 * i;
      *reinterpret_cast<u32x4*>(&sW[buf][row * BK + swz(row, sch) * 8]) = rw[i];
    }
  };

  f32x4 acc[NI][MI];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < MI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15;  // row inside a 16-row fragment
  const int fq = lane >> 4;  // 16-byte chunk inside the 32-deep k-step

  load_tile(0);
  store_tile(0);
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) load_tile(kt + 1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 wf[NI], xf[MI];
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int row = wn * (BN / 2) + i * 16 + fr;
        wf[i] = *reinterpret_cast<const bf16x8*>(&sW[buf][row * BK + swz(row, ks * 4 + fq) * 8]);
      }
#pragma unroll
      for (int j = 0; j < MI; ++j) {
        const int row = wm * 64 + j * 16 + fr;
        xf[j] = *reinterpret_cast<const bf16x8*>(&sX[buf][row * BK + swz(row, ks * 4 + fq) * 8]);
      }
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < MI; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) store_tile(buf ^ 1);
    __syncthreads();
  }

  gemm_epilogue<NI, MI, BN>(p, acc, m0 + wm * 64, n0 + wn * (BN / 2), fr, fq);
}

int launch_v2(hipStream_t s, const GemmArgs& a, bool conv, int bn, int rows);
int launch_v5(hipStream_t s, const GemmArgs& a, bool conv, int bn, int rows);
int launch_v4(hipStream_t s, const GemmArgs& a);
bool small_m_serves(const mx_gemm_desc* d, bool conv);      // gemm_small_m.hip: M <= 16, the weight-stream form
int launch_small_m(hipStream_t s, const GemmArgs& a);
bool conv_small_n_serves(const mx_gemm_desc* d);                // conv_small_n.hip: 3x3 conv with N <= 16 output channels (conv_out)
int launch_conv_small_n(hipStream_t s, const GemmArgs& a);
bool conv_small_cin_serves(const mx_gemm_desc* d);              // conv_small_n.hip: 3x3 conv over <= 8 non-zero input channels (conv_in)
int launch_conv_small_cin(hipStream_t s, const GemmArgs& a);

// Tile choice for the pipelined kernels.  Candidates (token rows x features): 256x256 (gemm_bf16_v3.hip), 256x160, 256x128,
// 128x160, 128x128 (gemm_bf16_v2.hip).  Estimated cost = full-chip rounds of 256 workgroups (one per CU) x (rows + features):
// the K loop of a tile is held by the CU's L2 -> LDS fetch stream, whose bytes per K tile are (rows + features) * 128.  A small
// problem therefore prefers small tiles (more CUs fetch in parallel: one 1024 px request gives M = 2048), a chip-filling one
// the tiling with the fewest rounds and the largest tile (fewest bytes per FLOP; the 256x256 kernel is further discounted by
// its measured advantage).  rows == 0: use the generic 128-row kernel.
constexpr size_t kSplitKWsBytes = 96u << 20;     // split-K scratch per stream (below): fp32 partial tiles / arrival tickets
constexpr size_t kSplitKCntBytes = 64u << 10;
struct TileChoice { int bn; int rows; int splitk = 1; };
// m-tiles of the launch for tiles of `rows` rows: the problems of a grouped launch are tiled one by one (no tile straddles two of them)
static long m_tiles_of(const mx_gemm_desc* d, int rows) {
  if (d->n_segs <= 0) return cdiv(d->M, rows);
  long t = 0;
  for (int i = 0; i < d->n_segs; ++i) t += cdiv(d->segs[i].M, rows);
  return t;
}
static long rows_of(const mx_gemm_desc* d) {
  if (d->n_segs <= 0) return d->M;
  long m = 0;
  for (int i = 0; i < d->n_segs; ++i) m += d->segs[i].M;
  return m;
}
static TileChoice pick_tile(const mx_gemm_desc* d, bool conv) {
  constexpr double v3_discount = 0.87;          // measured advantage of the 256 x 256 ping-pong kernel per byte fetched (round 1 A/B sweeps)
  const TileChoice none = {0, 0};
  const long Mtot = rows_of(d);
  if (Mtot < 128 || d->K < 128) return none;
  if (d->flags & (MX_EPI_OUT_F32 | MX_EPI_RES_BCAST)) return none;   // the register-exchange epilogue writes bf16 only and adds a per-row residual
  // its row walk steps 16 tokens at a time with one wrap per step (gemm_epilogue_regs): batches shorter than that go to the generic kernel
  if (d->n_segs <= 0) { if (d->rows_per_batch > 0 && d->rows_per_batch < 16) return none; }
  else for (int i = 0; i < d->n_segs; ++i) if (d->segs[i].rows_per_batch > 0 && d->segs[i].rows_per_batch < 16) return none;
  // their LDS-staged epilogue moves 16-byte pieces of C and of the residual
  if (d->ldc % 8 != 0 || ((uintptr_t)d->c & 15) != 0) return none;
  if (d->residual && (d->ldr % 8 != 0 || ((uintptr_t)d->residual & 15) != 0)) return none;
  const bool geglu = (d->flags & MX_EPI_GEGLU) != 0, qkv = (d->flags & MX_EPI_QKV) != 0;
  // the 256x256 kernel addresses its operands with 32-bit byte offsets from the base pointers (grouped: from the lowest problem base)
  bool fits32 = (long)d->N * d->K * 2 < (1L << 32);
  if (d->n_segs <= 0) {
    const long in_rows = d->a_batch_rows > 0 ? (long)(d->M / d->rows_per_batch + 1) * d->a_batch_rows : d->M;
    fits32 = fits32 && in_rows * d->lda * 2 < (1L << 32);
  } else {
    uintptr_t lo = (uintptr_t)d->segs[0].a;
    for (int i = 1; i < d->n_segs; ++i) lo = std::min(lo, (uintptr_t)d->segs[i].a);
    for (int i = 0; i < d->n_segs; ++i) {
      const mx_gemm_seg& g = d->segs[i];
      const long in_rows = g.a_batch_rows > 0 ? (long)(g.M / std::max(g.rows_per_batch, 1) + 1) * g.a_batch_rows : g.M;
      fits32 = fits32 && (long)((uintptr_t)g.a - lo) + in_rows * d->lda * 2 < (1L << 32);
    }
  }
  TileChoice best = none;
  double best_cost = 0;
  const TileChoice cands[5] = {{256, 256}, {160, 256}, {128, 256}, {160, 128}, {128, 128}};
  for (int c = 0; c < 5; ++c) {
    const int bn = cands[c].bn, rows = cands[c].rows;
    if (d->N % bn != 0 || Mtot < rows) continue;
    if (bn == 256 && (conv || !fits32 || d->a2 || d->ln_stats || d->stats_out)) continue;   // (built without those hooks)
    if (d->ln_final && bn != 256) continue;   // finalised statistics are the 256 x 256 kernel's form of the fold (the others read the slabs)
    if (geglu && bn == 160) continue;
    if (qkv && d->seg % 64 != 0) continue;
    if ((d->flags & MX_EPI_RMSNORM) && bn == 160) continue;   // a 64-wide head must lie inside one wave panel (gemm_epilogue_regs)
    if (qkv && bn != 256 && d->seg % (bn / 2) != 0) continue;
    const long tiles = m_tiles_of(d, rows) * (d->N / bn);
    const int ncu = cu_count();
    const double cost = (double)((tiles + ncu - 1) / ncu) * (rows + bn) * (bn == 256 ? v3_discount : 1.0);
    if (best.rows == 0 || cost < best_cost) { best = cands[c]; best_cost = cost; }
  }
  // Small launches (the 128-row tiles: one request, light mixed batches) leave CUs idle and run a long serial K loop whose iteration cannot be
  // shorter than the CU's LDS-DMA issue allows (0.55-0.8 us per 128 x 128 x 64 tile whatever the ring depth).  SPLIT-K deals the K tiles of an
  // output tile to `splitk` workgroups (gemm_bf16_v2.hip, splitk_combine).  What it costs was measured (round 4, tools/exp/splitk_bench.py,
  // profiles/r04_g_splitk_bench.txt): the fp32 partial tiles travel through memory -- slices x M x N x 4 bytes written through and read back
  // -- so M 2048, N 1280 in two slices moves 42 MB and the combine takes ~10 us: K 1280 got SLOWER (16.4 -> 21.2 us), K 5120 5 % faster
  // (43.3 -> 41.0), M 512 5 % faster.  The estimate below therefore charges that traffic at 4 TB/s and a split is taken only where it still
  // wins by 25 %: long K at small M x N (the convs and ff.net.2 of a single 512 px request: M 512).  Slicing K also changes the order in
  // which a row's products are added, so a split launch is not bit-equal to the unsplit one (every unsplit tiling is): the margin keeps the
  // marginal cases on the order that does not depend on what else shares the batch.
  if (d->splitk != 1 && best.rows == 128 && !d->a2 && d->K / 64 >= 16) {
    const int ncu = cu_count();
    const int nk = d->K / 64;
    double best_t = 0, unsplit_t = 0;
    TileChoice pick = best;
    const double mtot = (double)Mtot;
    for (int c = 3; c < 5; ++c) {
      const int bn = cands[c].bn;
      if (d->N % bn != 0 || (geglu && bn == 160) || (qkv && (d->seg % 64 != 0 || d->seg % (bn / 2) != 0)) || ((d->flags & MX_EPI_RMSNORM) && bn == 160)) continue;
      const long tiles = m_tiles_of(d, 128) * (d->N / bn);
      for (int sk = 1; sk <= 4; ++sk) {
        if (d->splitk > 1 && sk != 1 && sk != d->splitk) continue;            // a forced slice count (tests, A/B)
        if (nk / sk < 8 || (sk > 1 && tiles * sk > 2L * ncu)) continue;
        if (sk > 1 && (tiles * sk * 128L * bn * 4 > (long)kSplitKWsBytes || tiles * 4 > (long)kSplitKCntBytes)) continue;   // the partial tiles and tickets must fit the library's scratch
        const double combine = sk > 1 ? 2.0 + (double)sk * mtot * d->N * 8.0 / 4.0e6 : 0.0;      // us: partial tiles out and back at ~4 TB/s
        const double t = (double)((tiles * sk + ncu - 1) / ncu) * ((double)(nk / sk) * 0.57 * (128 + bn) / 256.0 + 5.0) + combine;
        if (sk == 1 && bn == best.bn) unsplit_t = t;
        if (best_t == 0 || t < best_t - 1e-9) { best_t = t; pick = TileChoice{bn, 128, sk}; }
      }
    }
    if (pick.splitk > 1 && unsplit_t > 0 && best_t <= 0.75 * unsplit_t) best = pick;
    if (d->splitk > 1) {                       // forced: the cheapest eligible tiling with that many slices
      double ft = 0;
      for (int c = 3; c < 5; ++c) {
        const int bn = cands[c].bn, sk = d->splitk;
        if (d->N % bn != 0 || (geglu && bn == 160) || (qkv && (d->seg % 64 != 0 || d->seg % (bn / 2) != 0)) || ((d->flags & MX_EPI_RMSNORM) && bn == 160)) continue;
        const long tiles = m_tiles_of(d, 128) * (d->N / bn);
        if (sk > 4 || nk / sk < 8 || tiles * sk * 128L * bn * 4 > (long)kSplitKWsBytes || tiles * 4 > (long)kSplitKCntBytes) continue;
        const double t = (double)((tiles * sk + ncu - 1) / ncu) * ((double)(nk / sk) * 0.57 * (128 + bn) / 256.0 + 5.0);
        if (ft == 0 || t < ft) { ft = t; best = TileChoice{bn, 128, sk}; }
      }
    }
  }
  return best;
}

// scratch of the split-K launches: fp32 partial tiles and one arrival counter per output tile, per stream (launches of one stream are ordered;
// concurrent streams -- the per-resolution sequences of a mixed batch -- must not share them).  Allocated at the first split launch of a stream,
// ALSO while that stream is being captured (advisor, round 4: a capture used to bake in the unsplit kernels, so eager and replayed forwards of one
// shape added their products in different orders): the allocation runs with the thread's capture mode relaxed and zeroes the counters on a private
// stream, neither of which touches the capturing stream.  Whether a launch is split therefore depends on its descriptor alone (pick_tile); a scratch
// that cannot be had is an error, not a silent change of summation order.  mx_gemm_release_scratch frees a stream's scratch (library unload frees all).
struct SplitKScratch { float* ws = nullptr; unsigned* cnt = nullptr; };
struct SplitKPool {
  std::mutex mu;
  std::unordered_map<hipStream_t, SplitKScratch> per_stream;
  static void drop(SplitKScratch& b) { if (b.ws) (void)hipFree(b.ws); if (b.cnt) (void)hipFree(b.cnt); b = SplitKScratch{}; }
  ~SplitKPool() { for (auto& kv : per_stream) drop(kv.second); }
};
static SplitKPool& splitk_pool() { static SplitKPool p; return p; }
static bool splitk_scratch(hipStream_t s, SplitKScratch& out) {
  SplitKPool& pool = splitk_pool();
  std::lock_guard<std::mutex> lock(pool.mu);
  auto it = pool.per_stream.find(s);
  if (it != pool.per_stream.end()) { out = it->second; return out.ws != nullptr; }
  hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
  (void)hipThreadExchangeStreamCaptureMode(&mode);
  SplitKScratch b;
  hipStream_t z = nullptr;
  bool ok = hipMalloc(&b.ws, kSplitKWsBytes) == hipSuccess && hipMalloc(&b.cnt, kSplitKCntBytes) == hipSuccess &&
            hipStreamCreateWithFlags(&z, hipStreamNonBlocking) == hipSuccess && hipMemsetAsync(b.cnt, 0, kSplitKCntBytes, z) == hipSuccess &&
            hipStreamSynchronize(z) == hipSuccess;
  if (z) (void)hipStreamDestroy(z);
  (void)hipThreadExchangeStreamCaptureMode(&mode);
  if (!ok) { (void)hipGetLastError(); SplitKPool::drop(b); return false; }      // (not remembered: a later launch may find memory)
  pool.per_stream[s] = b;
  out = b;
  return true;
}
static void splitk_release(hipStream_t s, bool all) {
  SplitKPool& pool = splitk_pool();
  std::lock_guard<std::mutex> lock(pool.mu);
  if (all) { for (auto& kv : pool.per_stream) { (void)hipStreamSynchronize(kv.first); SplitKPool::drop(kv.second); } pool.per_stream.clear(); return; }
  auto it = pool.per_stream.find(s);
  if (it == pool.per_stream.end()) return;
  (void)hipStreamSynchronize(s);
  SplitKPool::drop(it->second);
  pool.per_stream.erase(it);
}

// slabs of row statistics the launch of d writes: one per wave column panel of the register-exchange epilogue (gemm_epilogue_regs);
// 0 when the generic kernel serves d or the epilogue is not a plain bf16 store
static int stats_slabs_of(const mx_gemm_desc* d, bool conv, const TileChoice& tc) {
  if (conv || tc.bn == 0 || tc.bn == 256) return 0;     // (the 256 x 256 kernels are built without it: asking for stats_out moves the launch
                                                        //  to a 256 / 128-row tile, see pick_tile)
  if (d->flags & (MX_EPI_GEGLU | MX_EPI_QKV | MX_EPI_OUT_F32)) return 0;
  if (d->a_batch_rows > 0 || d->c_batch_rows > 0) return 0;
  for (int i = 0; i < d->n_segs; ++i) if (d->segs[i].a_batch_rows > 0 || d->segs[i].c_batch_rows > 0) return 0;
  const int panel = tc.bn / 2;                          // 4 x 2 waves of (16 MI) x (BN / 2)
  return d->N / panel;
}

// validation + the kernel argument block + the tile choice of mx_gemm / mx_conv3x3 (d); the dispatch follows in launch()
static int prepare(void* stream, const mx_gemm_desc* d, bool conv, GemmArgs& a, TileChoice& tc_out) {
  MX_CHECK(d != nullptr, "gemm: null descriptor");
  MX_CHECK(d->a && d->w && (d->c || (d->flags & MX_EPI_QKV)), "gemm: null operand");
  MX_CHECK(d->n_segs >= 0 && d->n_segs <= MX_MAX_SEGS && (d->n_segs == 0 || d->segs != nullptr), "gemm: bad n_segs / segs");
  MX_CHECK((d->n_segs > 0 || d->M > 0) && d->N > 0 && d->K > 0, "gemm: empty problem");
  for (int i = 0; i < d->n_segs; ++i) {          // a problem of a grouped launch has exactly the operands the descriptor names
    const mx_gemm_seg& g = d->segs[i];
    MX_CHECK(g.M > 0 && g.a && (g.c != nullptr) == (d->c != nullptr), "gemm: grouped launch: empty problem or missing a / c");
    MX_CHECK((g.a2 != nullptr) == (d->a2 != nullptr && !conv) && (g.residual != nullptr) == (d->residual != nullptr) && (g.vt != nullptr) == (d->vt != nullptr) &&
             (g.rowbias != nullptr) == (d->rowbias != nullptr) && (g.gate != nullptr) == (d->gate != nullptr) &&
             (g.ln_stats != nullptr) == (d->ln_stats != nullptr) && (g.stats_out != nullptr) == (d->stats_out != nullptr),
             "gemm: grouped launch: a problem's optional operands must match the descriptor's");
    const void* ptrs[] = {g.a, g.a2, g.c, g.residual, g.vt, g.rowbias, g.gate, g.ln_stats, g.stats_out};
    for (const void* q : ptrs) MX_CHECK(((uintptr_t)q & 15) == 0, "gemm: grouped launch: operand pointers must be 16-byte aligned");
    if (d->rowbias || d->gate || g.a_batch_rows > 0 || g.c_batch_rows > 0 || (d->flags & (MX_EPI_QKV | MX_EPI_RES_BCAST)))
      MX_CHECK(g.rows_per_batch > 0, "gemm: grouped launch: rows_per_batch required");
    if (g.a_batch_rows > 0) MX_CHECK(!conv && g.a_row_off >= 0 && g.a_row_off + g.rows_per_batch <= g.a_batch_rows, "gemm: grouped launch: bad input row remap");
    if (g.c_batch_rows > 0) MX_CHECK(g.c_row_off >= 0 && g.c_row_off + g.rows_per_batch <= g.c_batch_rows, "gemm: grouped launch: bad output row remap");
    if (d->ln_stats || d->stats_out)           // (advisor, round 3: the per-problem remaps were not covered by the descriptor-level check)
      MX_CHECK(g.a_batch_rows <= 0 && g.c_batch_rows <= 0, "gemm: grouped launch: the folded LayerNorm / stats_out exclude a problem's row remaps");
    if (d->flags & MX_EPI_QKV)
      MX_CHECK(g.M % g.rows_per_batch == 0 && g.ldvt >= MX_VT_LD(g.c_batch_rows > 0 ? g.c_batch_rows : g.rows_per_batch) && g.ldvt % 8 == 0,
               "gemm: grouped launch: QKV needs whole batches and ldvt >= MX_VT_LD(keys per batch)");
    if (conv) {
      const int Hv = g.Hin << d->up, Wv = g.Win << d->up;
      MX_CHECK(g.Hout == (Hv + d->stride - 1) / d->stride && g.Wout == (Wv + d->stride - 1) / d->stride && (long)g.B * g.Hout * g.Wout == g.M,
               "conv3x3: grouped launch: a problem's output grid does not match its input grid / stride / rows");
    }
    MX_CHECK((long)g.M * (conv ? 1 : d->lda) < 2147483647L, "gemm: grouped launch: operand exceeds 32-bit indexing");
  }
  MX_CHECK(d->K % BK == 0, "gemm: K must be a multiple of 64");
  MX_CHECK(d->N % 4 == 0, "gemm: N must be a multiple of 4");
  a.stagger_ticks = 0;
  a.vhalo = conv ? d->vhalo : 0;
  a.a2 = conv ? nullptr : (const bf16_t*)d->a2; a.lda2 = d->lda2; a.k_split = d->k_split;
  a.a = (const bf16_t*)d->a; a.w = (const bf16_t*)d->w; a.c = d->c;
  a.bias = d->bias; a.rowbias = d->rowbias; a.residual = (const bf16_t*)d->residual; a.vt = (bf16_t*)d->vt;
  a.M = d->M; a.N = d->N; a.K = d->K; a.lda = d->lda; a.ldc = d->ldc; a.ldr = d->ldr; a.ldrb = d->ldrb;
  a.rows_per_batch = d->rows_per_batch; a.flags = d->flags; a.seg = d->seg; a.period = d->period; a.ldvt = d->ldvt;
  a.B = d->B; a.Hin = d->Hin; a.Win = d->Win; a.Cin = d->Cin; a.Hout = d->Hout; a.Wout = d->Wout;
  a.stride = d->stride; a.up = d->up; a.corner_patch = d->corner_patch;
  a.a_batch_rows = d->a_batch_rows; a.a_row_off = d->a_row_off; a.c_batch_rows = d->c_batch_rows; a.c_row_off = d->c_row_off;
  a.gate = d->gate; a.ldg = d->ldg; a.out_scale = d->out_scale;
  a.rms_wq = d->rms_wq; a.rms_wk = d->rms_wk; a.rms_eps = d->rms_eps;
  a.ln_stats = d->ln_stats; a.ln_colsum = d->ln_colsum; a.ln_slabs = d->ln_slabs; a.ln_eps = d->ln_eps; a.stats_out = nullptr;
  a.ln_final = nullptr; a.ln_final_out = nullptr; a.ln_final_cnt = nullptr; a.ln_final_slabs = 0;
  if (d->ln_final) {
    MX_CHECK(!conv && !d->ln_stats && d->ln_colsum && d->n_segs == 0, "gemm: ln_final needs ln_colsum, excludes ln_stats and grouped launches (mx_gemm only)");
    MX_CHECK(!(d->flags & MX_EPI_RMSNORM) && d->a_batch_rows <= 0 && d->c_batch_rows <= 0 && !d->a2 && !d->rowbias && !d->gate && !d->stats_out,
             "gemm: ln_final excludes RMSNORM, the row remaps, the split A operand, per-sample vectors and stats_out");
    MX_CHECK((((uintptr_t)d->ln_final & 15) | ((uintptr_t)d->ln_colsum & 15)) == 0, "gemm: ln_final / ln_colsum must be 16-byte aligned");
    a.ln_final = d->ln_final;
  }
  a.nseg = d->n_segs; a.mt_total = 0;
  const bool grouped = d->n_segs > 0;
  if (d->ln_stats) {
    MX_CHECK(!conv && d->ln_colsum && d->ln_slabs > 0, "gemm: folded LayerNorm needs ln_colsum and ln_slabs > 0 (mx_gemm only)");
    MX_CHECK(!(d->flags & MX_EPI_RMSNORM) && d->a_batch_rows <= 0 && d->c_batch_rows <= 0 && !d->a2, "gemm: folded LayerNorm excludes RMSNORM, the row remaps and the split A operand");
    MX_CHECK((((uintptr_t)d->ln_stats & 15) | ((uintptr_t)d->ln_colsum & 15)) == 0, "gemm: ln_stats / ln_colsum must be 16-byte aligned");
  }
  a.xcd_map = 1;

  if (!conv && d->a2) {
    MX_CHECK(d->k_split > 0 && d->k_split < d->K && d->k_split % BK == 0, "gemm: k_split must be a multiple of 64 inside (0, K)");
    MX_CHECK(d->lda >= d->k_split && d->lda % 8 == 0 && d->lda2 >= d->K - d->k_split && d->lda2 % 8 == 0, "gemm: bad lda / lda2 for the split A operand");
    MX_CHECK(d->a_batch_rows <= 0 && ((uintptr_t)d->a2 & 15) == 0 && rows_of(d) * d->lda2 < 2147483647L, "gemm: split A operand excludes the row remap and needs 16-byte alignment");
  } else if (!conv) {
    MX_CHECK(d->lda >= d->K && d->lda % 8 == 0, "gemm: lda must be >= K and a multiple of 8");
  } else {
    MX_CHECK(d->Cin % BK == 0 && d->K == 9 * d->Cin, "conv3x3: Cin must be a multiple of 64 and K = 9*Cin");
    MX_CHECK(d->stride == 1 || d->stride == 2, "conv3x3: stride must be 1 or 2");
    MX_CHECK(d->up == 0 || d->up == 1, "conv3x3: up must be 0 or 1");
    MX_CHECK(!(d->up && d->stride != 1), "conv3x3: upsample only with stride 1");
    if (!grouped) {
      const int Hv = d->Hin << d->up, Wv = d->Win << d->up;
      MX_CHECK(d->Hout == (Hv + d->stride - 1) / d->stride && d->Wout == (Wv + d->stride - 1) / d->stride,
               "conv3x3: output grid does not match input grid / stride");
      MX_CHECK((long)d->B * d->Hout * d->Wout == d->M, "conv3x3: M != B*Hout*Wout");
    }
    MX_CHECK(2 * d->Cin <= 16384, "conv3x3: Cin > 8192 (the pipelined loader walks a 16 KB zero page for padding taps)");
    MX_CHECK(!(d->flags & MX_EPI_GEGLU), "conv3x3: no GEGLU epilogue");
    MX_CHECK(d->vhalo == 0 || (d->vhalo == 1 && d->corner_patch == 0), "conv3x3: vhalo must be 0 or 1 and excludes the sliced corner rule");
  }
  // the LDS-DMA loaders and the staged epilogue move 16-byte pieces: every base pointer must be 16-byte aligned
  {
    const void* ptrs[] = {d->a, d->w, d->c, d->bias, d->rowbias, d->residual, d->gate, d->rms_wq, d->rms_wk};
    for (const void* q : ptrs) MX_CHECK(((uintptr_t)q & 15) == 0, "gemm: operand pointers must be 16-byte aligned");
  }
  if (!grouped && (d->rowbias || d->gate || d->a_batch_rows > 0 || d->c_batch_rows > 0 || (d->flags & (MX_EPI_QKV | MX_EPI_RES_BCAST))))
    MX_CHECK(d->rows_per_batch > 0, "gemm: rows_per_batch required");
  if (d->gate) MX_CHECK(d->ldg >= d->N && d->ldg % 4 == 0, "gemm: bad ldg");
  if (!grouped && d->a_batch_rows > 0) MX_CHECK(!conv && d->a_row_off >= 0 && d->a_row_off + d->rows_per_batch <= d->a_batch_rows, "gemm: bad input row remap");
  if (!grouped && d->c_batch_rows > 0) MX_CHECK(d->c_row_off >= 0 && d->c_row_off + d->rows_per_batch <= d->c_batch_rows, "gemm: bad output row remap");
  if (d->rowbias) MX_CHECK(d->ldrb >= d->N && d->ldrb % 4 == 0, "gemm: bad ldrb");
  if (!grouped) {
    const long in_rows = (!conv && d->a_batch_rows > 0) ? (long)(d->M / d->rows_per_batch + 1) * d->a_batch_rows : d->M;
    MX_CHECK(in_rows * (conv ? 1 : d->lda) < 2147483647L, "gemm: operand exceeds 32-bit indexing");
  }
  MX_CHECK((long)d->N * d->K < 2147483647L, "gemm: operand exceeds 32-bit indexing");
  if (d->residual) MX_CHECK(d->ldr >= d->N && d->ldr % 4 == 0, "gemm: bad ldr");
  const bool use128 = (d->N % 128 == 0);
  const TileChoice tc = pick_tile(d, conv);
  tc_out = tc;
  a.splitk = 0; a.sk_ws = nullptr; a.sk_cnt = nullptr;
  if (tc.splitk > 1) {
    SplitKScratch sk;
    MX_CHECK(splitk_scratch((hipStream_t)stream, sk), "gemm: the split-K scratch (96 MB per stream) could not be allocated; set mx_gemm_desc.splitk = 1 to run unsplit");
    a.splitk = tc.splitk; a.sk_ws = sk.ws; a.sk_cnt = sk.cnt;
  }
  if (d->stats_out) {
    MX_CHECK(stats_slabs_of(d, conv, tc) > 0, "gemm: stats_out is not supported for this shape / epilogue (see mx_gemm_stats_slabs)");
    MX_CHECK(((uintptr_t)d->stats_out & 15) == 0, "gemm: stats_out must be 16-byte aligned");
    a.stats_out = d->stats_out;
  }
  a.gn_part = nullptr;
  if (d->gn_part_out) {
    MX_CHECK(!grouped && tc.rows == 256 && tc.bn != 256 && tc.bn != 0 && tc.splitk <= 1, "gemm: gn_part_out needs an ungrouped launch on a 256-row tile (mx_gemm_gn_partials_supported)");
    MX_CHECK(d->flags == 0 && !d->residual && !d->gate && d->out_scale == 0.f && !d->ln_stats && !d->ln_final && d->a_batch_rows <= 0 && d->c_batch_rows <= 0,
             "gemm: gn_part_out needs an epilogue of bias (+ row bias) only");
    MX_CHECK(d->M % 64 == 0 && (!d->rowbias || d->rows_per_batch % 64 == 0) && ((uintptr_t)d->gn_part_out & 15) == 0, "gemm: gn_part_out needs M % 64 == 0, rows_per_batch % 64 == 0 and 16-byte alignment");
    a.gn_part = d->gn_part_out;
  }
  if (d->ln_final_out) {
    MX_CHECK(d->stats_out && d->ln_final_cnt && !grouped && tc.rows == 256 && tc.bn != 256 && tc.bn != 0,
             "gemm: ln_final_out needs stats_out, ln_final_cnt and an ungrouped launch on a 256-row tile (mx_gemm_ln_final_supported)");
    MX_CHECK((((uintptr_t)d->ln_final_out & 15) | ((uintptr_t)d->ln_final_cnt & 3)) == 0, "gemm: ln_final_out must be 16-byte aligned");
    a.ln_final_out = d->ln_final_out; a.ln_final_cnt = d->ln_final_cnt; a.ln_final_slabs = stats_slabs_of(d, conv, tc);
  }
  if (grouped) {
    // the problems' tiles follow each other in the launch's tile list; the kernel argument's own a is the lowest problem base (the 256 x 256
    // kernel addresses A by 32-bit offsets from it: pick_tile checked the reach)
    const int rows = tc.rows > 0 ? tc.rows : BM;
    uintptr_t lo = (uintptr_t)d->segs[0].a;
    int t0 = 0;
    for (int i = 0; i < d->n_segs; ++i) {
      const mx_gemm_seg& g = d->segs[i];
      GemmSeg& o = a.prob[i];
      o.a = (const bf16_t*)g.a; o.a2 = conv ? nullptr : (const bf16_t*)g.a2; o.c = g.c; o.residual = (const bf16_t*)g.residual; o.vt = (bf16_t*)g.vt;
      o.rowbias = g.rowbias; o.gate = g.gate; o.ln_stats = g.ln_stats; o.stats_out = d->stats_out ? g.stats_out : nullptr;
      o.M = g.M; o.tile0 = t0; o.rows_per_batch = g.rows_per_batch; o.ldvt = g.ldvt;
      o.B = g.B; o.Hin = g.Hin; o.Win = g.Win; o.Hout = g.Hout; o.Wout = g.Wout;
      o.a_batch_rows = g.a_batch_rows; o.a_row_off = g.a_row_off; o.c_batch_rows = g.c_batch_rows; o.c_row_off = g.c_row_off;
      t0 += cdiv(g.M, rows);
      lo = std::min(lo, (uintptr_t)g.a);
    }
    a.mt_total = t0;
    a.a = (const bf16_t*)lo;
    a.M = (int)rows_of(d);
  }
  if (d->flags & MX_EPI_GEGLU) {
    MX_CHECK(use128, "gemm: GEGLU needs N % 128 == 0");
    MX_CHECK(!(d->flags & (MX_EPI_QKV | MX_EPI_OUT_F32)) && !d->residual && !d->rowbias && d->out_scale == 0.f, "gemm: GEGLU excludes other epilogues");
    MX_CHECK(d->ldc >= d->N / 2 && d->ldc % 4 == 0, "gemm: bad ldc for GEGLU");
  } else if (d->flags & MX_EPI_QKV) {
    MX_CHECK(d->seg > 0 && d->seg % 64 == 0 && d->period >= 2 && d->N % (d->seg * d->period) == 0, "gemm: bad QKV segments");
    MX_CHECK(d->vt != nullptr, "gemm: QKV needs vt");
    if (!grouped) {
      MX_CHECK(d->ldvt >= MX_VT_LD(d->c_batch_rows > 0 ? d->c_batch_rows : d->rows_per_batch), "gemm: QKV needs ldvt >= MX_VT_LD(keys per batch)");
      MX_CHECK(d->M % d->rows_per_batch == 0, "gemm: QKV needs M % rows_per_batch == 0");
    }
    MX_CHECK(d->ldc >= d->N / d->period * (d->period - 1) && d->ldc % 4 == 0, "gemm: bad ldc for QKV");
    MX_CHECK(!(d->flags & MX_EPI_OUT_F32), "gemm: QKV output is bf16");
    MX_CHECK(!d->rowbias && !d->gate && !d->residual, "gemm: QKV excludes the per-sample vectors and the residual (its V^T segment takes none of them)");
    if (d->flags & MX_EPI_RMSNORM)
      MX_CHECK(d->rms_wq && d->rms_wk && d->period == 3 && d->N % 128 == 0 && !conv, "gemm: RMSNORM needs rms_wq/rms_wk, period 3, N % 128 == 0");
  } else {
    MX_CHECK(d->ldc >= d->N && d->ldc % 4 == 0, "gemm: bad ldc");
  }
  return 0;
}

// the argument block of mx_gemm(d) for a launch that runs its tiles itself (attn_tail.hip): the same validation and tile choice
int gemm_prepare_for_chain(void* stream, const mx_gemm_desc* d, GemmArgs& a, int& bn, int& rows, int& splitk) {
  TileChoice tc;
  if (int rc = prepare(stream, d, false, a, tc)) return rc;
  bn = tc.bn; rows = tc.rows; splitk = tc.splitk;
  return 0;
}

static int launch(void* stream, const mx_gemm_desc* d, bool conv) {
  GemmArgs a;
  TileChoice tc;
  if (int rc = prepare(stream, d, conv, a, tc)) return rc;
  const bool grouped = d->n_segs > 0;
  const bool use128 = (d->N % 128 == 0);
  const int v2bn = tc.bn;
  dim3 block(256);
  hipStream_t s = (hipStream_t)stream;
  if (prof_enabled()) {
    // algorithmic work: true (unpadded) contraction; bytes = operands read once + result written once
    const double kk = conv ? 9.0 * d->Cin : (double)d->K;
    const double Mt = (double)rows_of(d);
    double in_elems = conv ? (double)d->B * d->Hin * d->Win * d->Cin : Mt * d->K;
    if (conv && grouped) { in_elems = 0; for (int i = 0; i < d->n_segs; ++i) in_elems += (double)d->segs[i].B * d->segs[i].Hin * d->segs[i].Win * d->Cin; }
    const double flops = 2.0 * Mt * (double)d->N * kk;
    const double bytes = 2.0 * (in_elems + (double)d->N * d->K + Mt * d->N);
    const int kind = v2bn == 256 ? PROF_GEMM_V3_256 : v2bn ? (conv ? PROF_CONV_V2_160 : PROF_GEMM_V2_160) + (v2bn == 160 ? 0 : 2) : (conv ? PROF_CONV128 : PROF_GEMM128) + (use128 ? 0 : 1);
    prof_begin(s, kind, flops, bytes, (int)Mt, d->N, (int)kk);
  }
  const int mt128 = grouped ? a.mt_total : cdiv(d->M, BM);      // m-tiles of the generic kernel
  if (d->ln_final) MX_CHECK(v2bn == 256, "gemm: ln_final is the 256 x 256 kernel's form of the folded LayerNorm; this shape does not run there (use ln_stats)");
  if (small_m_serves(d, conv)) {
    launch_small_m(s, a);                      // M <= 16: a weight stream (gemm_small_m.hip)
  } else if (conv && conv_small_n_serves(d)) {
    launch_conv_small_n(s, a);                 // N <= 16: the input read once (conv_small_n.hip)
  } else if (conv && conv_small_cin_serves(d)) {
    launch_conv_small_cin(s, a);               // <= 8 non-zero input channels: K = 72 (conv_small_n.hip)
  } else if (v2bn == 256) {
    MX_CHECK(launch_v4(s, a) == 0, "gemm: no 256 x 256 instantiation serves ln_final with this epilogue (GEGLU, QKV or plain bias only)");   // 256 x 256 ping-pong (gemm_bf16_v4.hip)
  } else if (v2bn) {
    // 256-row tiles: ping-pong schedule (gemm_bf16_v5.hip).  128-row tiles (small M) stay on the lock-step loop of gemm_bf16_v2.hip: the
    // ping-pong form is a tie there (same-box A/B, profiles/r03_d_gemm_bench_small_*: M2048 N1280 K1280 19.3 vs 19.2 us, conv B2 1280@32 98.9 vs
    // 108.5 us) -- with half the MFMAs per K tile its L phase (5 LDS-DMA issues + 14 fragment reads) outlasts the M phase
    if (tc.rows == 256) launch_v5(s, a, conv, v2bn, tc.rows);
    else launch_v2(s, a, conv, v2bn, tc.rows);
  } else if (use128) {
    dim3 grid(mt128, d->N / 128);
    if (conv) hipLaunchKernelGGL((gemm_kernel<128, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((gemm_kernel<128, false>), grid, block, 0, s, a);
  } else {
    dim3 grid(mt128, cdiv(d->N, 64));
    if (conv) hipLaunchKernelGGL((gemm_kernel<64, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((gemm_kernel<64, false>), grid, block, 0, s, a);
  }
  prof_end(s);
  MX_LAUNCH_CHECK();
  return 0;
}

}  // namespace mx

namespace mx {
// TAIL SPLIT (round 4).  The persistent 256 x 256 kernel walks whole rounds of one tile per CU; a launch whose tile count leaves a short last round
// (one 1024 px request: GEGLU M 2048 x N 10240 = 320 tiles = 1.25 rounds, 77 us for 1.25 rounds of work) pays a full round for it.  Where the tiles
// of the whole rounds are whole column panels, the launch is cut along N: columns [0, N1) keep the 256 x 256 kernel in whole rounds, the rest
// becomes a second launch on whatever tile the chooser gives it, accepted only if that is ONE round of a cheaper tile (128 x 128 / 128 x 160 /
// 256 x 128).  Columns are independent, so the results are those of the single launch bit for bit where both tilings add a row's products in
// the same order (every unsplit tiling does).  Plain and gated epilogues only (no QKV segments, statistics, fp32 output, grouped launches).
static bool tail_split(const mx_gemm_desc* d, mx_gemm_desc& d1, mx_gemm_desc& d2) {
  if (d->n_segs != 0 || d->M <= 0 || d->N <= 0 || d->K <= 0 || d->N % 256 != 0) return false;
  if (d->flags & (MX_EPI_QKV | MX_EPI_OUT_F32 | MX_EPI_RES_BCAST | MX_EPI_RMSNORM)) return false;
  if (d->stats_out || d->ln_stats || d->ln_final || d->ln_final_out || d->a_batch_rows > 0 || d->c_batch_rows > 0 || d->splitk > 1) return false;
  if (pick_tile(d, false).bn != 256) return false;
  const int ncu = cu_count();
  if (ncu <= 0) return false;
  const long mt = cdiv(d->M, 256), nt = d->N / 256, tiles = mt * nt;
  const long full = tiles / ncu, rem = tiles % ncu;
  if (full < 1 || rem == 0 || rem * 2 > ncu || (full * ncu) % mt != 0) return false;
  const long nt1 = full * ncu / mt;
  if (nt1 <= 0 || nt1 >= nt) return false;
  const int N1 = (int)nt1 * 256, N2 = d->N - N1;
  const bool geglu = (d->flags & MX_EPI_GEGLU) != 0;
  const long cofs = geglu ? N1 / 2 : N1;       // first output column of the second launch
  d1 = *d; d2 = *d;
  d1.N = N1; d2.N = N2;
  d1.splitk = 1; d2.splitk = 1;               // both halves add a row's products in the single launch's order (advisor, round 4)
  d2.w = (const char*)d->w + (size_t)N1 * d->K * 2;
  if (d->bias) d2.bias = d->bias + N1;
  d2.c = (char*)d->c + (size_t)cofs * 2;
  if (geglu && d->residual) return false;      // (the launcher rejects that pair anyway; its column offsets would differ)
  if (d->residual) d2.residual = (const char*)d->residual + (size_t)N1 * 2;
  if (d->rowbias) d2.rowbias = d->rowbias + N1;
  if (d->gate) d2.gate = d->gate + N1;
  const TileChoice t2 = pick_tile(&d2, false);
  if (t2.bn == 0 || t2.bn == 256 || t2.rows + t2.bn > 384) return false;
  if (m_tiles_of(&d2, t2.rows) * (N2 / t2.bn) > ncu) return false;
  return pick_tile(&d1, false).bn == 256;
}
}  // namespace mx

extern "C" int mx_gemm(void* stream, const mx_gemm_desc* d) {
  mx_gemm_desc d1, d2;
  if (d != nullptr && mx::tail_split(d, d1, d2)) { if (int rc = mx::launch(stream, &d1, false)) return rc; return mx::launch(stream, &d2, false); }
  return mx::launch(stream, d, false);
}
/* launches mx_gemm(d) issues: 2 where the tail split applies (tests, planning) */
extern "C" int mx_gemm_launches(const mx_gemm_desc* d) { mx_gemm_desc d1, d2; return d != nullptr && mx::tail_split(d, d1, d2) ? 2 : 1; }
extern "C" int mx_gemm_form(const mx_gemm_desc* d, int conv) {
  if (!d || mx::rows_of(d) <= 0 || d->N <= 0 || d->K <= 0) return MX_FORM_TILE_GENERIC;
  if (mx::small_m_serves(d, conv != 0)) return MX_FORM_SMALL_M;
  if (conv && mx::conv_small_n_serves(d)) return MX_FORM_CONV_SMALL_N;
  if (conv && mx::conv_small_cin_serves(d)) return MX_FORM_CONV_SMALL_CIN;
  const mx::TileChoice tc = mx::pick_tile(d, conv != 0);
  return tc.bn == 256 ? MX_FORM_PERSISTENT_256 : tc.bn == 0 ? MX_FORM_TILE_GENERIC : tc.rows == 256 ? MX_FORM_TILE_256 : MX_FORM_TILE_128;
}
extern "C" int mx_gemm_stats_slabs(const mx_gemm_desc* d) {
  if (!d || mx::rows_of(d) <= 0 || d->N <= 0 || d->K <= 0) return 0;
  mx_gemm_desc q = *d;
  if (!q.stats_out) q.stats_out = reinterpret_cast<float*>(16);      // the tile choice of the launch that asks for statistics
  return mx::stats_slabs_of(&q, false, mx::pick_tile(&q, false));
}
extern "C" int mx_gemm_ln_prefers_pass(const mx_gemm_desc* d) {
  if (!d || mx::rows_of(d) <= 0 || d->N <= 0 || d->K <= 0) return 0;
  mx_gemm_desc plain = *d;
  plain.ln_stats = nullptr; plain.stats_out = nullptr;
  return mx::pick_tile(&plain, false).bn == 256;
}
extern "C" int mx_gemm_gn_partials_supported(const mx_gemm_desc* d, int conv) {
  if (!d || d->n_segs > 0 || d->M <= 0 || d->N <= 0 || d->K <= 0 || d->M % 64 != 0) return 0;
  if (d->flags != 0 || d->residual || d->gate || d->out_scale != 0.f || d->ln_stats || d->ln_final || d->a_batch_rows > 0 || d->c_batch_rows > 0) return 0;
  if (d->rowbias && (d->rows_per_batch <= 0 || d->rows_per_batch % 64 != 0)) return 0;
  const mx::TileChoice tc = mx::pick_tile(d, conv != 0);
  return tc.rows == 256 && tc.bn != 256 && tc.bn != 0 && tc.splitk <= 1;
}
extern "C" int mx_gemm_ln_final_supported(const mx_gemm_desc* d) {
  if (!d || d->n_segs > 0 || d->M <= 0 || d->N <= 0 || d->K <= 0) return 0;
  mx_gemm_desc q = *d;
  if (!q.stats_out) q.stats_out = reinterpret_cast<float*>(16);      // (shape query: the chooser only looks at which operands exist)
  const mx::TileChoice tc = mx::pick_tile(&q, false);
  return tc.rows == 256 && tc.bn != 256 && tc.bn != 0 && mx::stats_slabs_of(&q, false, tc) > 0;
}
extern "C" int mx_conv3x3(void* stream, const mx_gemm_desc* d) { return mx::launch(stream, d, true); }
extern "C" void mx_gemm_release_scratch(void* stream, int all) { mx::splitk_release((hipStream_t)stream, all != 0); }
extern "C" int mx_gemm_splitk(const mx_gemm_desc* d, int conv) {
  if (!d || mx::rows_of(d) <= 0 || d->N <= 0 || d->K <= 0) return 0;
  return mx::pick_tile(d, conv != 0).splitk;
}
