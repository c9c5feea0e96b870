/*
 * mxdenoise -- MI355X (gfx950) native denoiser for the sduss/Mixfusion model slot.  C ABI.
 *
 * Plain pointers and sizes only (no torch types).  Every device pointer is a HIP device
 * address owned by the caller; every entry point takes an explicit hipStream_t (as void*),
 * launches asynchronously on it and never synchronises the device.  Return value: 0 on
 * success, non-zero on error (mx_last_error() gives the message for the calling thread).
 *
 * Reference interfaces replaced (paths relative to the sduss tree):
 *   inner boundary  esymred_mp.groupnorm / mock_groupnorm
 *                   sduss/model_executor/modules/kernels/norm_silu_concat.cpp:66-101
 *                   (called from modules/groupnorm.py:33,50,58)
 *   outer boundary  PatchUNet.forward            sduss/model_executor/modules/unet.py:205-530
 *                   (invoked once per step at
 *                    diffusers/pipelines/stable_diffusion_xl/pipeline_stable_diffusion_xl_esymred.py:369-380)
 *   step either side EulerDiscreteScheduler.batch_scale_model_input / batch_step
 *                   diffusers/schedulers/scheduling_euler_discrete.py:161-274 and the CFG combine
 *                   pipeline_stable_diffusion_xl_esymred.py:382-385
 */
#ifndef MXDENOISE_H
#define MXDENOISE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* element types of caller tensors */
enum { MX_F32 = 0, MX_F16 = 1, MX_BF16 = 2 };

const char* mx_last_error(void);
int mx_version(void);
/* Optional per-launch timing for bench.py's roofline leg: while enabled, every GEMM / conv / attention / norm launch
 * is bracketed by hipEvents on its own stream.  mx_profile_collect (after the stream is synchronised) fills
 * out[4 * MX_PROF_KINDS] = for kind in {gemm<128>, gemm<64>, conv3x3<128>, conv3x3<64>, attention, groupnorm,
 *           gemm_v2<160>, conv3x3_v2<160>, gemm_v2<128>, conv3x3_v2<128>, gemm 256x256, cross-attention (Lk <= 96)}:
 *           {launches, milliseconds, algorithmic flops, algorithmic bytes}. */
#define MX_PROF_KINDS 12
int mx_profile_enable(int on);
int mx_profile_collect(double* out);
/* per-launch records of the last collect: out[6*i..] = {kind, M, N, K, ms, flops}; returns the number written */
int mx_profile_records(double* out, int max_records);

/* ------------------------------------------------------------------------------------------
 * Inner boundary: drop-in for the reference's only native op (NCHW patch batches).
 *
 * mx_groupnorm_halo == esymred_mp.groupnorm (norm_silu_concat.cpp:75-101):
 *   per-(patch,group) moments -> cross-patch merge per latent (mean of means, mean of biased
 *   variances; norm_silu_concat.cu:361-386) -> y = x*(rstd*gamma) + (beta - rstd*gamma*mean)
 *   (no SiLU, cu:157-163) -> if padding: result placed in the interior of a zero-filled
 *   [N,C,H+2,W+2] tensor and each patch's edge rows/cols/corners scattered into its neighbours'
 *   halo cells (cu:164-241).  `cpg` is what the reference calls `group` (channels per group).
 *   x: [N,C,H,W] contiguous, y: [N,C,H+2p,W+2p] (p = padding?1:0), gamma/beta: [C] of `dtype`,
 *   latent_offset: int32[n_latents+1], patch_map: int32[N] (1-based latent index),
 *   padding_idx: int32[4N] = (top,left,bottom,right) neighbour patch index or -1.
 *   workspace: >= mx_groupnorm_halo_workspace_bytes(N, C, cpg) bytes of device scratch.
 *   Differences, on purpose: statistics are kept in fp32 (the reference rounds them to the
 *   tensor dtype, cpp:84-85); the merge is out of place (the reference's in-place merge races);
 *   no device synchronisation (reference: cudaDeviceSynchronize after every launch).
 *   The halo is GATHERED by the receiving plane through the inverse of padding_idx (receiver r's side d is written by the b with
 *   padding_idx[b][opposite(d)] == r) with the SENDER's statistics, so the result equals the reference's sender-driven scatter for every
 *   table in which a halo side has at most one writer -- symmetric adjacency (what split_sample produces) is NOT required; with two
 *   writers for one cell the reference's scatter races and the value here is that of one of them.  Entries must be -1 or in [0, N).
 *
 * mx_halo_only == esymred_mp.mock_groupnorm (cpp:66-74): the same scatter with identity values.
 * ------------------------------------------------------------------------------------------ */
size_t mx_groupnorm_halo_workspace_bytes(int N, int C, int cpg);
int mx_groupnorm_halo(void* stream, const void* x, const void* gamma, const void* beta, void* y,
                      int N, int C, int H, int W, int cpg, double eps, int padding,
                      const int32_t* latent_offset, int n_latents, const int32_t* patch_map,
                      const int32_t* padding_idx, int dtype, void* workspace);
int mx_halo_only(void* stream, const void* x, void* y, int N, int C, int H, int W,
                 const int32_t* padding_idx, int dtype);

/* ------------------------------------------------------------------------------------------
 * Building blocks of the step plan (NHWC / token-major bf16, fp32 accumulate).  Exported so the
 * parity tests can check each kernel against the oracle through the same ABI the plan uses.
 * ------------------------------------------------------------------------------------------ */

/* epilogue flags for mx_gemm / mx_conv3x3 */
enum {
  MX_EPI_SILU     = 1 << 0,  /* out = silu(v) */
  MX_EPI_GEGLU    = 1 << 1,  /* weight rows interleaved [32 hidden | 32 gate]; out[M, N/2] = h * gelu(g) */
  MX_EPI_OUT_F32  = 1 << 2,  /* C is fp32 instead of bf16 */
  MX_EPI_QKV      = 1 << 3,  /* column segments of width seg: segment s with (s % period) == period-1
                                is written transposed into vt[b][vcol][key]; others row-major, compacted.
                                Excludes rowbias / gate / residual (error otherwise) */
  MX_EPI_GELU_TANH = 1 << 4, /* out = gelu(v), tanh approximation (diffusers FeedForward "gelu-approximate") */
  MX_EPI_RES_BCAST = 1 << 5, /* residual row = output row modulo rows_per_batch (positional table broadcast over the batch) */
  MX_EPI_RMSNORM  = 1 << 6,  /* with MX_EPI_QKV: RMS-normalise every 64-wide head of the q and k segments (see rms_wq below) */
  MX_EPI_GELU     = 1 << 7,  /* out = gelu(v), exact (erf) form: the OpenCLIP text encoder's MLP */
  MX_EPI_QUICK_GELU = 1 << 8, /* out = v * sigmoid(1.702 v): the CLIP ViT-L text encoder's MLP */
  MX_EPI_GEGLU_TANH = 1 << 9 /* with MX_EPI_GEGLU: the gate takes the tanh form of GELU ("gelu_new": T5 v1.1 gated-gelu) */
};

/* One problem of a GROUPED launch (mx_gemm_desc.segs): mixed-resolution batches run every op of the step plan ONCE over the requests of all
 * resolutions present (the reference's reason to exist: modules/unet.py:104-185 cuts the latents of all resolutions into one patch batch).
 * The problems of a group share w, bias, N, K, the leading dimensions, flags and every scalar of the descriptor; each has its own rows,
 * per-batch structure and operand bases.  Output tiles never straddle two problems (a workgroup finds its problem from its tile index with
 * two compares and then runs the ordinary kernel body on it), nothing is padded in memory, and every output element sees exactly the
 * arithmetic of a separate launch with the same tile shape.  Pointer fields that the descriptor leaves NULL must be NULL here too. */
/* At most MX_MAX_SEGS resolutions share one launch sequence (the reference serves 512 / 768 / 1024 px).  A batch with MORE distinct resolutions is
 * not an error: the host mirrors (MxUNet.forward / SDXLDenoiser.denoising_step and the SD3 ones) then fall back to one launch sequence per
 * resolution on concurrent streams -- same results, the mixed-batch speed-up is lost. */
#define MX_MAX_SEGS 4
typedef struct mx_gemm_seg {
  const void* a;          /* this problem's A rows (conv: its first image) */
  const void* a2;
  void* c;
  const void* residual;
  void* vt;
  const float* rowbias;   /* row 0 = this problem's first sample */
  const float* gate;
  const float* ln_stats;
  float* stats_out;
  int M;                  /* rows (conv: B * Hout * Wout) */
  int rows_per_batch;
  int ldvt;
  int B, Hin, Win, Hout, Wout;                              /* conv geometry */
  int a_batch_rows, a_row_off, c_batch_rows, c_row_off;     /* joint-sequence remap */
} mx_gemm_seg;

typedef struct mx_gemm_desc {
  const void* a;        /* bf16 [M, K] row stride lda (elements); conv: NHWC input [B, Hin, Win, Cin] */
  const void* w;        /* bf16 [N, K] row-major (K = 9*Cin tap-major for conv) */
  void* c;              /* bf16 (or fp32) [M, ldc] */
  const float* bias;    /* fp32 [N] or NULL */
  const float* rowbias; /* fp32 [M / rows_per_batch, ldrb] or NULL (time-embedding add of resnet conv1) */
  const void* residual; /* bf16 [M, ldr] or NULL, added before the activation */
  void* vt;             /* MX_EPI_QKV: bf16 [batches][N/period][ldvt] */
  int M, N, K;
  int lda, ldc, ldr, ldrb;
  int rows_per_batch;
  int flags;
  int seg, period, ldvt; /* MX_EPI_QKV */
  /* conv geometry (ignored by mx_gemm) */
  int B, Hin, Win, Cin;  /* stored input */
  int Hout, Wout;        /* output grid; M = B*Hout*Wout */
  int stride;            /* 1 or 2 */
  int up;                /* 1: input is nearest-upsampled x2 on the fly (Hout = 2*Hin) */
  int corner_patch;      /* >0: sliced-mode halo-corner rule with this patch edge (output-grid pixels
                            for stride 1, input-grid pixels for stride 2); 0: plain zero padding */
  /* joint-sequence support (MMDiT): rows live in per-sample blocks of a longer sequence.
   * input  row of m = (m / rows_per_batch) * a_batch_rows + a_row_off + m % rows_per_batch   (a_batch_rows > 0)
   * output row of m = (m / rows_per_batch) * c_batch_rows + c_row_off + m % rows_per_batch   (c_batch_rows > 0);
   * it addresses C, the residual and, under MX_EPI_QKV, the key index of vt (then ldvt >= MX_VT_LD(c_batch_rows)). */
  int a_batch_rows, a_row_off, c_batch_rows, c_row_off;
  const float* gate;     /* fp32 [M / rows_per_batch, ldg] or NULL: v = gate * (acc + bias) before the residual add */
  int ldg;
  float out_scale;       /* != 0: v = (acc + bias) * out_scale, before row bias / gate / residual.  Under MX_EPI_QKV only the
                          * first segment of every group (q) is scaled -- for mx_attention_prescaled */
  /* MX_EPI_RMSNORM (with MX_EPI_QKV, period 3, N % 128 == 0): every 64-wide head of the q and k segments is RMS-normalised
   * before it is stored: x * rsqrt(mean(x^2) + rms_eps) * w, w = rms_wq / rms_wk (fp32 [64]); out_scale then multiplies the
   * normalised q.  Fuses diffusers' norm_q / norm_k (RMSNorm(64)) of the SD3 attention into the projection. */
  const float* rms_wq;
  const float* rms_wk;
  float rms_eps;
  /* conv3x3, patch-parallel (mx_unet_forward_pp): 1 = every image of `a` is stored with ONE extra row above and below
   * ([B, Hin + 2, Win, Cin], `a` pointing at the top halo row of image 0); taps that leave the image vertically read those
   * rows (the neighbour ranks' boundary rows, or zeros at the true image border) instead of zero padding. */
  int vhalo;
  /* mx_gemm: the A operand as a concatenation along K read in place: columns [0, k_split) from a (row stride lda), columns
   * [k_split, K) from a2 (row stride lda2); k_split % 64 == 0.  a2 == NULL: one source.  (The 1x1 shortcut conv of an up block
   * reads torch.cat([hidden_states, res_hidden_states]); not combined with the joint-sequence row remap.) */
  const void* a2;
  int lda2, k_split;
  /* LayerNorm folded into the GEMM that consumes it (the SDXL BasicTransformerBlock's norm1 / norm2 / norm3, modules/transformer.py:191,
   * 239, 266: each feeds exactly one linear).  With W' = W * gamma (per input feature, packed in `w`), colsum[n] = sum_k W'[n][k] (fp32, of
   * the bf16-rounded W') and bias' = bias + W beta (in `bias`):
   *     LayerNorm(x) W^T + bias  ==  rstd_m * (x W'^T - mean_m * colsum) + bias'
   * so the GEMM reads the UN-normalised x and its epilogue applies the row statistics: no LayerNorm pass, no rounding of the normalised
   * activations.  ln_stats != NULL selects it: fp32 pairs (sum x, sum x^2) per row and slab, ln_stats[(m * pitch + slab) * 2 + {0, 1}] with
   * pitch = MX_STATS_PITCH(ln_slabs), summed over ln_slabs slabs (written by the producing GEMM through stats_out, or by mx_row_stats
   * with one slab; entries past the last slab are never read); 16-byte aligned; the LN width is K.  Not combined with MX_EPI_RMSNORM or
   * the row remaps.
   * stats_out != NULL (plain bf16 output, no GEGLU / QKV / remap): the epilogue also writes, for every output row, the sum and the sum of
   * squares of the values it stores, one slab per wave column panel; mx_gemm_stats_slabs(d) tells how many slabs THIS launch writes
   * (0: the kernel chosen for d cannot, use mx_row_stats). */
  const float* ln_stats;
  const float* ln_colsum;
  int ln_slabs;
  float ln_eps;
  float* stats_out;
  /* grouped launch: n_segs in [1, MX_MAX_SEGS] problems (see mx_gemm_seg); 0 = the single problem described above.  With segs the
   * descriptor's own a / c / residual / vt / rowbias / gate / ln_stats / stats_out only say WHICH operands exist (non-NULL), M is ignored. */
  const mx_gemm_seg* segs;
  int n_segs;
  /* split-K (see mx_gemm_splitk): 0 = the library decides from the shape (a split is taken where its estimate beats the unsplit launch by 25 %),
   * 1 = never, 2..4 = that many slices wherever the launch is eligible at all (a 128-row tiling, >= 8 K tiles per slice, no split A operand) */
  int splitk;
  /* FINALISED row statistics (round 4): the form of the folded LayerNorm the persistent 256 x 256 kernel can afford.  A producer launched with
   * stats_out AND ln_final_out != NULL (256-row x 160 / 128 tiles only: mx_gemm_ln_final_supported) also leaves, per output row, the pair
   * (mean, rstd) over its N columns in ln_final_out[m * 2 + {0, 1}] (rstd with the producer's own ln_eps): the LAST workgroup of a 256-row
   * panel to finish -- a ticket in ln_final_cnt[panel], one unsigned per 256 rows, ZERO before the launch and zero again after it -- adds the
   * panel's slabs in slab order (bit-stable run to run).  A consumer launched with ln_final != NULL (and ln_colsum; ln_stats NULL; no grouped
   * launch, no row remap) reads those 8 bytes per row instead of the slabs: it runs on the 256 x 256 kernel where the plain launch would, and
   * the normalisation pass in front of it disappears.  16-byte aligned. */
  const float* ln_final;
  float* ln_final_out;
  unsigned* ln_final_cnt;
  /* GroupNorm statistics from the producing launch (round 4): gn_part_out != NULL makes a conv / GEMM on a 256-row tile whose epilogue is bias (+ per-sample
   * row bias) only ALSO leave, per 64 consecutive output rows and per channel, the sum and the sum of squares of its ACCUMULATORS -- the values it stores minus
   * the per-channel constants bias[n] + rowbias[sample][n] (fp32): gn_part_out[(m / 64 * N + n) * 2 + {0, 1}], M % 64 == 0 (and rows_per_batch % 64 == 0 with a row
   * bias).  mx_groupnorm_nhwc_from_partials, given the same bias / row bias, adds the constants back in closed form and needs no statistics pass over the tensor
   * (the reference's resnet: conv1 + time embedding -> norm2, resnet.py:414-429).  mx_gemm_gn_partials_supported(d,
   * conv) tells whether the launch can. */
  float* gn_part_out;
  /* mx_conv3x3 (round 5): > 0 = only the first cin_valid (<= 8) channels of every input pixel are non-zero -- the UNet's conv_in reads the 4 latent channels
   * zero-padded to Cin = 64.  The launch may then contract over 9 x 8 instead of 9 x Cin (conv_small_n.hip); 0 = every channel counts.  Never changes the result. */
  int cin_valid;
} mx_gemm_desc;

int mx_gemm(void* stream, const mx_gemm_desc* d);      /* C = A * W^T (+epilogue) */
/* 2 where mx_gemm cuts d along N (TAIL SPLIT: a launch of the persistent 256 x 256 kernel whose last round would be less than half full -- one
 * 1024 px request's GEGLU projection is 1.25 rounds -- runs its whole rounds there and the remaining column panels as ONE round of a smaller tile;
 * plain / gated epilogues, ungrouped, no statistics), else 1.  Shape-based and host-only. */
int mx_gemm_launches(const mx_gemm_desc* d);
int mx_gemm_stats_slabs(const mx_gemm_desc* d);        /* slabs d->stats_out receives from mx_gemm(d); 0 = not supported for this shape */
/* 1 when mx_gemm(d) with ln_stats would run on the persistent 256 x 256 kernel, where applying the statistics costs more than a separate
 * normalisation pass saves (10-20 us per launch in the hand-over between two tiles against an 11-us pass at M = 8192); the step plan then
 * normalises with mx_layernorm(gamma = NULL) and launches d without ln_stats.  The 256 / 128-row kernels hide the statistics behind their
 * first operand fetch: 0. */
int mx_gemm_ln_prefers_pass(const mx_gemm_desc* d);
/* 1 when mx_gemm(d) with stats_out can also write ln_final_out (the launch takes a 256-row tile of the register-exchange kernels, ungrouped) */
int mx_gemm_ln_final_supported(const mx_gemm_desc* d);
int mx_gemm_gn_partials_supported(const mx_gemm_desc* d, int conv);   /* 1 when mx_gemm / mx_conv3x3 (d) can write gn_part_out */
/* Which kernel family serves mx_gemm (conv = 0) / mx_conv3x3 (conv = 1) of d -- host only, the same chooser the launch uses:
 *   MX_FORM_TILE_GENERIC   the 128-row register-prefetch tile kernel (small or odd shapes)
 *   MX_FORM_TILE_256       a 256-row tile of the ping-pong kernels (256 x 160 / 128, one tile per CU)
 *   MX_FORM_TILE_128       a 128-row tile of the lock-step LDS-DMA kernel (small M; possibly split along K)
 *   MX_FORM_PERSISTENT_256 the persistent 256 x 256 kernel
 *   MX_FORM_SMALL_M        round 5: M <= 16 rows as a weight stream (bias, per-row residual, SiLU, bf16 / fp32 out; N % 16 == 0; above 64 MB of weights M (K + 8) <= 32 K elements)
 *   MX_FORM_CONV_SMALL_N   round 5: 3 x 3 conv with N <= 16 output channels (stride 1, bias only; weights + one staged chunk within 64 KB of LDS)
 *   MX_FORM_CONV_SMALL_CIN round 5: 3 x 3 conv whose descriptor names cin_valid <= 8 input channels (stride 1, bias only, N % 80 == 0) */
enum { MX_FORM_TILE_GENERIC = 0, MX_FORM_TILE_256 = 1, MX_FORM_TILE_128 = 2, MX_FORM_PERSISTENT_256 = 3, MX_FORM_SMALL_M = 4, MX_FORM_CONV_SMALL_N = 5, MX_FORM_CONV_SMALL_CIN = 6 };
int mx_gemm_form(const mx_gemm_desc* d, int conv);
#define MX_STATS_PITCH(slabs) (((slabs) + 3) & ~3)     /* slabs per row of a statistics buffer: [M][pitch][2] floats */
/* stats[m * 4 * 2 + {0, 1}] = (sum_c x[m][c], sum_c x[m][c]^2), x bf16 [M, C] with row stride ldx: the one-slab input of ln_stats
 * (buffer of M * MX_STATS_PITCH(1) * 2 floats) */
int mx_row_stats(void* stream, const void* x, int ldx, float* stats, int M, int C);
int mx_conv3x3(void* stream, const mx_gemm_desc* d);   /* implicit GEMM, pad 1 */
/* Slices the K range of d is dealt to (1 = no split).  Small launches -- fewer 128-row tiles than CUs and >= 16 K tiles: one request, light mixed
 * batches -- run SPLIT-K: every output tile is computed by up to 4 workgroups over disjoint K ranges; each leaves its fp32 partial tile in a
 * library-owned scratch (96 MB + counters per stream, allocated at a stream's first split launch -- also while that stream is being captured:
 * the allocation does not touch the capturing stream) and takes a ticket; the last arriver adds the partials IN SLICE ORDER (bit-stable run to
 * run) and runs the ordinary epilogue.  The value returned is what mx_gemm / mx_conv3x3 (d) DOES: the decision depends on the descriptor alone, so a
 * shape adds its products in the same order eagerly, under capture and on replay; a scratch that cannot be allocated makes the launch fail (set
 * d->splitk = 1 to run unsplit).  Shape-based and host-only. */
int mx_gemm_splitk(const mx_gemm_desc* d, int conv);
/* frees the split-K scratch of `stream` (all != 0: of every stream) after waiting for that stream; library unload frees what is left */
void mx_gemm_release_scratch(void* stream, int all);

/* ---- the ATTENTION TAIL of a BasicTransformerBlock as ONE launch (round 5; attn_tail.hip).  The four dependent launches
 *   y = attn1.to_out(ao) + y (+ row statistics)  ->  q2 = attn2.to_q(norm2(y))  ->  ao2 = softmax(q2 K^T) V over the text keys  ->  y = attn2.to_out(ao2) + y (+ statistics)
 * (transformer.py:204-262, attention.py:59-110) become work items of one persistent launch: a 256-row panel's items of stage s need stage s - 1 of the same
 * panel only, and the workgroups that take them hand the panel's rows over inside the launch (write-through stores, a ticket per panel and stage).  The
 * three descriptors are EXACTLY those of the separate launches (mx_gemm(out1); mx_gemm(to_q); mx_attention_cross_prescaled; mx_gemm(out2)) and the results
 * equal theirs bit for bit: the same tiles of the same kernels in the same order of summation.  mx_attn_tail_supported tells whether a descriptor can be
 * served (plain C x C linears on 256 x 160 tiles over M = B * L rows, L % 256 == 0, C = heads * 64, ctx_len <= 96; to_q reads out1's output and slab
 * statistics; ao / y / q2 / ao2 four different buffers, the two statistics buffers different).  mx_attn_tail_preferred: 1 when the step plans should
 * take it where it is supported -- MX_ATTN_TAIL=1 in the environment; the DEFAULT IS 0: measured on MI355X the chained launch ties with the four launches
 * in isolation and is 2 % slower inside the SDXL step, so it ships as an option, not as the plan's path.
 * sync: mx_attn_tail_sync_bytes(M) bytes of device memory, ZERO before the first launch; every launch leaves them zero (except the error word read by
 * mx_attn_tail_status: != 0 when a wait inside a launch gave up after ~2^22 polls instead of hanging the device -- that launch's output is invalid). ---- */
typedef struct mx_attn_tail_desc {
  mx_gemm_desc out1, to_q, out2;
  const void* k; int ldk;                    /* the layer's cross-attention keys: bf16 [B * ctx_len, ldk] */
  const void* vt; int ldvt; int64_t vt_batch_stride;   /* V^T in MX_VT_POS order, per sample */
  int B, heads, L, ctx_len;
  unsigned* sync;
} mx_attn_tail_desc;
size_t mx_attn_tail_sync_bytes(int M);
int mx_attn_tail_supported(const mx_attn_tail_desc* d);
int mx_attn_tail_preferred(void);
int mx_attn_tail(void* stream, const mx_attn_tail_desc* d);
int mx_attn_tail_status(void* stream, const unsigned* sync, unsigned* word);

/* V^T key order.  The attention kernel feeds its softmax accumulator straight back to the matrix core as the
 * P operand, and that register layout interleaves keys in blocks of four (lane half h owns keys 4h..4h+3 and
 * 8+4h..8+4h+3 of every 16).  V^T is therefore stored with bits 2 and 3 of the key index swapped, so that each
 * lane's eight V^T values are one aligned 16-byte word.  MX_VT_POS is its own inverse; rows are MX_VT_LD(Lk) long.
 * mx_gemm's MX_EPI_QKV epilogue writes this order; pad positions need not be initialised. */
#define MX_VT_POS(key) (((key) & ~12) | (((key) & 4) << 1) | (((key) & 8) >> 1))
#define MX_VT_LD(Lk) (((Lk) + 15) / 16 * 16)

/* softmax(Q K^T * scale) V per (batch, head); head_dim 64.
 * q: bf16 rows (b*Lq + i), head h at columns [64h, 64h+64), row stride ldq; k likewise (Lk, ldk);
 * vt: bf16, V transposed: element (b, h, d, key) at vt[b*vt_batch_stride + (h*64 + d)*ldvt + MX_VT_POS(key)],
 *     ldvt >= MX_VT_LD(Lk) and a multiple of 8;
 * o: bf16 [B*Lq, ldo]. */
int mx_attention(void* stream, const void* q, int ldq, const void* k, int ldk, const void* vt, int ldvt,
                 int64_t vt_batch_stride, void* o, int ldo, int B, int H, int Lq, int Lk, float scale);
/* Same, for q that the producer already multiplied by MX_ATTN_QSCALE(scale) = scale * log2(e) (mx_gemm's out_scale, or
 * mx_rmsnorm_heads' q_scale, do it in fp32 before their single rounding): the kernel then subtracts the softmax reference
 * inside the matrix core and runs ~35 % fewer vector instructions per key tile (attention.hip).  What the step plans use. */
#define MX_ATTN_QSCALE(scale) ((scale) * 1.4426950408889634f)
int mx_attention_prescaled(void* stream, const void* q, int ldq, const void* k, int ldk, const void* vt, int ldvt,
                           int64_t vt_batch_stride, void* o, int ldo, int B, int H, int Lq, int Lk);
/* the same through the SHORT-KEY kernel whatever Lq (Lk <= 96: every wave keeps the head's K / V^T in registers): what stage 2 of mx_attn_tail runs;
 * mx_attention_prescaled itself takes this kernel from Lq >= 2048 and the general one below (the two differ in bf16 rounding of P) */
int mx_attention_cross_prescaled(void* stream, const void* q, int ldq, const void* k, int ldk, const void* vt, int ldvt,
                                 int64_t vt_batch_stride, void* o, int ldo, int B, int H, int Lq, int Lk);
/* Grouped form: the attention problems of all resolutions present in a mixed batch in ONE launch (see mx_gemm_seg).  The problems share the row
 * strides and the head count; each has its own batch, sequence lengths and operand bases.  One kernel serves the whole launch (chosen by the
 * longest query sequence); the arithmetic per problem is that of mx_attention_prescaled with the same kernel. */
typedef struct mx_attn_problem {
  const void* q; const void* k; const void* vt; void* o;
  int64_t vt_batch_stride;
  int B, Lq, Lk, ldvt;
} mx_attn_problem;
int mx_attention_prescaled_grouped(void* stream, const mx_attn_problem* probs, int n, int ldq, int ldk, int ldo, int H);
/* softmax(q k^T + bias[h]) v: q and the fp32 bias [H][Lq][ldb] both already multiplied by log2(e) (T5: no 1/sqrt(d) scaling, bucketed relative
 * position bias shared by all layers); ldb % 4 == 0, ldb >= Lk rounded up to 64 */
int mx_attention_prescaled_bias(void* stream, const void* q, int ldq, const void* k, int ldk, const void* vt, int ldvt,
                                int64_t vt_batch_stride, void* o, int ldo, int B, int H, int Lq, int Lk, const float* bias, int ldb);
/* the same with the causal mask (key j counts for query i only when j <= i), Lq == Lk == L: the CLIP text encoders */
int mx_attention_prescaled_causal(void* stream, const void* q, int ldq, const void* k, int ldk, const void* vt, int ldvt,
                                  int64_t vt_batch_stride, void* o, int ldo, int B, int H, int L);

/* y = LayerNorm(x) * gamma + beta over the last dim C; x,y bf16 [M, C]; gamma/beta fp32 [C], or both NULL: plain (x - mean) * rstd, the
 * input of a linear whose weights carry the affine (weights folded for mx_gemm_desc.ln_stats work unchanged on it, without ln_stats) */
int mx_layernorm(void* stream, const void* x, void* y, const float* gamma, const float* beta,
                 int M, int C, float eps);

/* y = x * rsqrt(mean(x^2) + eps) * w over the last dim C (T5LayerNorm: no mean subtraction, no bias); x,y bf16 [M, C], w fp32 [C] */
int mx_rmsnorm(void* stream, const void* x, void* y, const float* w, int M, int C, float eps);

/* AdaLN modulate: y = LayerNorm(x) * (1 + scale[b]) + shift[b] (no affine), optional second output y2 with
 * (scale2, shift2) sharing the normalisation; x,y bf16 [M, C]; scale/shift fp32 rows with stride ldmod, b = row / rows_per_batch */
int mx_layernorm_mod(void* stream, const void* x, void* y, void* y2, const float* scale, const float* shift,
                     const float* scale2, const float* shift2, int ldmod, int M, int C, int rows_per_batch, float eps);
/* the same over a mixed batch: n groups of rows one after the other, group g = batches[g] samples of rows_per_batch[g] rows each; scale / shift
 * rows are per sample in group order (one launch for the image stream of all resolutions, see mx_gemm_seg) */
int mx_layernorm_mod_grouped(void* stream, const void* x, void* y, void* y2, const float* scale, const float* shift, const float* scale2,
                             const float* shift2, int ldmod, int C, float eps, const int* batches, const int* rows_per_batch, int n);
/* in-place RMSNorm over every 64-wide head of rows (b*batch_rows + row_off + i), i < rows_per_batch, of a bf16 [*, ld] matrix;
 * heads [0, heads_q) use weight wq[64], heads [heads_q, heads_total) use wk[64] (fp32) */
int mx_rmsnorm_heads(void* stream, void* x, int ld, int nbatch, int rows_per_batch, int batch_rows, int row_off,
                     int heads_total, int heads_q, const float* wq, const float* wk, float eps, float q_scale);
/* q_scale multiplies the q heads after normalisation (1 = none; MX_ATTN_QSCALE(1/8) for mx_attention_prescaled) */

/* NHWC GroupNorm (+ optional SiLU): x,y bf16 [B, H, W, C]; gamma/beta fp32 [C].
 * patch > 0 selects the reference's sliced statistics (average over patch x patch tiles of
 * per-tile mean and biased variance, norm_silu_concat.cu:361-386); 0 = exact GroupNorm.
 * workspace >= mx_groupnorm_nhwc_workspace_bytes(B, H, W, C). */
size_t mx_groupnorm_nhwc_workspace_bytes(int B, int H, int W, int C);
int mx_groupnorm_nhwc(void* stream, const void* x, void* y, const float* gamma, const float* beta,
                      int B, int H, int W, int C, int groups, float eps, int silu, int patch,
                      void* workspace);

/* The same over a channel concatenation read in place: channels [0, C1) from x ([B, H, W, C1]), channels [C1, C) from x2
 * ([B, H, W, C - C1]); y is [B, H, W, C].  Serves the up blocks' torch.cat([hidden_states, res_hidden_states], dim=1) feeding
 * norm1 (unet_2d_blocks.py via unet.py:458-462) without materialising the concatenation.  x2 == NULL: mx_groupnorm_nhwc. */
int mx_groupnorm_nhwc_cat(void* stream, const void* x, int C1, const void* x2, void* y, const float* gamma, const float* beta,
                          int B, int H, int W, int C, int groups, float eps, int silu, int patch, void* workspace);
/* GroupNorm (+SiLU) of x [B, H, W, C] from per-chunk partial sums a producing launch left (mx_gemm_desc.gn_part_out): part[(token / chunk * C + c) * 2 + {0, 1}] over
 * `chunk` consecutive tokens (chunk divides H * W), exact statistics only (no sliced form).  add_bias [C] (may be NULL) / add_rowbias [B][ldrb] (may be NULL): the
 * partial sums are those of x[.., c] - (add_bias[c] + add_rowbias[image][c]) -- the producer's bias and per-sample row bias.  Fold + apply: the statistics read
 * pass does not run.  workspace: mx_groupnorm_nhwc_workspace_bytes(B, H, W, C). */
int mx_groupnorm_nhwc_from_partials(void* stream, const void* x, void* y, const float* gamma, const float* beta, int B, int H, int W, int C, int groups, float eps,
                                    int silu, const float* part, int chunk, const float* add_bias, const float* add_rowbias, int ldrb, void* workspace);

/* Grouped form (see mx_gemm_seg): the GroupNorms of all resolutions present in a mixed batch as ONE stats / fold / apply launch each.  x2 as in
 * mx_groupnorm_nhwc_cat (all problems or none); workspace >= mx_groupnorm_nhwc_grouped_workspace_bytes(probs, n, C). */
typedef struct mx_gn_problem { const void* x; const void* x2; void* y; int B, H, W; } mx_gn_problem;
size_t mx_groupnorm_nhwc_grouped_workspace_bytes(const mx_gn_problem* probs, int n, int C);
int mx_groupnorm_nhwc_grouped(void* stream, const mx_gn_problem* probs, int n, int C1, const float* gamma, const float* beta, int C, int groups,
                              float eps, int silu, int patch, void* workspace);

/* ------------------------------------------------------------------------------------------
 * Outer boundary: the SDXL UNet in the model slot.
 * ------------------------------------------------------------------------------------------ */
typedef struct mx_unet_config {
  int in_channels, out_channels;
  int n_levels;                    /* <= 4 */
  int block_out_channels[4];
  int layers_per_block;
  int down_has_attn[4];
  int transformer_layers[4];
  int num_heads[4];                /* head_dim must be 64 */
  int cross_attention_dim;
  int addition_time_embed_dim;
  int projection_class_embeddings_input_dim;
  int norm_num_groups;
  float norm_eps, transformer_norm_eps, layer_norm_eps;
} mx_unet_config;

typedef struct mx_weight_entry {
  const char* name;   /* packed-tensor name, see sduss_amd/weights.py */
  uint64_t offset;    /* byte offset into the blob */
  uint64_t bytes;
} mx_weight_entry;

typedef struct mx_unet mx_unet;

mx_unet* mx_unet_create(const mx_unet_config* cfg);
void mx_unet_destroy(mx_unet* u);
/* the blob (device memory, packed by sduss_amd/weights.py) stays owned by the caller and must
 * outlive the handle; the table is copied */
int mx_unet_set_weights(mx_unet* u, const void* blob, uint64_t blob_bytes,
                        const mx_weight_entry* table, int n_entries);
/* PER-COMPOSITION store of the cross-attention K / V^T (round 5).  The text embeddings of a request do not change over its steps, yet the
 * reference's step re-concatenates them and every forward projects them again for all 70 transformer layers (pipeline_stable_diffusion_xl_esymred.py:
 * 287-339; attention.py:59-110 to_kv).  A caller that knows the batch composition names it before each forward: key != 0 announces that the NEXT
 * forward of this handle (ONE call: the key is consumed by it) receives encoder_hidden_states with the same CONTENT and row order as every earlier
 * forward announced with this key did (same batch, same ctx_len).  The first such forward projects into library-owned device buffers, later ones
 * read them -- the same GEMM's output, bit for bit (up to 4 compositions are kept, least recently used first out; ~26 MB per sample row at SDXL-base
 * width).  A forward that was not announced projects as before.  Applies to mx_unet_forward / _forward_mixed (not to the patch-parallel or block-cache entry points, nor under MX_GRAPH=1);
 * mx_unet_set_weights drops every stored projection.  Entries are handed between streams through events: forwards of one handle may be issued on
 * several streams, from one host thread. */
int mx_unet_set_context_key(mx_unet* u, uint64_t key);
int mx_unet_context_stats(const mx_unet* u, long* hits, long* misses);   /* forwards served from / written to the store since creation */
size_t mx_unet_workspace_bytes(const mx_unet* u, int batch, int H, int W, int ctx_len);
/* host-only walk of the step plan that resolves every packed tensor by name and size (no launches, no GPU needed) */
int mx_unet_validate(const mx_unet* u, int batch, int H, int W, int ctx_len);
/* One UNet forward over `batch` whole latents of one resolution (CFG rows included by the caller).
 *   latents  [batch, in_channels, H, W]  of `io_dtype` (NCHW, as the reference passes them)
 *   timesteps fp32 [batch]; ehs bf16 [batch, ctx_len, cross_attention_dim];
 *   text_embeds bf16 [batch, text_dim]; time_ids fp32 [batch, 6]
 *   out      [batch, out_channels, H, W] of `io_dtype`
 *   gn_patch 0 = is_sliced False (exact GroupNorm, zero-padded convs);
 *            p>0 = is_sliced True with latent patch edge p: patch-averaged GroupNorm statistics and the
 *            halo-corner rule, i.e. the same real-number arithmetic as the reference's sliced path, evaluated on whole
 *            images (not bit-for-bit: bf16 storage and fp32 statistics here, fp16 storage and fp16-rounded statistics
 *            there, norm_silu_concat.cpp:84-85; tolerance stated in tests/test_unet_gpu.py). */
int mx_unet_forward(mx_unet* u, void* stream, const void* latents, int io_dtype, const float* timesteps,
                    const void* ehs, const void* text_embeds, const float* time_ids, void* out,
                    int batch, int H, int W, int ctx_len, int gn_patch, void* workspace, size_t workspace_bytes);
/* Mixed-resolution batch (BASELINE configs[4]): the requests of EVERY resolution present run through ONE launch sequence --
 * what the reference obtains by cutting the latents of all resolutions into one batch of 256-px patches (modules/unet.py:104-185, split_sample)
 * and regrouping per latent before attention (attention.py:152-203).  Here nothing is cut: a group = the samples of one resolution, the
 * activations of a level are the groups' images one after the other, per-token ops are single launches over all rows and the ops with per-image
 * structure are grouped launches (mx_gemm_seg, mx_attention_prescaled_grouped, mx_groupnorm_nhwc_grouped).  The conditioning rows (timesteps, ehs,
 * text_embeds, time_ids) are those of all groups concatenated in group order -- the reference's row order, ascending resolution
 * (pipeline_stable_diffusion_xl_esymred.py:275-276).  Per request the arithmetic is that of mx_unet_forward on its group alone up to the tile
 * shapes a larger launch selects and the grouping of the fp32 partial sums of the GroupNorm / LayerNorm statistics. */
typedef struct mx_unet_group { const void* latents; void* out; int batch, H, W; } mx_unet_group;   /* [batch, C, H, W] of io_dtype each */
size_t mx_unet_workspace_bytes_mixed(const mx_unet* u, const mx_unet_group* groups, int n_groups, int ctx_len);
int mx_unet_forward_mixed(mx_unet* u, void* stream, const mx_unet_group* groups, int n_groups, int io_dtype, const float* timesteps,
                          const void* ehs, const void* text_embeds, const float* time_ids, int ctx_len, int gn_patch, void* workspace,
                          size_t workspace_bytes);
/* the same with a stage dump (tests): the NHWC bf16 activation of all groups after `stage`, [sum of the groups' pixels, C] */
int mx_unet_forward_mixed_trace(mx_unet* u, void* stream, const mx_unet_group* groups, int n_groups, int io_dtype, const float* timesteps,
                                const void* ehs, const void* text_embeds, const float* time_ids, int ctx_len, int gn_patch, void* workspace,
                                size_t workspace_bytes, const char* stage, void* stage_out, size_t stage_out_bytes);
/* debugging / parity: copy of the NHWC bf16 activation after the named stage of the LAST forward
 * is not kept; instead a forward can be asked to stop after `stage` and dump it (tests only). */
int mx_unet_forward_trace(mx_unet* u, void* stream, const void* latents, int io_dtype, const float* timesteps,
                          const void* ehs, const void* text_embeds, const float* time_ids, void* out,
                          int batch, int H, int W, int ctx_len, int gn_patch, void* workspace,
                          size_t workspace_bytes, const char* stage, void* stage_out, size_t stage_out_bytes);

/* ------------------------------------------------------------------------------------------
 * Patch parallelism (BASELINE.json configs[3]): ONE request's latent rows split over `world` GPUs, the bundled distrifuser
 * baseline's DistriUNetPP (distrifuser/distrifuser/distrifuser/models/distri_sdxl_unet_pp.py:15-216) in its synchronous mode
 * (every step exchanges fresh tensors; utils.py:119-214 PatchParallelismCommManager, modules/pp/{conv2d,groupnorm,attn}.py):
 *   - 3x3 convs read one boundary row of each neighbour rank (all-gather of [2][batch][W * C] per conv),
 *   - GroupNorm folds the ranks' per-(image, group) {sum, sum of squares} (all-gather of 16-byte records),
 *   - self-attention keeps its local queries and gathers the other ranks' K rows and V^T columns,
 *   - everything per-token (LayerNorm, projections, GEGLU, cross-attention, 1x1 convs) is local.
 * Rank r owns latent rows [r * H_local, (r + 1) * H_local); H_local must be divisible by 2^(levels - 1) and H_local * W / 4^level
 * a multiple of 64 at every attention level.  The collective is supplied by the caller: all_gather(ctx, stream, send, recv,
 * bytes_per_rank) must place rank k's `send` at recv + k * bytes_per_rank, ordered after prior work on `stream` and before
 * later work on it, and return 0.  sduss_amd/patch_parallel.py binds it to torch.distributed (backend "nccl" == RCCL over
 * xGMI on MI355X; gloo through host memory in the tests).  send / recv always lie inside `workspace`.
 * Results equal mx_unet_forward on the whole latent (gn_patch 0) up to the summation order of the GroupNorm statistics.
 * ------------------------------------------------------------------------------------------ */
typedef int (*mx_allgather_fn)(void* ctx, void* stream, const void* send, void* recv, size_t bytes_per_rank);
typedef struct mx_pp_comm { int rank, world; mx_allgather_fn all_gather; void* ctx; } mx_pp_comm;
size_t mx_unet_workspace_bytes_pp(const mx_unet* u, int batch, int H_local, int W, int ctx_len, int world);
/* latents_local / out_local: [batch, C, H_local, W] of io_dtype (this rank's rows, NCHW) */
int mx_unet_forward_pp(mx_unet* u, void* stream, const void* latents_local, int io_dtype, const float* timesteps, const void* ehs,
                       const void* text_embeds, const float* time_ids, void* out_local, int batch, int H_local, int W, int ctx_len,
                       const mx_pp_comm* comm, void* workspace, size_t workspace_bytes);
/* Stale-asynchronous steps: distrifuser's default mode after its warm-up (synchronous while counter <= warmup_steps: warmup_steps + 1 = 5
 * synchronous steps at the default of 4; utils.py:30-32, 180-214;
 * modules/pp/conv2d.py:97-117, attn.py:136-146, groupnorm.py:46-66).  Every exchange k of a forward owns region k of a state buffer the caller
 * keeps across steps.  A WARMUP step is a synchronous step that also leaves what it gathered in the state.  A STALE step reads, for every
 * exchange, the OTHER ranks' slots as they were sent one step earlier and its own slot fresh, and hands its fresh slot to
 * all_gather_async(ctx, stream, region, bytes_per_rank): an in-place all-gather over the `world` slots of `region` (this rank's slot is already
 * filled) that must start after the work queued on `stream` so far and may complete at any time before the NEXT forward's first launch -- the
 * caller orders that (sduss_amd/patch_parallel.py: a side stream and an event per step).  Approximate by construction: with unchanged inputs a
 * stale step reproduces the synchronous result bit for bit; otherwise it lags one step in what it sees of the other ranks' rows.
 * corrected_gn: 0 = "stale_gn" (fresh own sums beside stale remote ones), 1 = "corrected_async_gn", distrifuser's default (stale whole-image
 * moments + this rank's change, local variance where that turns negative).  distrifuser's unbiased-variance factor is not applied in either
 * mode: the synchronous arithmetic here is nn.GroupNorm's. */
#define MX_PP_SYNC 0
#define MX_PP_WARMUP 1
#define MX_PP_STALE 2
typedef int (*mx_allgather_inplace_fn)(void* ctx, void* stream, void* region, size_t bytes_per_rank);
typedef struct mx_pp_stale { void* state; size_t state_bytes; int mode; int corrected_gn; mx_allgather_inplace_fn all_gather_async; } mx_pp_stale;
size_t mx_unet_pp_state_bytes(const mx_unet* u, int batch, int H_local, int W, int ctx_len, int world);
int mx_unet_forward_pp_stale(mx_unet* u, void* stream, const void* latents_local, int io_dtype, const float* timesteps, const void* ehs,
                             const void* text_embeds, const float* time_ids, void* out_local, int batch, int H_local, int W, int ctx_len,
                             const mx_pp_comm* comm, const mx_pp_stale* stale, void* workspace, size_t workspace_bytes);
/* Host-only walk of the patch-parallel plan (no launches, no GPU): calls comm->all_gather once per exchange of a forward, in
 * order, with send / recv = (void*)(0x1000 + byte offset of the region inside the workspace). */
int mx_unet_pp_comm_plan(const mx_unet* u, int batch, int H_local, int W, int ctx_len, const mx_pp_comm* comm);
/* mx_attention_prescaled for K / V^T gathered rank-major: keys [c * key_chunk, (c + 1) * key_chunk) of batch b have their K rows at
 * k + c * k_chunk_stride + b * k_batch_stride (row stride ldk) and their V^T words at vt + c * vt_chunk_stride + b * vt_batch_stride
 * + (h * 64 + d) * ldvt + MX_VT_POS(key - c * key_chunk); key_chunk % 64 == 0, Lk % key_chunk == 0 (strides in elements). */
int mx_attention_prescaled_chunked(void* stream, const void* q, int ldq, const void* k, int ldk, const void* vt, int ldvt,
                                   int64_t vt_batch_stride, void* o, int ldo, int B, int H, int Lq, int Lk, int key_chunk,
                                   int64_t k_batch_stride, int64_t k_chunk_stride, int64_t vt_chunk_stride);

/* ------------------------------------------------------------------------------------------
 * Outer boundary: the SD3.5 MMDiT in the ``transformer`` slot
 * (PatchSD3Transformer2DModel.forward, sduss/model_executor/modules/SD3Transformer.py:60-262, invoked at
 *  pipelines/stable_diffusion_3/pipeline_stable_diffusion_3_esymred.py:312-322).
 * ------------------------------------------------------------------------------------------ */
typedef struct mx_mmdit_config {
  int patch_size, in_channels, out_channels;
  int num_layers;               /* <= 64 */
  int num_attention_heads;      /* head_dim is 64 */
  int joint_attention_dim, pooled_projection_dim, pos_embed_max_size;
  int dual_attention[64];       /* 1 where the block has the second, image-only attention (SD3.5 dual_attention_layers) */
  float norm_eps;
} mx_mmdit_config;

typedef struct mx_mmdit mx_mmdit;
mx_mmdit* mx_mmdit_create(const mx_mmdit_config* cfg);
void mx_mmdit_destroy(mx_mmdit* u);
int mx_mmdit_set_weights(mx_mmdit* u, const void* blob, uint64_t blob_bytes, const mx_weight_entry* table, int n_entries);
size_t mx_mmdit_workspace_bytes(const mx_mmdit* u, int batch, int H, int W, int ctx_len);
int mx_mmdit_validate(const mx_mmdit* u, int batch, int H, int W, int ctx_len);
/* latents [batch, in_channels, H, W] of io_dtype (NCHW); timesteps fp32 [batch]; ehs bf16 [batch, ctx_len, joint_attention_dim];
 * pooled bf16 [batch, pooled_projection_dim]; out [batch, out_channels, H, W] of io_dtype */
int mx_mmdit_forward(mx_mmdit* u, void* stream, const void* latents, int io_dtype, const float* timesteps, const void* ehs,
                     const void* pooled, void* out, int batch, int H, int W, int ctx_len, void* workspace, size_t workspace_bytes);
int mx_mmdit_forward_trace(mx_mmdit* u, void* stream, const void* latents, int io_dtype, const float* timesteps, const void* ehs,
                           const void* pooled, void* out, int batch, int H, int W, int ctx_len, void* workspace,
                           size_t workspace_bytes, const char* stage, void* stage_out, size_t stage_out_bytes);

/* Mixed-resolution batch for the SD3 / SD3.5 transformer, as mx_unet_forward_mixed: group g = the samples of one resolution ([batch, C, H, W]
 * latents in, the same shape out); timesteps / encoder_hidden_states / pooled_projections are the rows of all groups in group order.  The
 * reference re-chunks the tokens of all resolutions into one batch (modules/utils.py:86-122) and regroups them per latent before attention. */
size_t mx_mmdit_workspace_bytes_mixed(const mx_mmdit* u, const mx_unet_group* groups, int n_groups, int ctx_len);
int mx_mmdit_forward_mixed(mx_mmdit* u, void* stream, const mx_unet_group* groups, int n_groups, int io_dtype, const float* timesteps,
                           const void* encoder_hidden_states, const void* pooled_projections, int ctx_len, void* workspace, size_t workspace_bytes);

/* Patch parallelism for the SD3 / SD3.5 transformer (distrifuser models/distri_sd3_transformer_pp.py:87-97: the positional embedding is taken
 * for the whole grid, then the image tokens are sliced by rank; modules/pp/attn.py:202-277: the joint attention keeps its local queries -- this
 * rank's image tokens and the text tokens -- and gathers the other ranks' image K / V; the text stream is computed by every rank).
 * Rank r owns latent rows [r * H_local, (r + 1) * H_local); (H_local / patch_size) * (W / patch_size) must be a multiple of 16.
 * Per joint block: one all-gather of the image tokens' K rows [batch][L_local][d], one of their V^T columns [batch][d][L_local] (attn2 of the
 * dual blocks: two more).  stale = NULL: every step
 * synchronous; otherwise as mx_unet_forward_pp_stale.  Equal to mx_mmdit_forward on the whole latent up to the GEMM tile selection. */
size_t mx_mmdit_workspace_bytes_pp(const mx_mmdit* u, int batch, int H_local, int W, int ctx_len, int world);
size_t mx_mmdit_pp_state_bytes(const mx_mmdit* u, int batch, int H_local, int W, int ctx_len, int world);
int mx_mmdit_forward_pp(mx_mmdit* u, void* stream, const void* latents_local, int io_dtype, const float* timesteps, const void* encoder_hidden_states,
                        const void* pooled_projections, void* out_local, int batch, int H_local, int W, int ctx_len, const mx_pp_comm* comm,
                        const mx_pp_stale* stale, void* workspace, size_t workspace_bytes);
int mx_mmdit_pp_comm_plan(const mx_mmdit* u, int batch, int H_local, int W, int ctx_len, const mx_pp_comm* comm);

/* ------------------------------------------------------------------------------------------
 * The step after the loop: the SDXL VAE decoder (AutoencoderKL.decode as post_inference calls it,
 * pipelines/stable_diffusion_xl/pipeline_stable_diffusion_xl_esymred.py:406-463).  SURVEY.md section 8f rank 2.
 * latents [batch, latent_channels, H, W] of io_dtype (NCHW, NOT yet divided by the scaling factor: 1 / scaling_factor is folded
 * into the packed post_quant_conv weight) -> images [batch, out_channels, 8H, 8W] of out_dtype in [-1, 1] (before the pipeline's
 * image_processor.postprocess).
 * ------------------------------------------------------------------------------------------ */
typedef struct mx_vae_config {
  int latent_channels, out_channels;
  int n_levels;                   /* <= 4 */
  int block_out_channels[4];      /* encoder order, e.g. 128, 256, 512, 512: the decoder walks them backwards */
  int layers_per_block;           /* the decoder has layers_per_block + 1 resnets per up block */
  int norm_num_groups;
  float norm_eps;
} mx_vae_config;
typedef struct mx_vae mx_vae;
mx_vae* mx_vae_create(const mx_vae_config* cfg);
void mx_vae_destroy(mx_vae* v);
int mx_vae_set_weights(mx_vae* v, const void* blob, uint64_t blob_bytes, const mx_weight_entry* table, int n_entries);
size_t mx_vae_workspace_bytes(const mx_vae* v, int batch, int H, int W);
int mx_vae_validate(const mx_vae* v, int batch, int H, int W);
int mx_vae_decode(mx_vae* v, void* stream, const void* latents, int io_dtype, void* out, int out_dtype, int batch, int H, int W,
                  void* workspace, size_t workspace_bytes);

/* ------------------------------------------------------------------------------------------
 * CLIP text encoder (csrc/clip_text.cpp): what diffusers' encode_prompt runs per prompt before the denoising loop
 * (prepare_inference, pipeline_stable_diffusion_xl_esymred.py:118-140): transformers CLIPTextModel (SDXL text_encoder: ViT-L, quick_gelu,
 * no projection) and CLIPTextModelWithProjection (text_encoder_2: OpenCLIP bigG, gelu, text_projection).  Packed weights keep the
 * transformers names except q_proj / k_proj / v_proj, fused into "<layer>.self_attn.qkv_proj.{weight [3H, H], bias [3H]}".
 *   ids: int32 [batch, max_position_embeddings] token ids (the tokenizer stays on the host);
 *   hidden_out: bf16 [batch, L, H] = transformers' hidden_states[hidden_layer] (-2 for SDXL: the output of the last-but-one layer; -1 =
 *               the last layer's output, before final_layer_norm), or NULL;
 *   pooled_out: fp32 [batch, projection_dim] = text_projection(final_layer_norm(last)[eos position]), or NULL.
 * ------------------------------------------------------------------------------------------ */
typedef struct mx_clip_config {
  int vocab_size, hidden_size, intermediate_size, num_hidden_layers, num_attention_heads, max_position_embeddings;
  int hidden_act;                 /* 0 = quick_gelu, 1 = gelu (erf) */
  int projection_dim;             /* 0: no text_projection */
  int eos_token_id;               /* 2 = the legacy configs: pooled row = position of the largest id */
  int hidden_layer;               /* which hidden state hidden_out receives, a negative index as in Python: -2 (SDXL, clip_skip None), -1 ... */
  float layer_norm_eps;
} mx_clip_config;
typedef struct mx_clip mx_clip;
mx_clip* mx_clip_create(const mx_clip_config* cfg);
void mx_clip_destroy(mx_clip* c);
int mx_clip_set_weights(mx_clip* c, const void* blob, uint64_t blob_bytes, const mx_weight_entry* table, int n_entries);
size_t mx_clip_workspace_bytes(const mx_clip* c, int batch);
int mx_clip_validate(const mx_clip* c, int batch);
int mx_clip_encode(mx_clip* c, void* stream, const int32_t* ids, void* hidden_out, void* pooled_out, int batch, void* workspace,
                   size_t workspace_bytes);

/* ------------------------------------------------------------------------------------------
 * T5 v1.1 encoder (csrc/t5_text.cpp): SD3's text_encoder_3 (transformers T5EncoderModel, XXL: 24 blocks x 4096, 64 heads of 64, d_ff 10240,
 * gated gelu_new), run by encode_prompt on 256 token ids per prompt without an attention mask; out = last_hidden_state bf16 [batch, L, d_model].
 * Packed weights keep the transformers names except: q / k / v fused into "<block>.layer.0.SelfAttention.qkv.weight" [3 * 64 heads, d_model],
 * wi_1 | wi_0 interleaved for the GEGLU epilogue into "<block>.layer.1.DenseReluDense.wi.weight" [2 d_ff, d_model], and the relative position
 * bias of block 0 expanded for the sequence length into "encoder.position_bias.<L>" fp32 [heads, L, ceil64(L)], times log2(e)
 * (sduss_amd/t5.py::pack_t5).
 * ------------------------------------------------------------------------------------------ */
typedef struct mx_t5_config {
  int vocab_size, d_model, d_ff, num_layers, num_heads;    /* d_kv = 64 */
  float layer_norm_epsilon;
} mx_t5_config;
typedef struct mx_t5 mx_t5;
mx_t5* mx_t5_create(const mx_t5_config* cfg);
void mx_t5_destroy(mx_t5* t);
int mx_t5_set_weights(mx_t5* t, const void* blob, uint64_t blob_bytes, const mx_weight_entry* table, int n_entries);
size_t mx_t5_workspace_bytes(const mx_t5* t, int batch, int L);
int mx_t5_validate(const mx_t5* t, int batch, int L);
int mx_t5_encode(mx_t5* t, void* stream, const int32_t* ids, void* out, int batch, int L, void* workspace, size_t workspace_bytes);

/* ------------------------------------------------------------------------------------------
 * Block-skip cache ("Mix-Cache"): the reference's CacheManager with ESYMRED_USE_CACHE=TRUE (modules/cache_manager.py:101-191, called at the
 * top of each of the UNet's seven blocks -- three down, mid, three up: unet_2d_blocks.py:40,102,180,250,345).  Before a block runs, every
 * sample's block input (for the up blocks also each skip tensor it consumes) is compared with the input that block saw at its last run: the
 * mean squared difference.  The host predictor sees the reference's feature rows [block index, timestep, mse (, mse of each skip: oldest skip
 * first, the first-consumed one last, as res_hidden_states_tuple is ordered: cache_manager.py:110-121)] and answers
 * run / reuse per sample; a reused block's outputs (hidden state and, for the down blocks, the skip tensors) come from the cache.
 * Approximate by design and OFF on the exact path (mx_unet_forward never consults it).
 * Granularity of THIS entry (mx_unet_forward_cached, one resolution): the SAMPLE (a request's CFG row).  A block is skipped as a whole only when
 * NO sample asks to run it (save_and_get_block_states: mask.sum() == 0, cache_manager.py:60-67); when it runs, a sample that did not ask keeps the
 * OUTPUTS the block produced for it at its last run (the block is evaluated for the whole batch and those rows are restored from the state).
 * Round 4: this sample unit is this library's own reading, kept for callers that run unsliced -- the reference has no working unsliced cache (the
 * unsliced branches of its attention hand update_and_return ALL rows together with a partial mask, attention.py:204-224, which cannot broadcast),
 * and inside a running block it does not restore block outputs: GroupNorm, LayerNorms, projections, the feed-forward and the residual paths of a
 * patch that did not ask are computed fresh and only its convolutions and attention sub-blocks reuse their own cached outputs.  That behaviour, at
 * the reference's unit (the 256-px patch, is_sliced=True) and with the compute of the not-asking patches actually skipped, is
 * mx_unet_forward_cached_mixed below; the MMDiT's is mx_mmdit_forward_cached_mixed.  The cached state belongs to one batch composition: a different
 * batch_key, batch size or latent size invalidates it (every block runs once and refills it).  Not combined with patch parallelism.
 *   predict(ctx, block, is_up, n_samples, n_feat, timesteps[n], mse[n * n_feat], run_out[n]): mse = MX_MSE_UNCACHED when the block has no
 *   cached input; return non-zero to abort the forward.  The reference's predictors are cuML random forests that are not loadable here:
 *   sduss_amd/block_cache.py ships a threshold rule with the reference's forced recompute after four reuses (cache_manager.py:134,154) and
 *   takes any object with .predict(features).
 * ------------------------------------------------------------------------------------------ */
#define MX_MSE_UNCACHED 9.2233720368547758e18f      /* float(sys.maxsize), cache_manager.py:19 */
typedef int (*mx_skip_predict_fn)(void* ctx, int block, int is_up, int n_samples, int n_feat, const float* timesteps, const float* mse,
                                  unsigned char* run_out);
/* Host-side (no GPU) evaluation of a fitted binary random forest, the form the reference's predictors have (cuML forests; here fitted with
 * scikit-learn by tools/fit_skip_predictor.py and flattened by sduss_amd/block_cache.py CompiledForest): all trees' nodes in one array, node i
 * goes to left[i] when x[feature[i]] <= threshold[i] (x in fp32, as scikit-learn evaluates), else right[i]; left[i] < 0 marks a leaf whose
 * class-1 probability is p1[i]; out[r] = mean over trees > 0.5.  A mx_skip_predict_fn built on it costs microseconds per block. */
int mx_forest_predict(const int32_t* left, const int32_t* right, const int32_t* feature, const double* threshold, const double* p1,
                      const int32_t* roots, int n_trees, const float* X, int n_rows, int n_feat, unsigned char* out);
typedef void (*mx_skip_observe_fn)(void* ctx, int block, int n_samples, const float* out_mse);
typedef struct mx_block_cache {
  mx_skip_predict_fn predict;
  void* ctx;
  void* state;                /* device memory, mx_unet_block_cache_bytes(...) bytes, kept by the caller across steps */
  size_t state_bytes;
  uint64_t batch_key;         /* in: identifies the batch composition (e.g. a hash of the request ids in row order) */
  uint64_t cached_key;        /* library-owned from here on: zero-initialise the struct once */
  int cached_valid, cached_batch, cached_h, cached_w;
  unsigned blocks_run;        /* out: bit i set = block i ran in the last forward */
  unsigned blocks_run_hi;     /* out: blocks 32..63 (MMDiT) */
  mx_skip_observe_fn observe; /* optional (NULL): after a block RAN while its previous output was cached, the per-sample mean squared difference
                                 between the new and the previous hidden-state output -- the label a predictor is fitted on (the reference's
                                 files are named for it: exp/sdxl-upsample-threshold0.01.pkl); tools/fit_skip_predictor.py */
  /* Optional: one state row per REQUEST instead of per batch position -- what the reference's dictionaries keyed by request id give
   * (cache_manager.py:105-133: a request that stays while the batch around it changes keeps its cached tensors; a new one has none and makes
   * the block run).  slots (host array [batch]): the state row of each sample, distinct, in [0, n_slots); slot_valid (host array [batch]):
   * 1 = that row holds this sample's tensors of an earlier step at this latent size.  The caller owns the request -> row table
   * (sduss_amd/block_cache.py); the state is then mx_*_block_cache_bytes(u, n_slots, ...) large and batch_key / cached_* are not consulted. */
  const int32_t* slots;
  const unsigned char* slot_valid;
  int n_slots;
  /* mx_unet_forward_cached_mixed (the patch unit): the latent size the state rows are laid out for (every group's H, W <= these; multiples of
   * gn_patch), and, out, the patches that asked / the patches seen over the blocks of the last forward */
  int max_h, max_w;
  unsigned long long patches_asked, patches_total;
} mx_block_cache;
size_t mx_unet_block_cache_bytes(const mx_unet* u, int batch, int H, int W);
int mx_unet_forward_cached(mx_unet* u, void* stream, const void* latents, int io_dtype, const float* timesteps,
                           const void* encoder_hidden_states, const void* text_embeds, const float* time_ids, void* out,
                           int batch, int H, int W, int ctx_len, int gn_patch, void* workspace, size_t workspace_bytes, mx_block_cache* cache);
/* The cache AT THE REFERENCE'S OWN UNIT -- the 256-px patch -- over a mixed-resolution batch in ONE launch sequence (round 4).  The reference's
 * cache only works with is_sliced=True (its unsliced update_and_return receives all rows with a partial mask), which is what its mixed policies
 * force (policy/FCFS_Mixed.py:69-70); every cache is keyed "<request id>-<h>-<w>" (modules/utils.py:37,60; unet.py:163).  Semantics reproduced:
 *   - per block ONE decision for the patches of all samples of all groups: predict(ctx, block, is_up, n_patches, n_feat, timesteps[n_patches],
 *     mse[n_patches * n_feat], run_out[n_patches]) -- rows in the reference's order (group, sample, patch row, patch column), mse = mean over the
 *     patch's pixels and channels of (input - cached input)^2, for the up blocks also of each skip tensor (oldest first); MX_MSE_UNCACHED for a
 *     request without cached tensors (slot_valid 0), which always runs;
 *   - a block none of whose patches asks is skipped: every output (hidden state, the down blocks' skip tensors) comes from the state;
 *   - in a running block the ops that own a CacheManager in the reference -- resnet conv1 / conv2, the down / upsampler conv, attn1's core +
 *     to_out, the whole attn2 -- are COMPUTED FOR THE ASKING PATCHES ONLY (3x3 convs on a compact batch of halo'd patches gathered with the
 *     reference's halo / corner rule, per-token work on the compact rows, self-attention of the asking queries against all keys of their latent)
 *     and every other patch takes that op's own cached output; GroupNorm (statistics, halos, SiLU), the LayerNorms, proj_in / proj_out, the
 *     q | k | v projection, the feed-forward, the 1x1 shortcut, the time-embedding add and the residual adds run on all rows, on the fresh tensors
 *     (resnet.py:390-460, 280-378; transformer.py:32-128, 167-290; attention.py:59-232; cache_manager.py:84-99).
 * State: one row per request (slots / slot_valid / n_slots as above, required), rows laid out for max_h x max_w latents;
 * mx_unet_patch_cache_bytes(u, n_slots, max_h, max_w, gn_patch) bytes.  Needs gn_patch > 0 with every group's H, W multiples of it.  The
 * workspace is larger than mx_unet_workspace_bytes_mixed (compact patch batches): mx_unet_workspace_bytes_cached_mixed.  Not graph-captured
 * (one host decision per block).  Oracle: oracle/cache_patch_ref.py; tests/test_block_cache_gpu.py. */
size_t mx_unet_patch_cache_bytes(const mx_unet* u, int n_slots, int max_h, int max_w, int gn_patch);
size_t mx_unet_workspace_bytes_cached_mixed(const mx_unet* u, const mx_unet_group* groups, int n_groups, int ctx_len, int gn_patch);
int mx_unet_forward_cached_mixed(mx_unet* u, void* stream, const mx_unet_group* groups, int n_groups, int io_dtype, const float* timesteps,
                                 const void* encoder_hidden_states, const void* text_embeds, const float* time_ids, int ctx_len, int gn_patch,
                                 void* workspace, size_t workspace_bytes, mx_block_cache* cache);
/* The same for the SD3 / SD3.5 transformer: one cache point per joint block (SD3Transformer.py:54-57, 151, 172, 219-228 with
 * cache_manager.py:163-191): the feature row is [block index, timestep, mse of the image stream]; a reused block restores both the image and
 * the context stream it produced.  The reference forces a run after TWO reuses here (cache_manager.py:184). */
size_t mx_mmdit_block_cache_bytes(const mx_mmdit* u, int batch, int H, int W, int ctx_len);
int mx_mmdit_forward_cached(mx_mmdit* u, void* stream, const void* latents, int io_dtype, const float* timesteps,
                            const void* encoder_hidden_states, const void* pooled_projections, void* out, int batch, int H, int W,
                            int ctx_len, void* workspace, size_t workspace_bytes, mx_block_cache* cache);

/* The MMDiT cache at the reference's unit over a mixed batch in ONE launch sequence (round 4).  The unit is the token CHUNK: the sliced branch cuts
 * every latent into (res / patch)^2 equal token ranges keyed "<request id>-<k>" (modules/utils.py:86-122); `patch` is that patch edge in LATENT
 * pixels (patch_size / 8 of the reference's call).  Per joint block ONE predict call for the chunks of all samples of all groups (rows: group,
 * sample, chunk; n_feat 1; forced run after two reuses is the host's, cache_manager.py:163-191); a block none of whose chunks asks takes the image
 * and the text stream from the state (SD3Transformer.py:151-228).  In a running block every op runs on all tokens except the attention
 * (attention.py:296-372, 407-415): a resolution group with no asking chunk skips its joint attention and takes attn.output's cached to_out result
 * (image tokens) and attn.encoder_output's cached to_add_out result (text tokens); a group with any asking chunk computes it whole.  The image-only
 * attn2 of the dual blocks does the same, and a group whose asking ratio is <= 1/16 renews its asking chunks only (:303-325).  State: one row per
 * request (slots / slot_valid / n_slots / max_h / max_w as for mx_unet_forward_cached_mixed).  Oracle: oracle/cache_patch_ref.CachedSlicedMMDiTRef. */
size_t mx_mmdit_patch_cache_bytes(const mx_mmdit* u, int n_slots, int max_h, int max_w, int patch, int ctx_len);
size_t mx_mmdit_workspace_bytes_cached_mixed(const mx_mmdit* u, const mx_unet_group* groups, int n_groups, int ctx_len, int patch);
int mx_mmdit_forward_cached_mixed(mx_mmdit* u, void* stream, const mx_unet_group* groups, int n_groups, int io_dtype, const float* timesteps,
                                  const void* encoder_hidden_states, const void* pooled_projections, int ctx_len, int patch, void* workspace,
                                  size_t workspace_bytes, mx_block_cache* cache);

/* ------------------------------------------------------------------------------------------
 * The element-wise steps either side of the model call.
 * ------------------------------------------------------------------------------------------ */
/* out[b] = x[b mod n_lat] / sqrt(sigma[b mod n_lat]^2 + 1) for b in [0, n_rows)   (batch_scale_model_input, CFG
 * duplication of the latents fused: n_rows = 2*n_lat under CFG, pipeline_..._esymred.py:327) */
int mx_euler_scale_input(void* stream, const void* latents, void* out, const float* sigma,
                         int n_lat, int n_rows, int64_t elems_per_latent, int dtype);
/* latents <- latents + ((u + g (t - u))) * (sigma_next - sigma), fp32 math, stored in `dtype`
 * noise: [2*n_lat, elems] = [uncond..., cond...] (guidance > 0) ; epsilon-prediction Euler step */
int mx_cfg_euler_step(void* stream, const void* noise, void* latents, const float* sigma, const float* sigma_next,
                      float guidance_scale, int n_lat, int64_t elems_per_latent, int dtype);

/* latents <- latents + (sigma_next - sigma) * (u + g (t - u))   (FlowMatchEulerDiscreteScheduler.batch_step,
 * schedulers/scheduling_flow_match_euler_discrete.py:159-202; CFG combine pipeline_stable_diffusion_3_esymred.py:365-367) */
int mx_cfg_flow_step(void* stream, const void* noise, void* latents, const float* sigma, const float* sigma_next,
                     float guidance_scale, int n_lat, int64_t elems_per_latent, int dtype);

#ifdef __cplusplus
}
#endif
#endif /* MXDENOISE_H */
