// Step plan of the SD3.5 MMDiT in the sduss ``transformer`` slot: a flat launch sequence on one HIP stream over
// token-major bf16 activations in a caller-provided workspace.
//
// Replaces PatchSD3Transformer2DModel.forward (sduss/model_executor/modules/SD3Transformer.py:60-262) and the blocks it
// drives (PatchJointTransformerBlock modules/transformer.py:299-388, PatchSD3Attention modules/attention.py:241-424).
// With the block cache off the reference's sliced branch only re-chunks and regroups the token axis, so one plan serves
// is_sliced True and False (oracle/sd3_mmdit_ref.py).
// MI355X-first choices:
//   * every AdaLN projection of the step (24 x norm1 / norm1_context + norm_out, 325 d columns) is ONE GEMM on
//     silu(temb); LayerNorm-modulate kernels and GEMM epilogues (gated residual) read their fp32 rows from it;
//   * image and text tokens share one joint Q|K buffer, one V^T buffer and one O buffer per step: the two QKV GEMMs
//     write their rows straight into the joint sequence (row remap in the epilogue, V transposed), so the joint
//     attention of attention.py:347-350 is a single kernel launch with no torch.cat; to_out / to_add_out read their
//     token ranges back through the loader's row remap;
//   * PatchEmbed's 2x2 stride-2 conv is an im2col + GEMM whose epilogue adds bias and the cropped positional table.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <deque>
#include <functional>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/mxdenoise.h"
#include "common.h"
#include <map>

#include "graph_cache.h"
#include "pp_exchange.h"
#include "patch_cache.h"

namespace mx {
int launch_patchify(hipStream_t s, const void* in, int dtype, void* out, int B, int C, int H, int W, int ps);
int launch_unpatchify(hipStream_t s, const void* in, void* out, int dtype, int B, int C, int H, int W, int ps, int ld);
int launch_crop_pos(hipStream_t s, const void* table, void* out, int m, int h, int w, int d);
int launch_sinus_embed(hipStream_t s, const float* t, void* out, int B, int dim);
int launch_sq_diff_partial(hipStream_t s, const void* a, const void* b, long elems_per_sample, int B, double* partial, const int* slot = nullptr);
int launch_copy_rows(hipStream_t s, void* batch, void* slotted, size_t bytes_per_sample, int B, const int* slot, int scatter);
}  // namespace mx

using mx::bf16_t;

struct mx_mmdit {
  mx_mmdit_config cfg;
  const char* blob = nullptr;
  uint64_t blob_bytes = 0;
  std::unordered_map<std::string, std::pair<uint64_t, uint64_t>> table;
  mx::GraphCache graphs;   // hipGraph replay of the forward, keyed by its arguments (graph_cache.h)
  std::map<std::vector<long>, std::vector<size_t>> pp_sizes;   // recorded exchange sizes of the patch-parallel plan per shape (as in unet_sdxl.cpp)
};

namespace {

struct Arena {
  char* base; size_t cap; size_t top; size_t peak; bool dry;
  void* alloc(size_t bytes) {
    const size_t a = (top + 255) & ~(size_t)255;
    top = a + bytes;
    if (top > peak) peak = top;
    if (dry) return (void*)(uintptr_t)(0x1000 + a);
    return (top <= cap) ? base + a : nullptr;
  }
};

struct Plan {
  mx_mmdit* u;
  hipStream_t stream;
  Arena ar;
  int B, H, W, Lt;                 // B = samples of ALL groups; H, W = the first group's latent size (the only one unless mixed)
  // Mixed-resolution batch (mx_mmdit_forward_mixed): the requests of every resolution present in ONE launch sequence.  A group = the samples of
  // one resolution.  The image stream is the groups' token rows one after the other, the text stream is per sample (same length everywhere);
  // the joint q|k / V^T / O buffers hold each group's [image ; text] sequences at its own length.  Per-token ops are single launches; the ops
  // with per-sample structure -- AdaLN modulation, the q|k|v projections into the joint buffers, attention, the gated projections reading the
  // joint sequence back -- are GROUPED launches (mx_gemm_seg, mx_attention_prescaled_grouped, mx_layernorm_mod_grouped).  The reference
  // re-chunks the tokens of all resolutions into one batch (modules/utils.py:86-122) and regroups them per latent before attention
  // (attention.py:300-372).
  int ng = 1;
  int gB[MX_MAX_SEGS], gH[MX_MAX_SEGS], gW[MX_MAX_SEGS], gb0[MX_MAX_SEGS];
  const void* g_lat[MX_MAX_SEGS]; void* g_out[MX_MAX_SEGS];
  void set_single(int batch, int h, int w, const void* lat, void* out) { ng = 1; gB[0] = batch; gH[0] = h; gW[0] = w; gb0[0] = 0; g_lat[0] = lat; g_out[0] = out; B = batch; H = h; W = w; }
  typedef std::function<void(int, mx_gemm_seg&)> SegFill;      // fills problem g of a grouped GEMM
  mx_gemm_seg seg_buf[MX_MAX_SEGS];
  const bool* act = nullptr;        // cached mixed batch: the grouped launch covers these resolution groups only (nullptr: all)
  void attach(mx_gemm_desc& d, const SegFill& fill) {
    if (ng <= 1 || !fill) return;
    std::memset(seg_buf, 0, sizeof(seg_buf));
    int n = 0;
    for (int g = 0; g < ng; ++g) if (!act || act[g]) fill(g, seg_buf[n++]);
    d.segs = seg_buf; d.n_segs = n;
  }
  bool dry, lookup = false;
  bool mute = false;               // block-skip cache: the block is reused, nothing of it is launched
  bool quiet() const { return dry || mute; }
  // patch parallelism (mx_mmdit_forward_pp; distrifuser models/distri_sd3_transformer_pp.py:87-97, modules/pp/attn.py:202-277): this rank owns
  // the image tokens of H (local) latent rows; the text stream is computed by every rank; K / V^T of the image tokens are all-gathered
  mx::PPExchange px;
  bool is_pp() const { return px.world > 1; }
  bool all_gather(const void* send, void* recv, size_t bytes_per_rank) {
    if (!ok()) return false;
    if (const char* e = px.all_gather(stream, dry, send, recv, bytes_per_rank)) return fail(e);
    return true;
  }
  bool copy2d(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height) {
    if (!ok()) return false;
    if (quiet()) return true;
    if (hipMemcpy2DAsync(dst, dpitch, src, spitch, width, height, hipMemcpyDeviceToDevice, stream) != hipSuccess) return fail("patch-parallel: K / V assembly copy failed");
    return true;
  }
  // ---- the cache at the reference's unit over a mixed batch in ONE launch sequence (mx_mmdit_forward_cached_mixed) ----
  // The unit is the token CHUNK: every latent is cut into (res // patch)^2 equal token ranges keyed "<request id>-<k>" (modules/utils.py:86-122).
  // Per joint block ONE decision for the chunks of all samples of all groups (get_sd3_mask, cache_manager.py:163-191); a block none of whose
  // chunks asks takes both streams from the state; in a running block everything runs on all tokens except the attention: a resolution group
  // with no asking chunk skips it and takes attn.output / attn.encoder_output's cached results, a group with any asking chunk computes it whole
  // (attention.py:296-372; attn2 of the dual blocks: asking ratio <= 1/16 -> only the asking chunks are renewed, :303-325).
  bool pcm = false;
  int pcm_patch = 0, pcm_slots = 0, pcm_maxh = 0, pcm_maxw = 0, pcm_nc = 0;
  std::vector<mx::PcSample> pcm_img, pcm_ctx;
  std::vector<mx::PcRange> pcm_chunks;
  std::vector<int> pcm_chunk_b, pcm_chunk_g;
  mx::PcSample* pcm_dimg = nullptr; mx::PcSample* pcm_dctx = nullptr; mx::PcRange* pcm_dchunks = nullptr; mx::PcRange* pcm_dtmp = nullptr; double* pcm_dpart = nullptr;
  unsigned long long pcm_asked = 0, pcm_total = 0;
  // host copies of the range tables sent by hipMemcpyAsync: kept until the forward's final synchronisation (a pageable source must outlive the copy)
  std::deque<std::vector<mx::PcRange>> pcm_sent;
  ~Plan() { if (!pcm_sent.empty() && !dry) (void)hipStreamSynchronize(stream); }     // (an early error return: the copies may still be reading)
  size_t pcm_ncmax() const { return (size_t)pcm_slots * (pcm_maxh / pcm_patch) * (pcm_maxw / pcm_patch); }
  size_t pcm_head_bytes() const {
    const size_t nc = pcm_ncmax();
    return ((nc * 64 * sizeof(double) + 2 * (size_t)pcm_slots * sizeof(mx::PcSample) + 4 * nc * sizeof(mx::PcRange)) + 255) & ~(size_t)255;
  }
  mx_block_cache* bc = nullptr;    // mx_mmdit_forward_cached
  size_t bc_bytes = 0;             // state bytes the plan needs (also the dry answer of mx_mmdit_block_cache_bytes)
  unsigned long long blocks_run = 0;
  std::vector<float> h_timesteps;
  int bc_rows = 0;                 // samples a state tensor holds: the batch, or bc->n_slots with one slot per request
  const int* bc_dslot = nullptr;   // device copy of bc->slots
  std::vector<unsigned char> bc_valid;
  bool bc_all_valid = false, bc_any_valid = false;
  static size_t bc_scratch_bytes(int rows) { return ((size_t)rows * 64 * sizeof(double) + (size_t)rows * sizeof(int) + 255) & ~(size_t)255; }
  bool bc_store(char* region, const void* t, size_t per_sample_bytes) {
    if (bc_dslot) { if (mx::launch_copy_rows(stream, (void*)t, region, per_sample_bytes, B, bc_dslot, 1)) return fail(mx_last_error()); return true; }
    if (hipMemcpyAsync(region, t, per_sample_bytes * B, hipMemcpyDeviceToDevice, stream) != hipSuccess) return fail("block cache: copy into the state failed");
    return true;
  }
  bool bc_load(void* t, char* region, size_t per_sample_bytes) {
    if (bc_dslot) { if (mx::launch_copy_rows(stream, t, region, per_sample_bytes, B, bc_dslot, 0)) return fail(mx_last_error()); return true; }
    if (hipMemcpyAsync(t, region, per_sample_bytes * B, hipMemcpyDeviceToDevice, stream) != hipSuccess) return fail("block cache: copy out of the state failed");
    return true;
  }
  const char* stage = nullptr; void* stage_out = nullptr; size_t stage_bytes = 0; bool stage_hit = false;
  std::string err;

  bool fail(const std::string& m) { if (err.empty()) err = m; return false; }
  bool ok() const { return err.empty(); }
  const void* w(const std::string& name, size_t expect_bytes) {
    if (dry && !lookup) return (const void*)(uintptr_t)0x1000;
    auto it = u->table.find(name);
    if (it == u->table.end()) { fail("missing weight '" + name + "'"); return nullptr; }
    if (it->second.second != expect_bytes) {
      fail("weight '" + name + "' has " + std::to_string(it->second.second) + " bytes, expected " + std::to_string(expect_bytes));
      return nullptr;
    }
    return u->blob + it->second.first;
  }
  const bf16_t* wb(const std::string& name, size_t elems) { return (const bf16_t*)w(name, elems * 2); }
  const float* wf(const std::string& name, size_t elems) { return (const float*)w(name, elems * 4); }
  template <typename T> T* alloc(size_t elems) {
    T* p = (T*)ar.alloc(elems * sizeof(T));
    if (!p) fail("workspace too small");
    return p;
  }
  bool gemm(mx_gemm_desc& d) {
    if (!ok()) return false;
    if (quiet()) return true;
    if (mx_gemm(stream, &d)) return fail(std::string("gemm: ") + mx_last_error());
    return true;
  }
  // C = A W^T + bias with the optional fused pieces
  bool linear(const void* a, int lda, const std::string& name, void* c, int ldc, int M, int N, int K, int flags = 0,
              const void* residual = nullptr, int ldr = 0, const float* gate = nullptr, int ldg = 0, int rows_per_batch = 0,
              int a_batch_rows = 0, int a_row_off = 0, const SegFill& fill = nullptr) {
    mx_gemm_desc d; std::memset(&d, 0, sizeof(d));
    d.a = a; d.lda = lda; d.w = wb(name + ".weight", (size_t)N * K); d.bias = wf(name + ".bias", N);
    d.c = c; d.ldc = ldc; d.M = M; d.N = N; d.K = K; d.flags = flags; d.residual = residual; d.ldr = ldr;
    d.gate = gate; d.ldg = ldg; d.rows_per_batch = rows_per_batch; d.a_batch_rows = a_batch_rows; d.a_row_off = a_row_off;
    attach(d, fill);
    return gemm(d);
  }
  // fused q|k|v projection of `rows_per_batch` tokens per sample into the joint buffers at row offset `row_off`; the epilogue
  // RMS-normalises every q / k head (norm_q / norm_k, attention.py:332-346, 377-388) and scales q for mx_attention_prescaled
  bool qkv(const void* a, const std::string& name, const std::string& qnorm, const std::string& knorm, bf16_t* qk, bf16_t* vt, int ldvt,
           int M, int d_model, int rows_per_batch, int joint_rows, int row_off, const SegFill& fill = nullptr) {
    mx_gemm_desc d; std::memset(&d, 0, sizeof(d));
    d.a = a; d.lda = d_model; d.w = wb(name + ".weight", (size_t)3 * d_model * d_model); d.bias = wf(name + ".bias", 3 * d_model);
    d.c = qk; d.ldc = 2 * d_model; d.M = M; d.N = 3 * d_model; d.K = d_model; d.flags = MX_EPI_QKV | MX_EPI_RMSNORM; d.seg = d_model; d.period = 3;
    d.vt = vt; d.ldvt = ldvt; d.rows_per_batch = rows_per_batch; d.c_batch_rows = joint_rows; d.c_row_off = row_off;
    d.rms_wq = wf(qnorm, 64); d.rms_wk = wf(knorm, 64); d.rms_eps = u->cfg.norm_eps; d.out_scale = MX_ATTN_QSCALE(0.125f);
    attach(d, fill);
    return gemm(d);
  }
  // image: the rows are the image stream (per-group tokens per sample in a mixed batch); otherwise the text stream (Lt rows per sample)
  bool lnmod(const bf16_t* x, bf16_t* y, bf16_t* y2, const float* scale, const float* shift, const float* scale2,
             const float* shift2, int ldmod, int M, int C, int rows_per_batch, bool image = false) {
    if (!ok()) return false;
    if (quiet()) return true;
    if (image && ng > 1) {
      int rpb[MX_MAX_SEGS];
      const int ps = u->cfg.patch_size;
      for (int g = 0; g < ng; ++g) rpb[g] = (gH[g] / ps) * (gW[g] / ps);
      if (mx_layernorm_mod_grouped(stream, x, y, y2, scale, shift, scale2, shift2, ldmod, C, u->cfg.norm_eps, gB, rpb, ng))
        return fail(std::string("layernorm_mod: ") + mx_last_error());
      return true;
    }
    if (mx_layernorm_mod(stream, x, y, y2, scale, shift, scale2, shift2, ldmod, M, C, rows_per_batch, u->cfg.norm_eps))
      return fail(std::string("layernorm_mod: ") + mx_last_error());
    return true;
  }
  bool attention(const bf16_t* qk, int d_model, const bf16_t* vt, int ldvt, bf16_t* o, int heads, int L) {
    if (!ok()) return false;
    if (quiet()) return true;
    // q carries MX_ATTN_QSCALE(1/8) from the QKV epilogue (RMSNorm + out_scale)
    if (mx_attention_prescaled(stream, qk, 2 * d_model, qk + d_model, 2 * d_model, vt, ldvt, (int64_t)d_model * ldvt, o, d_model, B, heads, L, L))
      return fail(std::string("attention: ") + mx_last_error());
    return true;
  }
  // queries from the local q|k rows, keys from a packed K buffer of another length (patch-parallel: local queries, gathered keys)
  bool attention_qk(const bf16_t* q, const bf16_t* k, int d_model, const bf16_t* vt, int ldvt, bf16_t* o, int heads, int Lq, int Lk) {
    if (!ok()) return false;
    if (quiet()) return true;
    if (mx_attention_prescaled(stream, q, 2 * d_model, k, d_model, vt, ldvt, (int64_t)d_model * ldvt, o, d_model, B, heads, Lq, Lk))
      return fail(std::string("attention: ") + mx_last_error());
    return true;
  }
  void dump(const std::string& name, const bf16_t* t, size_t elems) {
    if (!stage || quiet() || !ok() || stage_hit) return;
    if (name != stage) return;
    if (elems * 2 > stage_bytes) { fail("stage buffer too small for '" + name + "'"); return; }
    if (hipMemcpyAsync(stage_out, t, elems * 2, hipMemcpyDeviceToDevice, stream) != hipSuccess) fail("stage copy failed");
    stage_hit = true;
  }

  bool run(const void* latents, int io_dtype, const float* timesteps, const void* ehs, const void* pooled, void* outp) {
    const mx_mmdit_config& c = u->cfg;
    const int d = c.num_attention_heads * 64;
    const int heads = c.num_attention_heads;
    const int ps = c.patch_size;
    const int h = H / ps, wd = W / ps;
    const int L = h * wd;
    const int Lj = L + Lt;
    // V^T rows padded to whole 128-byte lines: with the minimal MX_VT_LD(4429) = 4432 every 64-key tile row straddles two lines and the
    // joint attention ran at 850 TFLOP/s against 1000 at an aligned length (tools/exp/attn_shapes_probe.py)
    const int ldvt_j = (MX_VT_LD(Lj) + 63) / 64 * 64;
    const int ldvt_i = (MX_VT_LD(L) + 63) / 64 * 64;
    const int Kp = ps * ps * c.in_channels;
    // per group (one unless the batch is mixed): image tokens per sample, joint sequence, V^T row lengths, first rows / elements of the group
    // in the image stream, in the joint q|k / O buffers and in the two V^T buffers
    int gL[MX_MAX_SEGS], gLj[MX_MAX_SEGS], gldj[MX_MAX_SEGS], gldi[MX_MAX_SEGS];
    long r0[MX_MAX_SEGS], jr0[MX_MAX_SEGS], vj0[MX_MAX_SEGS], vi0[MX_MAX_SEGS];
    long rows_i = 0, rows_j = 0, el_vj = 0, el_vi = 0;
    for (int g = 0; g < ng; ++g) {
      gL[g] = (gH[g] / ps) * (gW[g] / ps); gLj[g] = gL[g] + Lt;
      gldj[g] = (MX_VT_LD(gLj[g]) + 63) / 64 * 64; gldi[g] = (MX_VT_LD(gL[g]) + 63) / 64 * 64;
      r0[g] = rows_i; jr0[g] = rows_j; vj0[g] = el_vj; vi0[g] = el_vi;
      rows_i += (long)gB[g] * gL[g]; rows_j += (long)gB[g] * gLj[g]; el_vj += (long)gB[g] * d * gldj[g]; el_vi += (long)gB[g] * d * gldi[g];
    }
    if (ng > 1 && (is_pp() || (bc && !pcm))) return fail("mmdit: a mixed-resolution batch runs neither patch-parallel nor through the per-sample block cache");
    const int MI = (int)rows_i, MT = B * Lt;
    // patch-parallel: L counts this rank's image tokens; the keys of the joint attention are all ranks' image tokens, then the text tokens
    const int world = px.world;
    const int Ltot = L * world, Ljt = Ltot + Lt;
    const int ldvt_jt = (MX_VT_LD(Ljt) + 63) / 64 * 64, ldvt_it = (MX_VT_LD(Ltot) + 63) / 64 * 64;
    if (is_pp() && L % 16 != 0) return fail("mmdit pp: local image tokens must be a multiple of 16 (V^T key order, MX_VT_POS)");
    if (is_pp() && bc) return fail("mmdit pp: not combined with the block-skip cache");

    // ---- AdaLN column layout of the one projection GEMM ----
    std::vector<int> off_img(c.num_layers), off_ctx(c.num_layers);
    int ntot = 0;
    for (int i = 0; i < c.num_layers; ++i) {
      const bool dual = c.dual_attention[i] != 0, last = i == c.num_layers - 1;
      off_img[i] = ntot; ntot += (dual ? 9 : 6) * d;
      off_ctx[i] = ntot; ntot += (last ? 2 : 6) * d;
    }
    const int off_out = ntot; ntot += 2 * d;

    // ---- conditioning (SD3Transformer.py:81): temb = timestep_embedder(sinusoid(t)) + text_embedder(pooled) ----
    bf16_t* tsin = alloc<bf16_t>((size_t)B * 256);
    if (ok() && !dry && mx::launch_sinus_embed(stream, timesteps, tsin, B, 256)) fail(mx_last_error());
    bf16_t* t1 = alloc<bf16_t>((size_t)B * d); bf16_t* t2 = alloc<bf16_t>((size_t)B * d);
    bf16_t* p1 = alloc<bf16_t>((size_t)B * d); bf16_t* semb = alloc<bf16_t>((size_t)B * d);
    linear(tsin, 256, "time_text_embed.timestep_embedder.linear_1", t1, d, B, d, 256, MX_EPI_SILU);
    linear(t1, d, "time_text_embed.timestep_embedder.linear_2", t2, d, B, d, d);
    linear(pooled, c.pooled_projection_dim, "time_text_embed.text_embedder.linear_1", p1, d, B, d, c.pooled_projection_dim, MX_EPI_SILU);
    // silu(temb): every consumer of temb applies SiLU first (AdaLayerNormZero / ZeroX / Continuous)
    linear(p1, d, "time_text_embed.text_embedder.linear_2", semb, d, B, d, d, MX_EPI_SILU, t2, d);
    float* mod = alloc<float>((size_t)B * ntot);
    linear(semb, d, "adaln_all", mod, ntot, B, ntot, d, MX_EPI_OUT_F32);

    // ---- PatchEmbed + positional table (:82-83), context_embedder (:115) ----
    bf16_t* patches = alloc<bf16_t>((size_t)MI * Kp);
    for (int g = 0; g < ng && ok() && !dry; ++g)
      if (mx::launch_patchify(stream, g_lat[g], io_dtype, patches + r0[g] * Kp, gB[g], c.in_channels, gH[g], gW[g], ps)) fail(mx_last_error());
    bf16_t* pos = alloc<bf16_t>(ng > 1 ? (size_t)MI * d : (size_t)Ltot * d);      // (mixed: one cropped table per group, [L_g, d] at the group's first row)
    if (ng > 1) {
      const bf16_t* table = wb("pos_embed.table", (size_t)c.pos_embed_max_size * c.pos_embed_max_size * d);
      for (int g = 0; g < ng && ok(); ++g) {
        if (gH[g] / ps > c.pos_embed_max_size || gW[g] / ps > c.pos_embed_max_size) return fail("mmdit: latent larger than the positional table");
        if (!dry && mx::launch_crop_pos(stream, table, pos + r0[g] * d, c.pos_embed_max_size, gH[g] / ps, gW[g] / ps, d)) fail(mx_last_error());
      }
    } else {
      // the centre crop is taken for the WHOLE grid (distri_sd3_transformer_pp.py:87 embeds before it slices); this rank reads its rows
      const bf16_t* table = wb("pos_embed.table", (size_t)c.pos_embed_max_size * c.pos_embed_max_size * d);
      if (h * world > c.pos_embed_max_size) return fail("mmdit: latent larger than the positional table");
      if (ok() && !dry && mx::launch_crop_pos(stream, table, pos, c.pos_embed_max_size, h * world, wd, d)) fail(mx_last_error());
      if (pos) pos += (size_t)px.rank * L * d;
    }
    bf16_t* x = alloc<bf16_t>((size_t)MI * d);
    linear(patches, Kp, "pos_embed.proj", x, d, MI, d, Kp, MX_EPI_RES_BCAST, pos, d, nullptr, 0, L, 0, 0, [&](int g, mx_gemm_seg& q) {
      q.a = patches + r0[g] * Kp; q.c = x + r0[g] * d; q.residual = pos + r0[g] * d; q.M = gB[g] * gL[g]; q.rows_per_batch = gL[g]; });
    bf16_t* ctx = alloc<bf16_t>((size_t)MT * d);
    linear(ehs, c.joint_attention_dim, "context_embedder", ctx, d, MT, d, c.joint_attention_dim);
    dump("embed", x, (size_t)MI * d);
    dump("context_embed", ctx, (size_t)MT * d);

    // ---- per-step buffers reused by every block ----
    bf16_t* xin = alloc<bf16_t>((size_t)MI * d);
    bf16_t* x2in = alloc<bf16_t>((size_t)MI * d);
    bf16_t* cin = alloc<bf16_t>((size_t)MT * d);
    bf16_t* qk_j = alloc<bf16_t>((size_t)rows_j * 2 * d);
    bf16_t* vt_j = alloc<bf16_t>((size_t)el_vj);
    bf16_t* o_j = alloc<bf16_t>((size_t)rows_j * d);
    bf16_t* qk_i = alloc<bf16_t>((size_t)MI * 2 * d);
    bf16_t* vt_i = alloc<bf16_t>((size_t)el_vi);
    bf16_t* o_i = alloc<bf16_t>((size_t)MI * d);
    bf16_t* ff = alloc<bf16_t>((size_t)MI * 4 * d);
    bf16_t* ffc = alloc<bf16_t>((size_t)MT * 4 * d);
    bf16_t* pcm_ti = pcm ? alloc<bf16_t>((size_t)MI * d) : nullptr;          // cached mixed batch: to_out / to_add_out results before the gate
    bf16_t* pcm_tc = pcm ? alloc<bf16_t>((size_t)MT * d) : nullptr;
    char* pcm_top = pcm ? (dry ? (char*)(uintptr_t)0x1000 : (char*)bc->state) + pcm_head_bytes() : nullptr;
    // patch-parallel: only the image tokens' K rows and V^T columns travel (what distrifuser gathers, modules/pp/attn.py:222-233): they are packed
    // into contiguous send buffers first -- the QKV epilogue writes q|k interleaved and V^T rows padded, with the text tokens behind the image ones
    bf16_t *k_send = nullptr, *v_send = nullptr, *k_g = nullptr, *v_g = nullptr, *k_all = nullptr, *vt_all = nullptr;
    if (is_pp()) {
      k_send = alloc<bf16_t>((size_t)B * L * d);
      v_send = alloc<bf16_t>((size_t)B * d * L);
      k_g = alloc<bf16_t>((size_t)world * B * L * d);
      v_g = alloc<bf16_t>((size_t)world * B * d * L);
      k_all = alloc<bf16_t>((size_t)B * Ljt * d);
      vt_all = alloc<bf16_t>((size_t)B * d * ldvt_jt);
    }
    // keys of the joint attention per sample: every rank's `L` image tokens in rank order, then the `tail` local-only (text) tokens:
    // k_all [B][world * L + tail][d], vt_all [B][d][ld_all]
    auto gather_kv = [&](bf16_t* qk_loc, bf16_t* vt_loc, int ld_loc, int tail, int ld_all) {
      const int rows_loc = L + tail, rows_all = Ltot + tail;
      const size_t krow = (size_t)d * 2, qrow = 2 * krow;
      for (int b = 0; b < B && ok(); ++b)               // K half of this sample's image rows
        copy2d((char*)k_send + (size_t)b * L * krow, krow, (char*)qk_loc + (size_t)b * rows_loc * qrow + krow, qrow, krow, L);
      copy2d(v_send, (size_t)L * 2, vt_loc, (size_t)ld_loc * 2, (size_t)L * 2, (size_t)B * d);
      all_gather(k_send, k_g, (size_t)B * L * krow);
      all_gather(v_send, v_g, (size_t)B * d * L * 2);
      for (int r = 0; r < world && ok(); ++r) {
        copy2d((char*)k_all + (size_t)r * L * krow, rows_all * krow, (char*)k_g + (size_t)r * B * L * krow, L * krow, L * krow, B);
        copy2d((char*)vt_all + (size_t)r * L * 2, (size_t)ld_all * 2, (char*)v_g + (size_t)r * B * d * L * 2, (size_t)L * 2, (size_t)L * 2, (size_t)B * d);
      }
      if (tail) {
        for (int b = 0; b < B && ok(); ++b)
          copy2d((char*)k_all + ((size_t)b * rows_all + Ltot) * krow, krow, (char*)qk_loc + ((size_t)b * rows_loc + L) * qrow + krow, qrow, krow, tail);
        copy2d((char*)vt_all + (size_t)Ltot * 2, (size_t)ld_all * 2, (char*)vt_loc + (size_t)L * 2, (size_t)ld_loc * 2, (size_t)MX_VT_LD(tail) * 2,
               (size_t)B * d);
      }
    };

    // Block-skip cache (mx_mmdit_forward_cached; the reference's per-block CacheManagers, SD3Transformer.py:54-57,151,172,219-228): a block
    // runs when any sample asks (state_mask.sum() > 0), otherwise the image and context streams take the values the block produced last
    // time.  State per block: [input x | output x | output context], each 256-byte aligned, after the comparison scratch.
    if (bc && bc_rows < B) bc_rows = B;
    const size_t bc_x = ((size_t)bc_rows * L * d * 2 + 255) & ~(size_t)255, bc_c = ((size_t)bc_rows * Lt * d * 2 + 255) & ~(size_t)255;
    const size_t bc_scratch = bc_scratch_bytes(bc_rows);
    if (bc) {
      bc_bytes = bc_scratch + (size_t)c.num_layers * (2 * bc_x + bc_c);
      if (!dry && bc_bytes > bc->state_bytes) fail("block cache: state buffer too small (mx_mmdit_block_cache_bytes)");
    }
    // decides block i; false = reuse.  Leaves the latest input in the cache (cache_manager.py:183)
    auto decide = [&](int i) -> bool {
      char* st = (char*)bc->state + bc_scratch + (size_t)i * (2 * bc_x + bc_c);
      std::vector<float> mse(B, MX_MSE_UNCACHED);
      if (bc_any_valid) {
        double* part = (double*)bc->state;
        std::vector<double> hp((size_t)B * 64);
        if (mx::launch_sq_diff_partial(stream, x, st, (long)L * d, B, part, bc_dslot)) { fail(mx_last_error()); return true; }
        if (hipMemcpyAsync(hp.data(), part, hp.size() * sizeof(double), hipMemcpyDeviceToHost, stream) != hipSuccess ||
            hipStreamSynchronize(stream) != hipSuccess) { fail("block cache: reading the input differences failed"); return true; }
        for (int s = 0; s < B; ++s) {
          if (!bc_valid[s]) continue;
          double t = 0.0;
          for (int k = 0; k < 64; ++k) t += hp[(size_t)s * 64 + k];
          mse[s] = (float)(t / ((double)L * d));
        }
      }
      std::vector<unsigned char> run(B, 1);
      if (bc->predict(bc->ctx, i, 0, B, 1, h_timesteps.data(), mse.data(), run.data())) { fail("block cache: the predictor failed"); return true; }
      bool any = !bc_all_valid;
      for (int s = 0; s < B; ++s) any = any || run[s] != 0;
      bc_store(st, x, (size_t)L * d * 2);
      return any;
    };
    auto after = [&](int i, bool ran, bool last) {
      char* st = (char*)bc->state + bc_scratch + (size_t)i * (2 * bc_x + bc_c);
      if (ran && bc_all_valid && bc->observe) {              // how far the block's image-stream output moved since its last run (fitting labels)
        double* part = (double*)bc->state;
        std::vector<double> hp((size_t)B * 64);
        std::vector<float> om(B);
        if (mx::launch_sq_diff_partial(stream, x, st + bc_x, (long)L * d, B, part, bc_dslot)) { fail(mx_last_error()); return; }
        if (hipMemcpyAsync(hp.data(), part, hp.size() * sizeof(double), hipMemcpyDeviceToHost, stream) != hipSuccess ||
            hipStreamSynchronize(stream) != hipSuccess) { fail("block cache: reading the output differences failed"); return; }
        for (int s = 0; s < B; ++s) {
          double t = 0.0;
          for (int k = 0; k < 64; ++k) t += hp[(size_t)s * 64 + k];
          om[s] = (float)(t / ((double)L * d));
        }
        bc->observe(bc->ctx, i, B, om.data());
      }
      bool okc = ran ? bc_store(st + bc_x, x, (size_t)L * d * 2) : bc_load(x, st + bc_x, (size_t)L * d * 2);
      if (okc && !last) okc = ran ? bc_store(st + 2 * bc_x, ctx, (size_t)Lt * d * 2) : bc_load(ctx, st + 2 * bc_x, (size_t)Lt * d * 2);
      if (ran) blocks_run |= 1ull << i;
    };

    for (int i = 0; i < c.num_layers && ok(); ++i) {
      const std::string b = "transformer_blocks." + std::to_string(i);
      const bool dual = c.dual_attention[i] != 0, last = i == c.num_layers - 1;
      if (pcm) {
        const long Lmax = (long)(pcm_maxh / ps) * (pcm_maxw / ps);
        const long row_i = Lmax * d, row_c = (long)Lt * d;
        auto region = [&](long row_elems) { char* r = pcm_top; pcm_top += ((size_t)row_elems * pcm_slots * 2 + 255) & ~(size_t)255; return r; };
        char* r_in = region(row_i); char* r_out = region(row_i); char* r_octx = last ? nullptr : region(row_c);
        char* r_a = region(row_i); char* r_ae = last ? nullptr : region(row_c); char* r_a2 = dual ? region(row_i) : nullptr;
        if (dry) continue;
        if ((size_t)(pcm_top - (char*)bc->state) > bc->state_bytes) { fail("mmdit patch cache: state buffer too small (mx_mmdit_patch_cache_bytes)"); break; }
        long maxL = 0; for (int g = 0; g < ng; ++g) maxL = std::max<long>(maxL, gL[g]);
        auto img_copy = [&](void* t, char* reg, int to_batch, const float* vec, const void* res, int gate) {
          if (ok() && mx::launch_pc_image_copy(stream, t, reg, pcm_dimg, B, 0, d, row_i, to_batch, vec, ntot, res, maxL * d, gate)) fail(mx_last_error()); };
        auto ctx_copy = [&](void* t, char* reg, int to_batch, const float* vec, const void* res, int gate) {
          if (ok() && mx::launch_pc_image_copy(stream, t, reg, pcm_dctx, B, 0, d, row_c, to_batch, vec, ntot, res, (long)Lt * d, gate)) fail(mx_last_error()); };
        // ---- decision, per chunk ----
        const int NC = pcm_nc;
        std::vector<float> mse(NC, MX_MSE_UNCACHED);
        if (bc_any_valid) {
          std::vector<double> hp((size_t)NC * 64);
          if (mx::launch_pc_range_sq_diff(stream, x, r_in, row_i, d, pcm_dchunks, NC, pcm_dpart)) { fail(mx_last_error()); break; }
          if (hipMemcpyAsync(hp.data(), pcm_dpart, hp.size() * sizeof(double), hipMemcpyDeviceToHost, stream) != hipSuccess ||
              hipStreamSynchronize(stream) != hipSuccess) { fail("mmdit patch cache: reading the input differences failed"); break; }
          for (int j = 0; j < NC; ++j) {
            if (!bc_valid[pcm_chunk_b[j]]) continue;
            double t = 0.0;
            for (int k = 0; k < 64; ++k) t += hp[(size_t)j * 64 + k];
            mse[j] = (float)(t / ((double)pcm_chunks[j].rows * d));
          }
        }
        std::vector<unsigned char> run(NC, 1);
        std::vector<float> tpp(NC);
        for (int j = 0; j < NC; ++j) tpp[j] = h_timesteps[pcm_chunk_b[j]];
        if (bc->predict(bc->ctx, i, 0, NC, 1, tpp.data(), mse.data(), run.data())) { fail("mmdit patch cache: the predictor failed"); break; }
        bool gany[MX_MAX_SEGS] = {false, false, false, false};
        int gask[MX_MAX_SEGS] = {0, 0, 0, 0}, gtot[MX_MAX_SEGS] = {0, 0, 0, 0};
        bool any = false;
        for (int j = 0; j < NC; ++j) {
          if (!bc_valid[pcm_chunk_b[j]]) run[j] = 1;
          const int g = pcm_chunk_g[j];
          ++gtot[g];
          if (run[j]) { gany[g] = true; any = true; ++gask[g]; ++pcm_asked; }
          ++pcm_total;
        }
        img_copy(x, r_in, 0, nullptr, nullptr, 0);                           // the cached input is always the latest one (cache_manager.py:183)
        if (!any) {                                                           // SD3Transformer.py:219-228: both streams from the block's caches
          img_copy(x, r_out, 1, nullptr, nullptr, 0);
          if (!last) ctx_copy(ctx, r_octx, 1, nullptr, nullptr, 0);
          continue;
        }
        blocks_run |= 1ull << i;
        const float* mi = mod + off_img[i];
        const float* mc = mod + off_ctx[i];
        // ranges of the samples of the active groups (image rows / text rows), uploaded per use
        auto upload = [&](const std::vector<mx::PcRange>& v, int slot) -> const mx::PcRange* {
          mx::PcRange* dst = pcm_dtmp + (size_t)slot * pcm_ncmax();
          if (!ok() || v.empty()) return dst;
          pcm_sent.push_back(v);                // (advisor, round 4: the loop-local table died before the asynchronous copy had to have read it)
          if (hipMemcpyAsync(dst, pcm_sent.back().data(), v.size() * sizeof(mx::PcRange), hipMemcpyHostToDevice, stream) != hipSuccess)
            fail("mmdit patch cache: sending a range table failed");
          return dst;
        };
        std::vector<mx::PcRange> act_img, act_ctx;
        for (int g = 0; g < ng; ++g) if (gany[g]) for (int k = 0; k < gB[g]; ++k) {
          const int bb = gb0[g] + k;
          act_img.push_back(mx::PcRange{pcm_img[bb].row0, gL[g], pcm_img[bb].slot, 0});
          act_ctx.push_back(mx::PcRange{(long long)bb * Lt, Lt, pcm_img[bb].slot, 0});
        }
        const mx::PcRange* d_act_img = upload(act_img, 0);
        const mx::PcRange* d_act_ctx = upload(act_ctx, 1);
        lnmod(x, xin, dual ? x2in : nullptr, mi + d, mi, dual ? mi + 7 * d : nullptr, dual ? mi + 6 * d : nullptr, ntot, MI, d, L, true);
        if (last) lnmod(ctx, cin, nullptr, mc, mc + d, nullptr, nullptr, ntot, MT, d, Lt);
        else lnmod(ctx, cin, nullptr, mc + d, mc, nullptr, nullptr, ntot, MT, d, Lt);
        // the q | k | v projections of both streams: all groups (the reference projects every chunk, attention.py:257-285)
        qkv(xin, b + ".attn.to_qkv", b + ".attn.norm_q.weight", b + ".attn.norm_k.weight", qk_j, vt_j, ldvt_j, MI, d, L, Lj, 0, [&](int g, mx_gemm_seg& q) {
          q.a = xin + r0[g] * d; q.c = qk_j + jr0[g] * 2 * d; q.vt = vt_j + vj0[g]; q.M = gB[g] * gL[g]; q.rows_per_batch = gL[g]; q.ldvt = gldj[g];
          q.c_batch_rows = gLj[g]; q.c_row_off = 0; });
        qkv(cin, b + ".attn.add_qkv", b + ".attn.norm_added_q.weight", b + ".attn.norm_added_k.weight", qk_j, vt_j, ldvt_j, MT, d, Lt, Lj, L, [&](int g, mx_gemm_seg& q) {
          q.a = cin + (long)gb0[g] * Lt * d; q.c = qk_j + jr0[g] * 2 * d; q.vt = vt_j + vj0[g]; q.M = gB[g] * Lt; q.rows_per_batch = Lt; q.ldvt = gldj[g];
          q.c_batch_rows = gLj[g]; q.c_row_off = gL[g]; });
        auto attention_active = [&](bf16_t* qk, bf16_t* vt, bf16_t* o, const long* row0s, const long* vt0s, const int* Ls, const int* lds) {
          mx_attn_problem pr[MX_MAX_SEGS];
          int n = 0;
          for (int g = 0; g < ng; ++g) if (gany[g]) {
            pr[n].q = qk + row0s[g] * 2 * d; pr[n].k = qk + row0s[g] * 2 * d + d; pr[n].vt = vt + vt0s[g]; pr[n].o = o + row0s[g] * d;
            pr[n].vt_batch_stride = (int64_t)d * lds[g]; pr[n].B = gB[g]; pr[n].Lq = Ls[g]; pr[n].Lk = Ls[g]; pr[n].ldvt = lds[g];
            ++n;
          }
          if (ok() && n && mx_attention_prescaled_grouped(stream, pr, n, 2 * d, 2 * d, d, heads)) fail(std::string("attention: ") + mx_last_error());
        };
        attention_active(qk_j, vt_j, o_j, jr0, vj0, gLj, gldj);
        act = gany;
        // to_out of the active groups' image rows, no gate / residual: what attn.output caches (attention.py:407-415)
        linear(o_j, d, b + ".attn.to_out.0", pcm_ti, d, MI, d, d, 0, nullptr, 0, nullptr, 0, L, Lj, 0, [&](int g, mx_gemm_seg& q) {
          q.a = o_j + jr0[g] * d; q.c = pcm_ti + r0[g] * d; q.M = gB[g] * gL[g]; q.rows_per_batch = gL[g]; q.a_batch_rows = gLj[g]; q.a_row_off = 0; });
        act = nullptr;
        if (ok() && mx::launch_pc_range_copy(stream, pcm_ti, r_a, row_i, d, d_act_img, (int)act_img.size(), 0, maxL * d)) fail(mx_last_error());
        img_copy(x, r_a, 1, mi + 2 * d, x, 1);                                 // x += gate_msa * attn.output (fresh or cached)   (transformer.py:344-345)
        if (dual) {
          qkv(x2in, b + ".attn2.to_qkv", b + ".attn2.norm_q.weight", b + ".attn2.norm_k.weight", qk_i, vt_i, ldvt_i, MI, d, L, 0, 0, [&](int g, mx_gemm_seg& q) {
            q.a = x2in + r0[g] * d; q.c = qk_i + r0[g] * 2 * d; q.vt = vt_i + vi0[g]; q.M = gB[g] * gL[g]; q.rows_per_batch = gL[g]; q.ldvt = gldi[g]; });
          attention_active(qk_i, vt_i, o_i, r0, vi0, gL, gldi);
          act = gany;
          linear(o_i, d, b + ".attn2.to_out.0", pcm_ti, d, MI, d, d, 0, nullptr, 0, nullptr, 0, L, 0, 0, [&](int g, mx_gemm_seg& q) {
            q.a = o_i + r0[g] * d; q.c = pcm_ti + r0[g] * d; q.M = gB[g] * gL[g]; q.rows_per_batch = gL[g]; });
          act = nullptr;
          // a group whose asking ratio is <= 1/16 renews its asking chunks only (attention.py:303-325); the others renew every chunk
          std::vector<mx::PcRange> rng2;
          for (int g = 0; g < ng; ++g) if (gany[g]) {
            const bool sparse = gask[g] * 16 <= gtot[g];
            if (!sparse) { for (int k = 0; k < gB[g]; ++k) { const int bb = gb0[g] + k; rng2.push_back(mx::PcRange{pcm_img[bb].row0, gL[g], pcm_img[bb].slot, 0}); } }
            else for (int j = 0; j < NC; ++j) if (pcm_chunk_g[j] == g && run[j]) rng2.push_back(pcm_chunks[j]);
          }
          const mx::PcRange* d_rng2 = upload(rng2, 2);
          if (ok() && mx::launch_pc_range_copy(stream, pcm_ti, r_a2, row_i, d, d_rng2, (int)rng2.size(), 0, maxL * d)) fail(mx_last_error());
          img_copy(x, r_a2, 1, mi + 8 * d, x, 1);
        }
        lnmod(x, xin, nullptr, mi + 4 * d, mi + 3 * d, nullptr, nullptr, ntot, MI, d, L, true);
        linear(xin, d, b + ".ff.net.0.proj", ff, 4 * d, MI, 4 * d, d, MX_EPI_GELU_TANH);
        linear(ff, 4 * d, b + ".ff.net.2", x, d, MI, d, 4 * d, 0, x, d, mi + 5 * d, ntot, L, 0, 0, [&](int g, mx_gemm_seg& q) {
          q.a = ff + r0[g] * 4 * d; q.c = x + r0[g] * d; q.residual = x + r0[g] * d; q.gate = mi + 5 * d + (long)gb0[g] * ntot; q.M = gB[g] * gL[g];
          q.rows_per_batch = gL[g]; });
        if (!last) {
          act = gany;
          linear(o_j, d, b + ".attn.to_add_out", pcm_tc, d, MT, d, d, 0, nullptr, 0, nullptr, 0, Lt, Lj, L, [&](int g, mx_gemm_seg& q) {
            q.a = o_j + jr0[g] * d; q.c = pcm_tc + (long)gb0[g] * Lt * d; q.M = gB[g] * Lt; q.rows_per_batch = Lt; q.a_batch_rows = gLj[g]; q.a_row_off = gL[g]; });
          act = nullptr;
          if (ok() && mx::launch_pc_range_copy(stream, pcm_tc, r_ae, row_c, d, d_act_ctx, (int)act_ctx.size(), 0, (long)Lt * d)) fail(mx_last_error());
          ctx_copy(ctx, r_ae, 1, mc + 2 * d, ctx, 1);                          // ctx += c_gate_msa * attn.encoder_output (fresh or cached)
          lnmod(ctx, cin, nullptr, mc + 4 * d, mc + 3 * d, nullptr, nullptr, ntot, MT, d, Lt);
          linear(cin, d, b + ".ff_context.net.0.proj", ffc, 4 * d, MT, 4 * d, d, MX_EPI_GELU_TANH);
          linear(ffc, 4 * d, b + ".ff_context.net.2", ctx, d, MT, d, 4 * d, 0, ctx, d, mc + 5 * d, ntot, Lt);
        }
        img_copy(x, r_out, 0, nullptr, nullptr, 0);
        if (!last) ctx_copy(ctx, r_octx, 0, nullptr, nullptr, 0);
        continue;
      }
      const bool cached = bc != nullptr && !dry;
      const bool ran = cached ? decide(i) : true;
      if (!ok()) break;
      mute = !ran;
      const float* mi = mod + off_img[i];   // chunks: shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp[, shift2, scale2, gate2]
      const float* mc = mod + off_ctx[i];
      // AdaLN-Zero(-X) on the image stream, AdaLN-Zero / -continuous on the context stream (transformer.py:316-328)
      lnmod(x, xin, dual ? x2in : nullptr, mi + d, mi, dual ? mi + 7 * d : nullptr, dual ? mi + 6 * d : nullptr, ntot, MI, d, L, true);
      if (last) lnmod(ctx, cin, nullptr, mc, mc + d, nullptr, nullptr, ntot, MT, d, Lt);        // continuous: (scale, shift)
      else lnmod(ctx, cin, nullptr, mc + d, mc, nullptr, nullptr, ntot, MT, d, Lt);
      // joint attention (attention.py:256-372): image rows first, then text rows
      // (mixed batch: group g's image rows go to rows [0, L_g) of its samples' joint sequences, the text rows behind them)
      qkv(xin, b + ".attn.to_qkv", b + ".attn.norm_q.weight", b + ".attn.norm_k.weight", qk_j, vt_j, ldvt_j, MI, d, L, Lj, 0, [&](int g, mx_gemm_seg& q) {
        q.a = xin + r0[g] * d; q.c = qk_j + jr0[g] * 2 * d; q.vt = vt_j + vj0[g]; q.M = gB[g] * gL[g]; q.rows_per_batch = gL[g]; q.ldvt = gldj[g];
        q.c_batch_rows = gLj[g]; q.c_row_off = 0; });
      qkv(cin, b + ".attn.add_qkv", b + ".attn.norm_added_q.weight", b + ".attn.norm_added_k.weight", qk_j, vt_j, ldvt_j, MT, d, Lt, Lj, L, [&](int g, mx_gemm_seg& q) {
        q.a = cin + (long)gb0[g] * Lt * d; q.c = qk_j + jr0[g] * 2 * d; q.vt = vt_j + vj0[g]; q.M = gB[g] * Lt; q.rows_per_batch = Lt; q.ldvt = gldj[g];
        q.c_batch_rows = gLj[g]; q.c_row_off = gL[g]; });
      if (is_pp()) {
        gather_kv(qk_j, vt_j, ldvt_j, Lt, ldvt_jt);
        attention_qk(qk_j, k_all, d, vt_all, ldvt_jt, o_j, heads, Lj, Ljt);
      } else if (ng > 1) {
        mx_attn_problem pr[MX_MAX_SEGS];
        for (int g = 0; g < ng; ++g) {
          pr[g].q = qk_j + jr0[g] * 2 * d; pr[g].k = qk_j + jr0[g] * 2 * d + d; pr[g].vt = vt_j + vj0[g]; pr[g].o = o_j + jr0[g] * d;
          pr[g].vt_batch_stride = (int64_t)d * gldj[g]; pr[g].B = gB[g]; pr[g].Lq = gLj[g]; pr[g].Lk = gLj[g]; pr[g].ldvt = gldj[g];
        }
        if (ok() && !quiet() && mx_attention_prescaled_grouped(stream, pr, ng, 2 * d, 2 * d, d, heads)) fail(std::string("attention: ") + mx_last_error());
      } else
      attention(qk_j, d, vt_j, ldvt_j, o_j, heads, Lj);
      // x += gate_msa * to_out(attn[:, :L])                                   (transformer.py:344-345)
      linear(o_j, d, b + ".attn.to_out.0", x, d, MI, d, d, 0, x, d, mi + 2 * d, ntot, L, Lj, 0, [&](int g, mx_gemm_seg& q) {
        q.a = o_j + jr0[g] * d; q.c = x + r0[g] * d; q.residual = x + r0[g] * d; q.gate = mi + 2 * d + (long)gb0[g] * ntot; q.M = gB[g] * gL[g];
        q.rows_per_batch = gL[g]; q.a_batch_rows = gLj[g]; q.a_row_off = 0; });
      if (dual) {                                                             // attn2: image-only self-attention (:347-357)
        qkv(x2in, b + ".attn2.to_qkv", b + ".attn2.norm_q.weight", b + ".attn2.norm_k.weight", qk_i, vt_i, ldvt_i, MI, d, L, 0, 0, [&](int g, mx_gemm_seg& q) {
          q.a = x2in + r0[g] * d; q.c = qk_i + r0[g] * 2 * d; q.vt = vt_i + vi0[g]; q.M = gB[g] * gL[g]; q.rows_per_batch = gL[g]; q.ldvt = gldi[g]; });
        if (is_pp()) {
          gather_kv(qk_i, vt_i, ldvt_i, 0, ldvt_it);
          attention_qk(qk_i, k_all, d, vt_all, ldvt_it, o_i, heads, L, Ltot);
        } else if (ng > 1) {
          mx_attn_problem pr[MX_MAX_SEGS];
          for (int g = 0; g < ng; ++g) {
            pr[g].q = qk_i + r0[g] * 2 * d; pr[g].k = qk_i + r0[g] * 2 * d + d; pr[g].vt = vt_i + vi0[g]; pr[g].o = o_i + r0[g] * d;
            pr[g].vt_batch_stride = (int64_t)d * gldi[g]; pr[g].B = gB[g]; pr[g].Lq = gL[g]; pr[g].Lk = gL[g]; pr[g].ldvt = gldi[g];
          }
          if (ok() && !quiet() && mx_attention_prescaled_grouped(stream, pr, ng, 2 * d, 2 * d, d, heads)) fail(std::string("attention: ") + mx_last_error());
        } else
        attention(qk_i, d, vt_i, ldvt_i, o_i, heads, L);
        linear(o_i, d, b + ".attn2.to_out.0", x, d, MI, d, d, 0, x, d, mi + 8 * d, ntot, L, 0, 0, [&](int g, mx_gemm_seg& q) {
          q.a = o_i + r0[g] * d; q.c = x + r0[g] * d; q.residual = x + r0[g] * d; q.gate = mi + 8 * d + (long)gb0[g] * ntot; q.M = gB[g] * gL[g];
          q.rows_per_batch = gL[g]; });
      }
      // x += gate_mlp * ff(LN(x) * (1 + scale_mlp) + shift_mlp)               (:359-366)
      lnmod(x, xin, nullptr, mi + 4 * d, mi + 3 * d, nullptr, nullptr, ntot, MI, d, L, true);
      linear(xin, d, b + ".ff.net.0.proj", ff, 4 * d, MI, 4 * d, d, MX_EPI_GELU_TANH);
      linear(ff, 4 * d, b + ".ff.net.2", x, d, MI, d, 4 * d, 0, x, d, mi + 5 * d, ntot, L, 0, 0, [&](int g, mx_gemm_seg& q) {
        q.a = ff + r0[g] * 4 * d; q.c = x + r0[g] * d; q.residual = x + r0[g] * d; q.gate = mi + 5 * d + (long)gb0[g] * ntot; q.M = gB[g] * gL[g];
        q.rows_per_batch = gL[g]; });
      if (!last) {                                                            // context stream (:371-386)
        linear(o_j, d, b + ".attn.to_add_out", ctx, d, MT, d, d, 0, ctx, d, mc + 2 * d, ntot, Lt, Lj, L, [&](int g, mx_gemm_seg& q) {
          q.a = o_j + jr0[g] * d; q.c = ctx + (long)gb0[g] * Lt * d; q.residual = ctx + (long)gb0[g] * Lt * d; q.gate = mc + 2 * d + (long)gb0[g] * ntot;
          q.M = gB[g] * Lt; q.rows_per_batch = Lt; q.a_batch_rows = gLj[g]; q.a_row_off = gL[g]; });
        lnmod(ctx, cin, nullptr, mc + 4 * d, mc + 3 * d, nullptr, nullptr, ntot, MT, d, Lt);
        linear(cin, d, b + ".ff_context.net.0.proj", ffc, 4 * d, MT, 4 * d, d, MX_EPI_GELU_TANH);
        linear(ffc, 4 * d, b + ".ff_context.net.2", ctx, d, MT, d, 4 * d, 0, ctx, d, mc + 5 * d, ntot, Lt);
        dump(b + ".context", ctx, (size_t)MT * d);
      }
      dump(b, x, (size_t)MI * d);
      mute = false;
      if (cached && ok()) after(i, ran, last);
    }
    if (pcm) bc_bytes = (size_t)(pcm_top - (dry ? (char*)(uintptr_t)0x1000 : (char*)bc->state));
    // ---- norm_out (AdaLN-continuous) + proj_out + unpatchify (SD3Transformer.py:238-259) ----
    lnmod(x, xin, nullptr, mod + off_out, mod + off_out + d, nullptr, nullptr, ntot, MI, d, L, true);
    const int No = ps * ps * c.out_channels;
    bf16_t* o = alloc<bf16_t>((size_t)MI * No);
    linear(xin, d, "proj_out", o, No, MI, No, d);
    dump("proj_out", o, (size_t)MI * No);
    for (int g = 0; g < ng && ok() && !dry; ++g)
      if (mx::launch_unpatchify(stream, o + r0[g] * No, g_out[g], io_dtype, gB[g], c.out_channels, gH[g], gW[g], ps, No)) fail(mx_last_error());
    if (stage && !dry && ok() && !stage_hit) fail(std::string("unknown stage '") + stage + "'");
    // the range tables of the patch cache were copied from pcm_sent asynchronously: they are released only once the stream has read them
    if (!pcm_sent.empty()) { if (!dry && hipStreamSynchronize(stream) != hipSuccess) fail("mmdit patch cache: final synchronisation failed"); pcm_sent.clear(); }
    return ok();
  }
};

int check_cfg(const mx_mmdit_config* c) {
  MX_CHECK(c != nullptr, "mmdit: null config");
  MX_CHECK(c->num_layers >= 1 && c->num_layers <= 64, "mmdit: num_layers must be 1..64");
  MX_CHECK(c->num_attention_heads >= 1, "mmdit: bad head count (head_dim is fixed at 64)");
  MX_CHECK(c->patch_size >= 1 && (c->patch_size * c->patch_size * c->in_channels) % 64 == 0, "mmdit: patch_size^2 * in_channels must be a multiple of 64");
  MX_CHECK((c->patch_size * c->patch_size * c->out_channels) % 4 == 0, "mmdit: bad out_channels");
  MX_CHECK(c->joint_attention_dim % 64 == 0 && c->pooled_projection_dim % 64 == 0, "mmdit: conditioning widths must be multiples of 64");
  MX_CHECK(c->pos_embed_max_size > 0, "mmdit: pos_embed_max_size required");
  return 0;
}

int forward_impl(mx_mmdit* u, void* stream, const void* latents, int io_dtype, const float* timesteps, const void* ehs,
                 const void* pooled, void* out, int batch, int H, int W, int ctx_len, void* workspace, size_t workspace_bytes,
                 const char* stage, void* stage_out, size_t stage_bytes, bool dry, size_t* peak, bool lookup = false,
                 const mx_pp_comm* comm = nullptr, const mx_pp_stale* stale = nullptr, size_t* state_need = nullptr,
                 const mx_unet_group* groups = nullptr, int n_groups = 0) {
  MX_CHECK(u != nullptr, "mmdit: null handle");
  if (groups) {        // mixed-resolution batch: `batch`, H, W describe the first group; every group is validated here
    MX_CHECK(n_groups >= 1 && n_groups <= MX_MAX_SEGS && comm == nullptr, "mmdit: a mixed batch has 1..MX_MAX_SEGS resolution groups and does not run patch-parallel");
    batch = groups[0].batch; H = groups[0].H; W = groups[0].W; latents = groups[0].latents; out = groups[0].out;
    for (int g = 0; g < n_groups; ++g) {
      MX_CHECK(groups[g].batch > 0 && groups[g].H > 0 && groups[g].W > 0 && groups[g].H % u->cfg.patch_size == 0 && groups[g].W % u->cfg.patch_size == 0 &&
               groups[g].H / u->cfg.patch_size <= u->cfg.pos_embed_max_size && groups[g].W / u->cfg.patch_size <= u->cfg.pos_embed_max_size, "mmdit: bad group shape");
      MX_CHECK(dry || (groups[g].latents && groups[g].out), "mmdit: null group operand");
    }
  }
  MX_CHECK(batch > 0 && H > 0 && W > 0 && ctx_len > 0, "mmdit: bad shape");
  MX_CHECK(H % u->cfg.patch_size == 0 && W % u->cfg.patch_size == 0, "mmdit: H, W must be multiples of patch_size");
  MX_CHECK(H / u->cfg.patch_size <= u->cfg.pos_embed_max_size && W / u->cfg.patch_size <= u->cfg.pos_embed_max_size, "mmdit: latent larger than the positional table");
  const bool pp = comm != nullptr && comm->world > 1;
  if (pp) MX_CHECK(comm->rank >= 0 && comm->rank < comm->world && (dry || comm->all_gather != nullptr), "mmdit pp: bad communicator");
  if (!dry) {
    MX_CHECK(latents && timesteps && ehs && pooled && out && workspace, "mmdit: null operand");
    MX_CHECK(u->blob != nullptr, "mmdit: weights not set");
    MX_CHECK(io_dtype == MX_F32 || io_dtype == MX_F16 || io_dtype == MX_BF16, "mmdit: bad io dtype");
  }
  std::string err;
  size_t plan_peak = 0;
  auto enqueue = [&](hipStream_t s) {
    Plan p;
    p.u = u; p.stream = s; p.Lt = ctx_len;
    p.set_single(batch, H, W, latents, out);
    if (groups) {
      p.ng = n_groups; p.B = 0;
      for (int g = 0; g < n_groups; ++g) {
        p.gB[g] = groups[g].batch; p.gH[g] = groups[g].H; p.gW[g] = groups[g].W; p.gb0[g] = p.B; p.g_lat[g] = groups[g].latents; p.g_out[g] = groups[g].out;
        p.B += groups[g].batch;
      }
    }
    p.dry = dry; p.lookup = lookup; p.stage = stage; p.stage_out = stage_out; p.stage_bytes = stage_bytes;
    p.ar.base = (char*)workspace; p.ar.cap = workspace_bytes; p.ar.top = 0; p.ar.peak = 0; p.ar.dry = dry;
    if (pp) p.px.set(comm, stale);
    if (pp && stale) {      // the state layout (exchanges dealt into chunks, pp_exchange.h) from a host-only recording walk of the same plan
      const std::vector<long> lk = {(long)batch, (long)H, (long)W, (long)ctx_len, (long)comm->world, (long)io_dtype};
      auto it = u->pp_sizes.find(lk);
      if (it == u->pp_sizes.end()) {
        std::vector<size_t> sizes;
        Plan q = p;
        q.dry = true; q.ar.dry = true; q.ar.base = nullptr; q.ar.cap = 0; q.stage = nullptr; q.lookup = false;
        q.px.record = &sizes;
        if (!q.run(nullptr, io_dtype, nullptr, nullptr, nullptr, nullptr)) { err = q.err; return false; }
        it = u->pp_sizes.emplace(lk, std::move(sizes)).first;
      }
      p.px.build_layout(it->second);
    }
    const bool okr = p.run(latents, io_dtype, timesteps, ehs, pooled, out);
    plan_peak = p.ar.peak;
    if (state_need) *state_need = p.px.state_top;
    if (!okr) err = p.err;
    return okr;
  };
  bool okr;
  if (dry || stage || pp) {     // (the all-gather callbacks of a patch-parallel forward cannot be captured)
    okr = enqueue((hipStream_t)stream);
  } else {
    std::vector<uint64_t> key = {(uint64_t)batch, (uint64_t)H, (uint64_t)W, (uint64_t)ctx_len, (uint64_t)io_dtype,
                                 (uint64_t)(uintptr_t)latents, (uint64_t)(uintptr_t)timesteps, (uint64_t)(uintptr_t)ehs,
                                 (uint64_t)(uintptr_t)pooled, (uint64_t)(uintptr_t)out, (uint64_t)(uintptr_t)workspace,
                                 (uint64_t)workspace_bytes, (uint64_t)(uintptr_t)u->blob};
    for (int g = 1; g < n_groups; ++g)
      for (uint64_t v : {(uint64_t)groups[g].batch, (uint64_t)groups[g].H, (uint64_t)groups[g].W, (uint64_t)(uintptr_t)groups[g].latents, (uint64_t)(uintptr_t)groups[g].out})
        key.push_back(v);
    okr = u->graphs.run((hipStream_t)stream, key, enqueue, /*capture_on_miss=*/n_groups <= 1);
  }
  if (peak) *peak = plan_peak;
  if (!okr) { mx::set_error(err); return 1; }
  return 0;
}

}  // namespace

extern "C" mx_mmdit* mx_mmdit_create(const mx_mmdit_config* cfg) {
  if (check_cfg(cfg)) return nullptr;
  mx_mmdit* u = new mx_mmdit();
  u->cfg = *cfg;
  return u;
}
extern "C" void mx_mmdit_destroy(mx_mmdit* u) { delete u; }

extern "C" int mx_mmdit_set_weights(mx_mmdit* u, const void* blob, uint64_t blob_bytes, const mx_weight_entry* table, int n) {
  MX_CHECK(u && blob && table && n > 0, "mmdit_set_weights: bad arguments");
  u->graphs.clear();    // captured graphs hold addresses resolved through the old table
  u->table.clear();
  for (int i = 0; i < n; ++i) {
    MX_CHECK(table[i].name != nullptr, "mmdit_set_weights: null name");
    MX_CHECK(table[i].offset % 16 == 0, "mmdit_set_weights: tensor offsets must be 16-byte aligned");
    MX_CHECK(table[i].offset + table[i].bytes <= blob_bytes, "mmdit_set_weights: entry exceeds blob");
    u->table[table[i].name] = {table[i].offset, table[i].bytes};
  }
  u->blob = (const char*)blob;
  u->blob_bytes = blob_bytes;
  return 0;
}

extern "C" size_t mx_mmdit_workspace_bytes(const mx_mmdit* u, int batch, int H, int W, int ctx_len) {
  if (!u) return 0;
  size_t peak = 0;
  if (forward_impl(const_cast<mx_mmdit*>(u), nullptr, nullptr, MX_BF16, nullptr, nullptr, nullptr, nullptr, batch, H, W, ctx_len,
                   nullptr, 0, nullptr, nullptr, 0, true, &peak))
    return 0;
  return peak + 4096;
}

extern "C" int mx_mmdit_validate(const mx_mmdit* u, int batch, int H, int W, int ctx_len) {
  MX_CHECK(u && u->blob, "mmdit_validate: weights not set");
  return forward_impl(const_cast<mx_mmdit*>(u), nullptr, nullptr, MX_BF16, nullptr, nullptr, nullptr, nullptr, batch, H, W, ctx_len,
                      nullptr, 0, nullptr, nullptr, 0, true, nullptr, true);
}

extern "C" int mx_mmdit_forward(mx_mmdit* u, void* stream, const void* latents, int io_dtype, const float* timesteps,
                                const void* ehs, const void* pooled, void* out, int batch, int H, int W, int ctx_len,
                                void* workspace, size_t workspace_bytes) {
  return forward_impl(u, stream, latents, io_dtype, timesteps, ehs, pooled, out, batch, H, W, ctx_len, workspace, workspace_bytes,
                      nullptr, nullptr, 0, false, nullptr);
}

/* ---- mixed-resolution batch: ONE launch sequence over the requests of every resolution present (see mx_unet_forward_mixed) ---- */
extern "C" size_t mx_mmdit_workspace_bytes_mixed(const mx_mmdit* u, const mx_unet_group* groups, int n_groups, int ctx_len) {
  if (!u || !groups || n_groups < 1 || n_groups > MX_MAX_SEGS) return 0;
  size_t peak = 0;
  if (forward_impl(const_cast<mx_mmdit*>(u), nullptr, nullptr, MX_BF16, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, ctx_len, nullptr, 0, nullptr, nullptr, 0,
                   true, &peak, false, nullptr, nullptr, nullptr, groups, n_groups))
    return 0;
  return peak + 4096;
}

extern "C" int mx_mmdit_forward_mixed(mx_mmdit* u, void* stream, const mx_unet_group* groups, int n_groups, int io_dtype, const float* timesteps,
                                      const void* encoder_hidden_states, const void* pooled_projections, int ctx_len, void* workspace,
                                      size_t workspace_bytes) {
  MX_CHECK(groups != nullptr, "mmdit_forward_mixed: null groups");
  return forward_impl(u, stream, nullptr, io_dtype, timesteps, encoder_hidden_states, pooled_projections, nullptr, 0, 0, 0, ctx_len, workspace,
                      workspace_bytes, nullptr, nullptr, 0, false, nullptr, false, nullptr, nullptr, nullptr, groups, n_groups);
}

/* ---- patch parallelism (distrifuser models/distri_sd3_transformer_pp.py, modules/pp/attn.py:202-277) ---- */
extern "C" size_t mx_mmdit_workspace_bytes_pp(const mx_mmdit* u, int batch, int H_local, int W, int ctx_len, int world) {
  if (!u) return 0;
  size_t peak = 0;
  mx_pp_comm c; c.rank = 0; c.world = world; c.all_gather = nullptr; c.ctx = nullptr;
  if (forward_impl(const_cast<mx_mmdit*>(u), nullptr, nullptr, MX_BF16, nullptr, nullptr, nullptr, nullptr, batch, H_local, W, ctx_len, nullptr, 0,
                   nullptr, nullptr, 0, true, &peak, false, &c))
    return 0;
  return peak + 4096;
}

extern "C" size_t mx_mmdit_pp_state_bytes(const mx_mmdit* u, int batch, int H_local, int W, int ctx_len, int world) {
  if (!u) return 0;
  size_t need = 0;
  mx_pp_comm c; c.rank = 0; c.world = world; c.all_gather = nullptr; c.ctx = nullptr;
  mx_pp_stale st{}; st.mode = MX_PP_WARMUP;
  if (forward_impl(const_cast<mx_mmdit*>(u), nullptr, nullptr, MX_BF16, nullptr, nullptr, nullptr, nullptr, batch, H_local, W, ctx_len, nullptr, 0,
                   nullptr, nullptr, 0, true, nullptr, false, &c, &st, &need))
    return 0;
  return need + 256;
}

extern "C" int mx_mmdit_forward_pp(mx_mmdit* u, void* stream, const void* latents_local, int io_dtype, const float* timesteps, const void* ehs,
                                   const void* pooled, void* out_local, int batch, int H_local, int W, int ctx_len, const mx_pp_comm* comm,
                                   const mx_pp_stale* stale, void* workspace, size_t workspace_bytes) {
  MX_CHECK(comm != nullptr, "mmdit_forward_pp: null communicator");
  if (stale) {
    MX_CHECK(stale->mode == MX_PP_WARMUP || stale->mode == MX_PP_STALE, "mmdit_forward_pp: stale->mode must be MX_PP_WARMUP or MX_PP_STALE");
    MX_CHECK(stale->state != nullptr && ((uintptr_t)stale->state & 255) == 0, "mmdit_forward_pp: state must be 256-byte aligned device memory");
    MX_CHECK(stale->mode != MX_PP_STALE || stale->all_gather_async != nullptr, "mmdit_forward_pp: a stale step needs all_gather_async");
  }
  return forward_impl(u, stream, latents_local, io_dtype, timesteps, ehs, pooled, out_local, batch, H_local, W, ctx_len, workspace, workspace_bytes,
                      nullptr, nullptr, 0, false, nullptr, false, comm, stale);
}

extern "C" int mx_mmdit_pp_comm_plan(const mx_mmdit* u, int batch, int H_local, int W, int ctx_len, const mx_pp_comm* comm) {
  MX_CHECK(u && comm && comm->all_gather, "mmdit_pp_comm_plan: bad arguments");
  return forward_impl(const_cast<mx_mmdit*>(u), nullptr, nullptr, MX_BF16, nullptr, nullptr, nullptr, nullptr, batch, H_local, W, ctx_len, nullptr, 0,
                      nullptr, nullptr, 0, true, nullptr, false, comm);
}

/* ---- the cache at the reference's unit (token chunks) over a mixed batch in ONE launch sequence (include/mxdenoise.h) ---- */
namespace {
int pcm_setup(Plan& p, mx_mmdit* u, const mx_unet_group* groups, int n_groups, int ctx_len, int patch, const mx_block_cache* cache, bool dry) {
  MX_CHECK(u != nullptr, "mmdit: null handle");
  MX_CHECK(groups && n_groups >= 1 && n_groups <= MX_MAX_SEGS, "mmdit_forward_cached_mixed: 1..MX_MAX_SEGS resolution groups");
  MX_CHECK(u->cfg.num_layers <= 64, "mmdit_forward_cached_mixed: at most 64 blocks");
  const int ps = u->cfg.patch_size;
  MX_CHECK(patch > 0 && patch % ps == 0, "mmdit_forward_cached_mixed: the chunk unit (latent pixels per patch edge) must be a multiple of patch_size");
  MX_CHECK(cache && cache->n_slots > 0 && cache->max_h > 0 && cache->max_w > 0 && cache->max_h % patch == 0 && cache->max_w % patch == 0,
           "mmdit_forward_cached_mixed: cache->n_slots, max_h, max_w (multiples of the patch) are required");
  MX_CHECK(ctx_len > 0, "mmdit: bad shape");
  p.u = u; p.Lt = ctx_len; p.dry = dry;
  p.ng = n_groups; p.B = 0;
  p.pcm = true; p.pcm_patch = patch; p.pcm_slots = cache->n_slots; p.pcm_maxh = cache->max_h; p.pcm_maxw = cache->max_w;
  p.pcm_img.clear(); p.pcm_ctx.clear(); p.pcm_chunks.clear(); p.pcm_chunk_b.clear(); p.pcm_chunk_g.clear();
  long long row0 = 0;
  for (int g = 0; g < n_groups; ++g) {
    MX_CHECK(groups[g].batch > 0 && groups[g].H > 0 && groups[g].W > 0 && groups[g].H % patch == 0 && groups[g].W % patch == 0,
             "mmdit_forward_cached_mixed: every group's H, W must be multiples of the patch");
    MX_CHECK(groups[g].H <= cache->max_h && groups[g].W <= cache->max_w, "mmdit_forward_cached_mixed: a group is larger than the state rows (max_h, max_w)");
    MX_CHECK(groups[g].H / ps <= u->cfg.pos_embed_max_size && groups[g].W / ps <= u->cfg.pos_embed_max_size, "mmdit: latent larger than the positional table");
    MX_CHECK(dry || (groups[g].latents && groups[g].out), "mmdit: null group operand");
    p.gB[g] = groups[g].batch; p.gH[g] = groups[g].H; p.gW[g] = groups[g].W; p.gb0[g] = p.B; p.g_lat[g] = groups[g].latents; p.g_out[g] = groups[g].out;
    const int L = (groups[g].H / ps) * (groups[g].W / ps), nc = (groups[g].H / patch) * (groups[g].W / patch);
    MX_CHECK(L % nc == 0, "mmdit_forward_cached_mixed: the tokens of a latent must split into equal chunks");
    for (int k = 0; k < groups[g].batch; ++k) {
      const int b = p.B + k;
      const int slot = (!dry && cache->slots) ? cache->slots[b] : b;
      p.pcm_img.push_back(mx::PcSample{row0, L, 1, slot, 1});
      p.pcm_ctx.push_back(mx::PcSample{(long long)b * ctx_len, ctx_len, 1, slot, 1});
      for (int j = 0; j < nc; ++j) {
        p.pcm_chunks.push_back(mx::PcRange{row0 + (long long)j * (L / nc), L / nc, slot, j * (L / nc)});
        p.pcm_chunk_b.push_back(b); p.pcm_chunk_g.push_back(g);
      }
      row0 += L;
    }
    p.B += groups[g].batch;
  }
  p.H = groups[0].H; p.W = groups[0].W;
  p.pcm_nc = (int)p.pcm_chunks.size();
  MX_CHECK(p.B <= cache->n_slots, "mmdit_forward_cached_mixed: more samples than state rows (n_slots)");
  return 0;
}
size_t pcm_dry(const mx_mmdit* u, const mx_unet_group* groups, int n_groups, int ctx_len, int patch, const mx_block_cache* sizing, bool want_state) {
  Plan p;
  if (pcm_setup(p, const_cast<mx_mmdit*>(u), groups, n_groups, ctx_len, patch, sizing, true)) return 0;
  p.stream = nullptr; p.ar.base = nullptr; p.ar.cap = 0; p.ar.top = 0; p.ar.peak = 0; p.ar.dry = true;
  p.bc = const_cast<mx_block_cache*>(sizing);
  if (!p.run(nullptr, MX_BF16, nullptr, nullptr, nullptr, nullptr)) { mx::set_error(p.err); return 0; }
  return want_state ? p.bc_bytes + 256 : p.ar.peak + 256;
}
}  // namespace

extern "C" size_t mx_mmdit_patch_cache_bytes(const mx_mmdit* u, int n_slots, int max_h, int max_w, int patch, int ctx_len) {
  if (!u || n_slots <= 0 || max_h <= 0 || max_w <= 0 || patch <= 0 || ctx_len <= 0) { mx::set_error("mmdit_patch_cache_bytes: bad arguments"); return 0; }
  mx_block_cache sizing{};
  sizing.n_slots = n_slots; sizing.max_h = max_h; sizing.max_w = max_w;
  mx_unet_group g{nullptr, nullptr, n_slots, max_h, max_w};
  return pcm_dry(u, &g, 1, ctx_len, patch, &sizing, true);
}

extern "C" size_t mx_mmdit_workspace_bytes_cached_mixed(const mx_mmdit* u, const mx_unet_group* groups, int n_groups, int ctx_len, int patch) {
  mx_block_cache sizing{};
  if (groups) for (int g = 0; g < n_groups && g < MX_MAX_SEGS; ++g) {
    sizing.n_slots += groups[g].batch; sizing.max_h = std::max(sizing.max_h, groups[g].H); sizing.max_w = std::max(sizing.max_w, groups[g].W);
  }
  return pcm_dry(u, groups, n_groups, ctx_len, patch, &sizing, false);
}

extern "C" int mx_mmdit_forward_cached_mixed(mx_mmdit* u, void* stream, const mx_unet_group* groups, int n_groups, int io_dtype, const float* timesteps,
                                             const void* ehs, const void* pooled, int ctx_len, int patch, void* workspace, size_t workspace_bytes,
                                             mx_block_cache* cache) {
  MX_CHECK(cache && cache->predict && cache->state && cache->slots && cache->slot_valid, "mmdit_forward_cached_mixed: cache with predict, state, slots and slot_valid is required");
  MX_CHECK(((uintptr_t)cache->state & 255) == 0, "mmdit_forward_cached_mixed: cache->state must be 256-byte aligned");
  Plan p;
  if (pcm_setup(p, u, groups, n_groups, ctx_len, patch, cache, false)) return 1;
  MX_CHECK(timesteps && ehs && pooled && workspace, "mmdit: null operand");
  MX_CHECK(u->blob != nullptr, "mmdit: weights not set");
  MX_CHECK(io_dtype == MX_F32 || io_dtype == MX_F16 || io_dtype == MX_BF16, "mmdit: bad io dtype");
  const int B = p.B;
  std::vector<char> seen(cache->n_slots, 0);
  p.bc_valid.assign(B, 0);
  p.bc_all_valid = true; p.bc_any_valid = false;
  for (int b = 0; b < B; ++b) {
    MX_CHECK(cache->slots[b] >= 0 && cache->slots[b] < cache->n_slots && !seen[cache->slots[b]], "mmdit_forward_cached_mixed: slots must be distinct and inside [0, n_slots)");
    seen[cache->slots[b]] = 1;
    p.bc_valid[b] = cache->slot_valid[b] ? 1 : 0;
    p.bc_all_valid = p.bc_all_valid && p.bc_valid[b]; p.bc_any_valid = p.bc_any_valid || p.bc_valid[b];
  }
  p.stream = (hipStream_t)stream;
  p.ar.base = (char*)workspace; p.ar.cap = workspace_bytes; p.ar.top = 0; p.ar.peak = 0; p.ar.dry = false;
  p.bc = cache;
  MX_CHECK(p.pcm_head_bytes() <= cache->state_bytes, "mmdit_forward_cached_mixed: state buffer too small");
  const size_t ncm = p.pcm_ncmax();
  char* hp = (char*)cache->state;
  p.pcm_dpart = (double*)hp; hp += ncm * 64 * sizeof(double);
  p.pcm_dimg = (mx::PcSample*)hp; hp += (size_t)p.pcm_slots * sizeof(mx::PcSample);
  p.pcm_dctx = (mx::PcSample*)hp; hp += (size_t)p.pcm_slots * sizeof(mx::PcSample);
  p.pcm_dchunks = (mx::PcRange*)hp; hp += ncm * sizeof(mx::PcRange);
  p.pcm_dtmp = (mx::PcRange*)hp;
  p.h_timesteps.resize(B);
  if (hipMemcpyAsync(p.pcm_dimg, p.pcm_img.data(), (size_t)B * sizeof(mx::PcSample), hipMemcpyHostToDevice, p.stream) != hipSuccess ||
      hipMemcpyAsync(p.pcm_dctx, p.pcm_ctx.data(), (size_t)B * sizeof(mx::PcSample), hipMemcpyHostToDevice, p.stream) != hipSuccess ||
      hipMemcpyAsync(p.pcm_dchunks, p.pcm_chunks.data(), (size_t)p.pcm_nc * sizeof(mx::PcRange), hipMemcpyHostToDevice, p.stream) != hipSuccess ||
      hipMemcpyAsync(p.h_timesteps.data(), timesteps, (size_t)B * sizeof(float), hipMemcpyDeviceToHost, p.stream) != hipSuccess ||
      hipStreamSynchronize(p.stream) != hipSuccess) {
    mx::set_error("mmdit_forward_cached_mixed: moving the tables failed");
    return 1;
  }
  const bool okr = p.run(groups[0].latents, io_dtype, timesteps, ehs, pooled, groups[0].out);
  cache->blocks_run = (unsigned)(p.blocks_run & 0xffffffffull); cache->blocks_run_hi = (unsigned)(p.blocks_run >> 32);
  cache->patches_asked = p.pcm_asked; cache->patches_total = p.pcm_total;
  if (!okr) { mx::set_error(p.err); return 1; }
  return 0;
}

/* ---- block-skip cache (include/mxdenoise.h; SD3Transformer.py:151-228 with cache_manager.py:163-191) ---- */
extern "C" size_t mx_mmdit_block_cache_bytes(const mx_mmdit* u, int batch, int H, int W, int ctx_len) {
  if (!u || batch <= 0 || H <= 0 || W <= 0 || ctx_len <= 0 || H % u->cfg.patch_size || W % u->cfg.patch_size) return 0;
  Plan p;
  mx_block_cache sizing{};
  p.u = const_cast<mx_mmdit*>(u); p.stream = nullptr; p.set_single(batch, H, W, nullptr, nullptr); p.Lt = ctx_len;
  p.dry = true; p.ar.base = nullptr; p.ar.cap = 0; p.ar.top = 0; p.ar.peak = 0; p.ar.dry = true;
  p.bc = &sizing; p.bc_rows = batch;
  if (!p.run(nullptr, MX_BF16, nullptr, nullptr, nullptr, nullptr)) { mx::set_error(p.err); return 0; }
  return p.bc_bytes;
}

extern "C" int mx_mmdit_forward_cached(mx_mmdit* u, void* stream, const void* latents, int io_dtype, const float* timesteps, const void* ehs,
                                       const void* pooled, void* out, int batch, int H, int W, int ctx_len, void* workspace,
                                       size_t workspace_bytes, mx_block_cache* cache) {
  MX_CHECK(u != nullptr, "mmdit: null handle");
  MX_CHECK(cache && cache->predict && cache->state, "mmdit_forward_cached: cache, cache->predict and cache->state are required");
  MX_CHECK(u->cfg.num_layers <= 64, "mmdit_forward_cached: at most 64 blocks");
  MX_CHECK(batch > 0 && H > 0 && W > 0 && ctx_len > 0, "mmdit: bad shape");
  MX_CHECK(H % u->cfg.patch_size == 0 && W % u->cfg.patch_size == 0, "mmdit: H, W must be multiples of patch_size");
  MX_CHECK(H / u->cfg.patch_size <= u->cfg.pos_embed_max_size && W / u->cfg.patch_size <= u->cfg.pos_embed_max_size, "mmdit: latent larger than the positional table");
  MX_CHECK(latents && timesteps && ehs && pooled && out && workspace, "mmdit: null operand");
  MX_CHECK(u->blob != nullptr, "mmdit: weights not set");
  MX_CHECK(io_dtype == MX_F32 || io_dtype == MX_F16 || io_dtype == MX_BF16, "mmdit: bad io dtype");
  MX_CHECK(((uintptr_t)cache->state & 255) == 0, "mmdit_forward_cached: cache->state must be 256-byte aligned");
  Plan p;
  p.bc_valid.assign(batch, 0);
  if (cache->slots) {                      // one state row per request (see mx_block_cache)
    MX_CHECK(cache->slot_valid != nullptr && cache->n_slots >= batch, "mmdit_forward_cached: slots need slot_valid and n_slots >= batch");
    std::vector<char> seen(cache->n_slots, 0);
    for (int b = 0; b < batch; ++b) {
      MX_CHECK(cache->slots[b] >= 0 && cache->slots[b] < cache->n_slots && !seen[cache->slots[b]], "mmdit_forward_cached: slots must be distinct and inside [0, n_slots)");
      seen[cache->slots[b]] = 1;
      p.bc_valid[b] = cache->slot_valid[b] ? 1 : 0;
    }
    p.bc_rows = cache->n_slots;
  } else {
    cache->cached_valid = cache->cached_valid && cache->cached_key == cache->batch_key && cache->cached_batch == batch && cache->cached_h == H &&
                          cache->cached_w == W;
    p.bc_valid.assign(batch, cache->cached_valid ? 1 : 0);
    p.bc_rows = batch;
  }
  p.bc_all_valid = true; p.bc_any_valid = false;
  for (int b = 0; b < batch; ++b) { p.bc_all_valid = p.bc_all_valid && p.bc_valid[b]; p.bc_any_valid = p.bc_any_valid || p.bc_valid[b]; }
  p.u = u; p.stream = (hipStream_t)stream; p.set_single(batch, H, W, latents, out); p.Lt = ctx_len;
  p.dry = false;
  p.ar.base = (char*)workspace; p.ar.cap = workspace_bytes; p.ar.top = 0; p.ar.peak = 0; p.ar.dry = false;
  p.bc = cache;
  if (cache->slots) {
    MX_CHECK(Plan::bc_scratch_bytes(p.bc_rows) <= cache->state_bytes, "mmdit_forward_cached: state buffer too small");
    int* dslot = (int*)((char*)cache->state + (size_t)p.bc_rows * 64 * sizeof(double));
    if (hipMemcpyAsync(dslot, cache->slots, (size_t)batch * sizeof(int), hipMemcpyHostToDevice, p.stream) != hipSuccess) {
      mx::set_error("mmdit_forward_cached: sending the slot table failed");
      return 1;
    }
    p.bc_dslot = dslot;
  }
  p.h_timesteps.resize(batch);
  if (hipMemcpyAsync(p.h_timesteps.data(), timesteps, (size_t)batch * sizeof(float), hipMemcpyDeviceToHost, p.stream) != hipSuccess ||
      hipStreamSynchronize(p.stream) != hipSuccess) {
    cache->cached_valid = 0;
    mx::set_error("mmdit_forward_cached: reading the timesteps failed");
    return 1;
  }
  const bool okr = p.run(latents, io_dtype, timesteps, ehs, pooled, out);
  cache->blocks_run = (unsigned)p.blocks_run;
  cache->blocks_run_hi = (unsigned)(p.blocks_run >> 32);
  if (!okr) { cache->cached_valid = 0; mx::set_error(p.err); return 1; }
  cache->cached_valid = 1; cache->cached_key = cache->batch_key; cache->cached_batch = batch; cache->cached_h = H; cache->cached_w = W;
  return 0;
}

extern "C" int mx_mmdit_forward_trace(mx_mmdit* u, void* stream, const void* latents, int io_dtype, const float* timesteps,
                                      const void* ehs, const void* pooled, void* out, int batch, int H, int W, int ctx_len,
                                      void* workspace, size_t workspace_bytes, const char* stage, void* stage_out,
                                      size_t stage_out_bytes) {
  MX_CHECK(stage && stage_out, "mmdit_forward_trace: stage and stage_out required");
  return forward_impl(u, stream, latents, io_dtype, timesteps, ehs, pooled, out, batch, H, W, ctx_len, workspace, workspace_bytes,
                      stage, stage_out, stage_out_bytes, false, nullptr);
}
