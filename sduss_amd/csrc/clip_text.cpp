// Step plan of a CLIP TEXT ENCODER (transformers CLIPTextModel / CLIPTextModelWithProjection), the step BEFORE the denoising loop: the
// reference's prepare_inference calls diffusers' encode_prompt (pipeline_stable_diffusion_xl_esymred.py:118-140), which runs the two
// text encoders of SDXL on the 77 token ids of each prompt and takes hidden_states[-2] of both plus the pooled, projected embedding of
// the second.  Same building blocks as the denoisers: LayerNorm, fused q / k / v GEMM with the V^T
// epilogue, flash attention (64-wide heads, here with the causal mask), GEMM epilogues (bias, residual, quick-GELU / GELU).
//   encode(ids):  x = tok[ids] + pos;  per layer: x += out_proj(attn_causal(qkv(LN1 x)));  x += fc2(act(fc1(LN2 x)));
//                 hidden = hidden_states[hidden_layer] ;  pooled = text_projection(LN_final(x_last)[eos])
#include <hip/hip_runtime.h>

#include <cstring>
#include <string>
#include <unordered_map>

#include "../../include/mxdenoise.h"
#include "common.h"

namespace mx {
int launch_clip_embed(hipStream_t s, const int* ids, const bf16_t* tok, const bf16_t* pos, bf16_t* out, int rows, int L, int H, int vocab);
int launch_clip_pool(hipStream_t s, const int* ids, const bf16_t* x, bf16_t* out, int B, int L, int H, int eos_id);
}  // namespace mx

using mx::bf16_t;

struct mx_clip {
  mx_clip_config cfg;
  const char* blob = nullptr;
  uint64_t blob_bytes = 0;
  std::unordered_map<std::string, std::pair<uint64_t, uint64_t>> table;
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
  mx_clip* u; hipStream_t stream; Arena ar; int B; bool dry, lookup; std::string err;
  bool ok() const { return err.empty(); }
  bool fail(const std::string& m) { if (err.empty()) err = m; return false; }
  const void* w(const std::string& name, size_t bytes) {
    if (dry && !lookup) return (const void*)(uintptr_t)0x1000;
    auto it = u->table.find(name);
    if (it == u->table.end()) { fail("missing weight '" + name + "'"); return nullptr; }
    if (it->second.second != bytes) { fail("weight '" + name + "' has " + std::to_string(it->second.second) + " bytes, expected " + std::to_string(bytes)); return nullptr; }
    return u->blob + it->second.first;
  }
  const bf16_t* wb(const std::string& n, size_t e) { return (const bf16_t*)w(n, e * 2); }
  const float* wf(const std::string& n, size_t e) { return (const float*)w(n, e * 4); }
  template <typename T> T* alloc(size_t elems) { T* p = (T*)ar.alloc(elems * sizeof(T)); if (!p) fail("workspace too small"); return p; }
  bool gemm(mx_gemm_desc& d) {
    if (!ok()) return false;
    if (dry) return true;
    if (mx_gemm(stream, &d)) return fail(std::string("gemm: ") + mx_last_error());
    return true;
  }
  bool linear(const bf16_t* a, const std::string& stem, void* c, int M, int N, int K, const void* residual = nullptr, int flags = 0, bool bias = true) {
    mx_gemm_desc d; std::memset(&d, 0, sizeof(d));
    d.a = a; d.lda = K; d.w = wb(stem + ".weight", (size_t)N * K); d.bias = bias ? wf(stem + ".bias", N) : nullptr;
    d.c = c; d.ldc = N; d.M = M; d.N = N; d.K = K; d.residual = residual; d.ldr = N; d.flags = flags;
    return gemm(d);
  }
  bool layernorm(const bf16_t* x, bf16_t* y, const std::string& stem, int M, int H) {
    const float* g = wf(stem + ".weight", H); const float* b = wf(stem + ".bias", H);
    if (ok() && !dry && mx_layernorm(stream, x, y, g, b, M, H, u->cfg.layer_norm_eps)) fail(std::string("layernorm: ") + mx_last_error());
    return ok();
  }

  bool run(const int* ids, void* hidden_out, void* pooled_out) {
    const mx_clip_config& c = u->cfg;
    const int L = c.max_position_embeddings, H = c.hidden_size, heads = c.num_attention_heads;
    const int M = B * L;
    const int ldvt = MX_VT_LD(L);
    const int keep = c.num_hidden_layers + 1 + c.hidden_layer; // index into transformers' hidden_states (embeddings, layer 1, ..., layer N): -2 = output of layer N - 1
    bf16_t* x = alloc<bf16_t>((size_t)M * H);
    bf16_t* n = alloc<bf16_t>((size_t)M * H);
    bf16_t* qk = alloc<bf16_t>((size_t)M * 2 * H);
    bf16_t* vt = alloc<bf16_t>((size_t)B * H * ldvt);
    bf16_t* ao = alloc<bf16_t>((size_t)M * H);
    bf16_t* ff = alloc<bf16_t>((size_t)M * c.intermediate_size);
    const std::string tm = "text_model.";
    const bf16_t* tok = wb(tm + "embeddings.token_embedding.weight", (size_t)c.vocab_size * H);
    const bf16_t* pos = wb(tm + "embeddings.position_embedding.weight", (size_t)L * H);
    if (ok() && !dry && mx::launch_clip_embed(stream, ids, tok, pos, x, M, L, H, c.vocab_size)) fail(mx_last_error());
    const int act = c.hidden_act == 1 ? MX_EPI_GELU : MX_EPI_QUICK_GELU;
    const bool need_last = pooled_out != nullptr;
    for (int l = 0; l < c.num_hidden_layers && ok(); ++l) {
      if (l == keep && hidden_out && ok() && !dry &&
          hipMemcpyAsync(hidden_out, x, (size_t)M * H * 2, hipMemcpyDeviceToDevice, stream) != hipSuccess) fail("hidden copy failed");
      if (l >= keep && !need_last) break;                      // nothing after hidden_states[hidden_layer] is asked for
      const std::string p = tm + "encoder.layers." + std::to_string(l);
      layernorm(x, n, p + ".layer_norm1", M, H);
      {
        mx_gemm_desc d; std::memset(&d, 0, sizeof(d));         // q | k | v fused; q scaled for the prescaled attention, V written transposed
        d.a = n; d.lda = H; d.w = wb(p + ".self_attn.qkv_proj.weight", (size_t)3 * H * H); d.bias = wf(p + ".self_attn.qkv_proj.bias", 3 * H);
        d.c = qk; d.ldc = 2 * H; d.M = M; d.N = 3 * H; d.K = H; d.flags = MX_EPI_QKV; d.seg = H; d.period = 3; d.vt = vt; d.ldvt = ldvt;
        d.rows_per_batch = L; d.out_scale = MX_ATTN_QSCALE(0.125f);
        gemm(d);
      }
      if (ok() && !dry && mx_attention_prescaled_causal(stream, qk, 2 * H, qk + H, 2 * H, vt, ldvt, (int64_t)H * ldvt, ao, H, B, heads, L))
        fail(std::string("attention: ") + mx_last_error());
      linear(ao, p + ".self_attn.out_proj", x, M, H, H, x);
      layernorm(x, n, p + ".layer_norm2", M, H);
      linear(n, p + ".mlp.fc1", ff, M, c.intermediate_size, H, nullptr, act);
      linear(ff, p + ".mlp.fc2", x, M, H, c.intermediate_size, x);
    }
    if (keep == c.num_hidden_layers && hidden_out && ok() && !dry &&
        hipMemcpyAsync(hidden_out, x, (size_t)M * H * 2, hipMemcpyDeviceToDevice, stream) != hipSuccess) fail("hidden copy failed");
    if (need_last) {
      if (c.projection_dim <= 0) return fail("clip: pooled output asked of an encoder without text_projection");
      layernorm(x, n, tm + "final_layer_norm", M, H);
      bf16_t* pooled = alloc<bf16_t>((size_t)B * H);
      if (ok() && !dry && mx::launch_clip_pool(stream, ids, n, pooled, B, L, H, c.eos_token_id == 2 ? -1 : c.eos_token_id)) fail(mx_last_error());
      mx_gemm_desc d; std::memset(&d, 0, sizeof(d));           // text_projection: no bias, fp32 out
      d.a = pooled; d.lda = H; d.w = wb("text_projection.weight", (size_t)c.projection_dim * H); d.c = pooled_out; d.ldc = c.projection_dim;
      d.M = B; d.N = c.projection_dim; d.K = H; d.flags = MX_EPI_OUT_F32;
      gemm(d);
    }
    return ok();
  }
};

int run_impl(mx_clip* u, void* stream, const int* ids, void* hidden_out, void* pooled_out, int batch, void* ws, size_t ws_bytes, bool dry, bool lookup,
             size_t* peak, bool want_pooled) {
  MX_CHECK(u != nullptr, "clip: null handle");
  MX_CHECK(batch > 0, "clip: bad batch");
  if (!dry) MX_CHECK(ids && ws && u->blob && (hidden_out || pooled_out), "clip: null operand or weights not set");
  Plan p;
  p.u = u; p.stream = (hipStream_t)stream; p.B = batch; p.dry = dry; p.lookup = lookup;
  p.ar.base = (char*)ws; p.ar.cap = ws_bytes; p.ar.top = 0; p.ar.peak = 0; p.ar.dry = dry;
  const bool okr = p.run(ids, dry ? (void*)16 : hidden_out, dry ? (want_pooled ? (void*)16 : nullptr) : pooled_out);
  if (peak) *peak = p.ar.peak;
  if (!okr) { mx::set_error(p.err); return 1; }
  return 0;
}
}  // namespace

extern "C" mx_clip* mx_clip_create(const mx_clip_config* c) {
  if (!c || c->hidden_size <= 0 || c->num_attention_heads <= 0 || c->hidden_size != 64 * c->num_attention_heads || c->hidden_size % 64 != 0 ||
      c->intermediate_size % 64 != 0 || c->num_hidden_layers < 1 || c->max_position_embeddings < 1 || c->max_position_embeddings > 4096 ||
      c->vocab_size < 1 || c->hidden_layer > -1 || c->num_hidden_layers + 1 + c->hidden_layer < 0 || c->hidden_act < 0 || c->hidden_act > 1 ||
      (c->projection_dim > 0 && c->projection_dim % 4 != 0)) {
    mx::set_error("clip: bad config (heads of 64, hidden_act 0 = quick_gelu | 1 = gelu, hidden_layer in [-(layers + 1), -1])");
    return nullptr;
  }
  mx_clip* u = new mx_clip();
  u->cfg = *c;
  return u;
}
extern "C" void mx_clip_destroy(mx_clip* u) { delete u; }
extern "C" int mx_clip_set_weights(mx_clip* u, const void* blob, uint64_t blob_bytes, const mx_weight_entry* table, int n) {
  MX_CHECK(u && blob && table && n > 0, "clip_set_weights: bad arguments");
  u->table.clear();
  for (int i = 0; i < n; ++i) {
    MX_CHECK(table[i].name != nullptr && table[i].offset % 16 == 0 && table[i].offset + table[i].bytes <= blob_bytes, "clip_set_weights: bad entry");
    u->table[table[i].name] = {table[i].offset, table[i].bytes};
  }
  u->blob = (const char*)blob; u->blob_bytes = blob_bytes;
  return 0;
}
extern "C" size_t mx_clip_workspace_bytes(const mx_clip* u, int batch) {
  size_t peak = 0;
  if (run_impl(const_cast<mx_clip*>(u), nullptr, nullptr, nullptr, nullptr, batch, nullptr, 0, true, false, &peak, u && u->cfg.projection_dim > 0)) return 0;
  return peak + 4096;
}
extern "C" int mx_clip_validate(const mx_clip* u, int batch) {
  MX_CHECK(u && u->blob, "clip_validate: weights not set");
  return run_impl(const_cast<mx_clip*>(u), nullptr, nullptr, nullptr, nullptr, batch, nullptr, 0, true, true, nullptr, u->cfg.projection_dim > 0);
}
extern "C" int mx_clip_encode(mx_clip* u, void* stream, const int32_t* ids, void* hidden_out, void* pooled_out, int batch, void* workspace,
                              size_t workspace_bytes) {
  return run_impl(u, stream, ids, hidden_out, pooled_out, batch, workspace, workspace_bytes, false, false, nullptr, pooled_out != nullptr);
}
