// Step plan of the T5 v1.1 ENCODER (transformers T5EncoderModel), SD3's third text encoder: prepare_inference of the SD3 pipeline calls diffusers'
// encode_prompt, whose _get_t5_prompt_embeds runs text_encoder_3 on 256 token ids per prompt WITHOUT an attention mask and keeps the last
// hidden state [n, 256, 4096] (pipeline_stable_diffusion_3_esymred.py, prepare_inference).
//   h = embed[ids];  per block:  h += Wo attn(q, k, v) of RMSNorm(h) -- no 1/sqrt(d) scaling, softmax(q k^T + position_bias[head]) with the
//   bucketed relative position bias of block 0 shared by all blocks;  h += wo (gelu_new(wi_0 n) * (wi_1 n)), n = RMSNorm(h);  out = RMSNorm(h).
// Same building blocks as the denoisers (64-wide heads): fused q / k / v GEMM with the V^T epilogue, the flash kernel with an additive bias
// (mx_attention_prescaled_bias; bias and q both carry log2(e)), the GEGLU epilogue in its tanh form, residuals in the GEMM epilogues.
#include <hip/hip_runtime.h>

#include <cstring>
#include <string>
#include <unordered_map>

#include "../../include/mxdenoise.h"
#include "common.h"

namespace mx {
int launch_clip_embed(hipStream_t s, const int* ids, const bf16_t* tok, const bf16_t* pos, bf16_t* out, int rows, int L, int H, int vocab);
}  // namespace mx

using mx::bf16_t;

struct mx_t5 {
  mx_t5_config cfg;
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
  mx_t5* u; hipStream_t stream; Arena ar; int B, L; bool dry, lookup; std::string err;
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
  bool linear(const bf16_t* a, const std::string& name, void* c, int M, int N, int K, int ldc, const void* residual = nullptr, int flags = 0) {
    mx_gemm_desc d; std::memset(&d, 0, sizeof(d));
    d.a = a; d.lda = K; d.w = wb(name, (size_t)N * K); d.c = c; d.ldc = ldc; d.M = M; d.N = N; d.K = K; d.residual = residual; d.ldr = ldc; d.flags = flags;
    return gemm(d);
  }
  bool rmsnorm(const bf16_t* x, bf16_t* y, const std::string& name, int M, int C) {
    const float* g = wf(name, C);
    if (ok() && !dry && mx_rmsnorm(stream, x, y, g, M, C, u->cfg.layer_norm_epsilon)) fail(std::string("rmsnorm: ") + mx_last_error());
    return ok();
  }

  bool run(const int* ids, void* out) {
    const mx_t5_config& c = u->cfg;
    const int D = c.d_model, heads = c.num_heads, inner = heads * 64, F = c.d_ff;
    const int M = B * L;
    const int ldvt = MX_VT_LD(L);
    const int ldb = (L + 63) / 64 * 64;
    bf16_t* h = alloc<bf16_t>((size_t)M * D);
    bf16_t* n = alloc<bf16_t>((size_t)M * D);
    bf16_t* qk = alloc<bf16_t>((size_t)M * 2 * inner);
    bf16_t* vt = alloc<bf16_t>((size_t)B * inner * ldvt);
    bf16_t* ao = alloc<bf16_t>((size_t)M * inner);
    bf16_t* ff = alloc<bf16_t>((size_t)M * F);
    const bf16_t* tok = wb("encoder.embed_tokens.weight", (size_t)c.vocab_size * D);
    // [heads][L][ldb] fp32, times log2(e): computed from block 0's relative_attention_bias at pack time for THIS sequence length
    const float* bias = wf("encoder.position_bias." + std::to_string(L), (size_t)heads * L * ldb);
    if (ok() && !dry && mx::launch_clip_embed(stream, ids, tok, nullptr, h, M, L, D, c.vocab_size)) fail(mx_last_error());
    for (int l = 0; l < c.num_layers && ok(); ++l) {
      const std::string p = "encoder.block." + std::to_string(l);
      rmsnorm(h, n, p + ".layer.0.layer_norm.weight", M, D);
      {
        mx_gemm_desc d; std::memset(&d, 0, sizeof(d));         // q | k | v fused, no bias; q carries log2(e) only (T5 does not scale by 1/sqrt(d))
        d.a = n; d.lda = D; d.w = wb(p + ".layer.0.SelfAttention.qkv.weight", (size_t)3 * inner * D);
        d.c = qk; d.ldc = 2 * inner; d.M = M; d.N = 3 * inner; d.K = D; d.flags = MX_EPI_QKV; d.seg = inner; d.period = 3; d.vt = vt; d.ldvt = ldvt;
        d.rows_per_batch = L; d.out_scale = MX_ATTN_QSCALE(1.0f);
        gemm(d);
      }
      if (ok() && !dry && mx_attention_prescaled_bias(stream, qk, 2 * inner, qk + inner, 2 * inner, vt, ldvt, (int64_t)inner * ldvt, ao, inner, B, heads, L, L, bias, ldb))
        fail(std::string("attention: ") + mx_last_error());
      linear(ao, p + ".layer.0.SelfAttention.o.weight", h, M, D, inner, D, h);
      rmsnorm(h, n, p + ".layer.1.layer_norm.weight", M, D);
      // wi_1 (linear) | wi_0 (gate) interleaved for the GEGLU epilogue; the gate takes gelu_new = the tanh form
      linear(n, p + ".layer.1.DenseReluDense.wi.weight", ff, M, 2 * F, D, F, nullptr, MX_EPI_GEGLU | MX_EPI_GEGLU_TANH);
      linear(ff, p + ".layer.1.DenseReluDense.wo.weight", h, M, D, F, D, h);
    }
    const float* gf = wf("encoder.final_layer_norm.weight", D);
    if (ok() && !dry && mx_rmsnorm(stream, h, out, gf, M, D, c.layer_norm_epsilon)) fail(std::string("rmsnorm: ") + mx_last_error());
    return ok();
  }
};

int run_impl(mx_t5* u, void* stream, const int* ids, void* out, int batch, int L, void* ws, size_t ws_bytes, bool dry, bool lookup, size_t* peak) {
  MX_CHECK(u != nullptr, "t5: null handle");
  MX_CHECK(batch > 0 && L > 0 && L % 8 == 0 && L <= 4096, "t5: bad batch / sequence length (a multiple of 8, <= 4096)");
  if (!dry) MX_CHECK(ids && out && ws && u->blob, "t5: null operand or weights not set");
  Plan p;
  p.u = u; p.stream = (hipStream_t)stream; p.B = batch; p.L = L; p.dry = dry; p.lookup = lookup;
  p.ar.base = (char*)ws; p.ar.cap = ws_bytes; p.ar.top = 0; p.ar.peak = 0; p.ar.dry = dry;
  const bool okr = p.run(ids, out);
  if (peak) *peak = p.ar.peak;
  if (!okr) { mx::set_error(p.err); return 1; }
  return 0;
}
}  // namespace

extern "C" mx_t5* mx_t5_create(const mx_t5_config* c) {
  if (!c || c->d_model <= 0 || c->d_model % 64 != 0 || c->d_model > 4096 || c->num_heads <= 0 || c->d_ff <= 0 || c->d_ff % 64 != 0 || c->num_layers < 1 ||
      c->vocab_size < 1) {
    mx::set_error("t5: bad config (d_model a multiple of 64 and <= 4096, d_kv = 64, d_ff a multiple of 64)");
    return nullptr;
  }
  mx_t5* u = new mx_t5();
  u->cfg = *c;
  return u;
}
extern "C" void mx_t5_destroy(mx_t5* u) { delete u; }
extern "C" int mx_t5_set_weights(mx_t5* u, const void* blob, uint64_t blob_bytes, const mx_weight_entry* table, int n) {
  MX_CHECK(u && blob && table && n > 0, "t5_set_weights: bad arguments");
  u->table.clear();
  for (int i = 0; i < n; ++i) {
    MX_CHECK(table[i].name != nullptr && table[i].offset % 16 == 0 && table[i].offset + table[i].bytes <= blob_bytes, "t5_set_weights: bad entry");
    u->table[table[i].name] = {table[i].offset, table[i].bytes};
  }
  u->blob = (const char*)blob; u->blob_bytes = blob_bytes;
  return 0;
}
extern "C" size_t mx_t5_workspace_bytes(const mx_t5* u, int batch, int L) {
  size_t peak = 0;
  if (run_impl(const_cast<mx_t5*>(u), nullptr, nullptr, nullptr, batch, L, nullptr, 0, true, false, &peak)) return 0;
  return peak + 4096;
}
extern "C" int mx_t5_validate(const mx_t5* u, int batch, int L) {
  MX_CHECK(u && u->blob, "t5_validate: weights not set");
  return run_impl(const_cast<mx_t5*>(u), nullptr, nullptr, nullptr, batch, L, nullptr, 0, true, true, nullptr);
}
extern "C" int mx_t5_encode(mx_t5* u, void* stream, const int32_t* ids, void* out, int batch, int L, void* workspace, size_t workspace_bytes) {
  return run_impl(u, stream, ids, out, batch, L, workspace, workspace_bytes, false, false, nullptr);
}
