// Step plan of the SDXL VAE DECODER (AutoencoderKL.decode), the step AFTER the denoising loop: the reference's post_inference
// (sduss/model_executor/diffusers/pipelines/stable_diffusion_xl/pipeline_stable_diffusion_xl_esymred.py:406-463) divides the latents
// by the scaling factor and calls self.vae.decode (un-vendored diffusers 0.32.1 AutoencoderKL).
// Same building blocks as the UNet plan (NHWC bf16, implicit-GEMM 3x3 convs with the nearest-2x upsample fused in the loader,
// GroupNorm+SiLU, GEMM epilogues); the one new piece is the mid block's single-head 512-wide attention, run as
// GEMM (q k^T / sqrt(C)) -> row softmax -> GEMM (P V) because the flash kernel is specialised for 64-wide heads.
//   decode(z):  post_quant_conv(z / scaling)  ->  conv_in  ->  mid: resnet, attention, resnet  ->  up blocks (3 resnets each,
//               nearest-2x + conv between them)  ->  GroupNorm + SiLU  ->  conv_out      (the scaling is folded into post_quant_conv)
// The reference runs this in fp32 (force_upcast, :48-52); here bf16 storage / fp32 accumulate, tolerance stated in tests/test_vae_gpu.py.
#include <hip/hip_runtime.h>

#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/mxdenoise.h"
#include "common.h"

namespace mx {
int launch_prep_latent(hipStream_t s, const void* in, int dtype, void* out, int B, int Cin, int HW, int CP);
int launch_nhwc_to_nchw(hipStream_t s, const void* in, void* out, int dtype, int B, int C, int HW, int ld);
int launch_softmax_rows(hipStream_t s, void* scores, long rows, int L, long ld);
size_t gn_workspace_exact(int B, int H, int W, int C, int patch);
}  // namespace mx

using mx::bf16_t;

struct mx_vae {
  mx_vae_config cfg;
  const char* blob = nullptr;
  uint64_t blob_bytes = 0;
  std::unordered_map<std::string, std::pair<uint64_t, uint64_t>> table;
};

namespace {
constexpr int kPad = 64;   // the 4 latent channels are zero-padded to one K tile

struct Arena {
  char* base; size_t cap; size_t top; size_t peak; bool dry;
  void* alloc(size_t bytes) {
    const size_t a = (top + 255) & ~(size_t)255;
    top = a + bytes;
    if (top > peak) peak = top;
    if (dry) return (void*)(uintptr_t)(0x1000 + a);
    return (top <= cap) ? base + a : nullptr;
  }
  size_t mark() const { return top; }
  void release(size_t m) { top = m; }
};

struct Plan {
  mx_vae* u; hipStream_t stream; Arena ar; int B, H, W; bool dry, lookup; std::string err;
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

  bool gemm(mx_gemm_desc& d, bool conv) {
    if (!ok()) return false;
    if (dry) return true;
    if (conv ? mx_conv3x3(stream, &d) : mx_gemm(stream, &d)) return fail(std::string("gemm/conv: ") + mx_last_error());
    return true;
  }
  bool linear(const bf16_t* a, int lda, const bf16_t* wgt, const float* bias, void* c, int ldc, int M, int N, int K, const void* residual = nullptr,
              int ldr = 0, float out_scale = 0.f) {
    mx_gemm_desc d; std::memset(&d, 0, sizeof(d));
    d.a = a; d.lda = lda; d.w = wgt; d.bias = bias; d.c = c; d.ldc = ldc; d.M = M; d.N = N; d.K = K; d.residual = residual; d.ldr = ldr; d.out_scale = out_scale;
    return gemm(d, false);
  }
  bool conv(const bf16_t* x, int h, int wd, int Cin, const std::string& prefix, bf16_t* out, int Cout, int up, const void* residual = nullptr) {
    mx_gemm_desc d; std::memset(&d, 0, sizeof(d));
    d.a = x; d.w = wb(prefix + ".weight", (size_t)Cout * 9 * Cin); d.bias = wf(prefix + ".bias", Cout);
    d.c = out; d.ldc = Cout; d.B = B; d.Hin = h; d.Win = wd; d.Cin = Cin; d.Hout = h << up; d.Wout = wd << up; d.stride = 1; d.up = up;
    d.M = B * d.Hout * d.Wout; d.N = Cout; d.K = 9 * Cin; d.rows_per_batch = d.Hout * d.Wout; d.residual = residual; d.ldr = Cout;
    return gemm(d, true);
  }
  bool groupnorm(const bf16_t* x, bf16_t* y, const std::string& prefix, int h, int wd, int C, bool silu) {
    if (!ok()) return false;
    const size_t m = ar.mark();
    void* ws = ar.alloc(mx::gn_workspace_exact(B, h, wd, C, 0));
    if (!ws) return fail("workspace too small");
    const float* g = wf(prefix + ".weight", C); const float* b = wf(prefix + ".bias", C);
    if (ok() && !dry && mx_groupnorm_nhwc(stream, x, y, g, b, B, h, wd, C, u->cfg.norm_num_groups, u->cfg.norm_eps, silu ? 1 : 0, 0, ws))
      fail(std::string("groupnorm: ") + mx_last_error());
    ar.release(m);
    return ok();
  }
  // diffusers ResnetBlock2D without time embedding: GN-SiLU-conv, GN-SiLU-conv, (1x1 shortcut), add
  bf16_t* resnet(const std::string& p, const bf16_t* x, int h, int wd, int Cin, int Cout) {
    const size_t M = (size_t)B * h * wd;
    bf16_t* out = alloc<bf16_t>(M * Cout);
    const size_t m = ar.mark();
    bf16_t* n1 = alloc<bf16_t>(M * Cin);
    groupnorm(x, n1, p + ".norm1", h, wd, Cin, true);
    bf16_t* h1 = alloc<bf16_t>(M * Cout);
    conv(n1, h, wd, Cin, p + ".conv1", h1, Cout, 0);
    bf16_t* n2 = alloc<bf16_t>(M * Cout);
    groupnorm(h1, n2, p + ".norm2", h, wd, Cout, true);
    const bf16_t* sc = x;
    if (Cin != Cout) {
      bf16_t* s2 = alloc<bf16_t>(M * Cout);
      linear(x, Cin, wb(p + ".conv_shortcut.weight", (size_t)Cout * Cin), wf(p + ".conv_shortcut.bias", Cout), s2, Cout, (int)M, Cout, Cin);
      sc = s2;
    }
    conv(n2, h, wd, Cout, p + ".conv2", out, Cout, 0, sc);
    ar.release(m);
    return out;
  }
  // mid-block attention: one head of width C over the h*wd tokens of each image
  bf16_t* attention(const std::string& p, const bf16_t* x, int h, int wd, int C) {
    const int L = h * wd;
    const size_t M = (size_t)B * L;
    bf16_t* out = alloc<bf16_t>(M * C);
    const size_t m = ar.mark();
    if (L % 8 != 0) fail("vae attention: tokens per image must be a multiple of 8");
    bf16_t* n = alloc<bf16_t>(M * C);
    groupnorm(x, n, p + ".group_norm", h, wd, C, false);
    bf16_t* q = alloc<bf16_t>(M * C);
    bf16_t* k = alloc<bf16_t>(M * C);
    bf16_t* vt = alloc<bf16_t>((size_t)C * L);         // V^T of one image: [C, L]
    bf16_t* sc = alloc<bf16_t>((size_t)L * L);         // scores of one image
    bf16_t* o = alloc<bf16_t>(M * C);
    linear(n, C, wb(p + ".to_q.weight", (size_t)C * C), wf(p + ".to_q.bias", C), q, C, (int)M, C, C);
    linear(n, C, wb(p + ".to_k.weight", (size_t)C * C), wf(p + ".to_k.bias", C), k, C, (int)M, C, C);
    const float scale = 1.0f / std::sqrt((float)C);
    for (int b = 0; b < B && ok(); ++b) {
      const bf16_t* nb = n + (size_t)b * L * C;
      // V^T = Wv n^T (the value bias is added after P V: the rows of P sum to one)
      linear(wb(p + ".to_v.weight", (size_t)C * C), C, nb, nullptr, vt, L, C, L, C);
      linear(q + (size_t)b * L * C, C, k + (size_t)b * L * C, nullptr, sc, L, L, L, C, nullptr, 0, scale);
      if (ok() && !dry && mx::launch_softmax_rows(stream, sc, L, L, L)) fail(mx_last_error());
      linear(sc, L, vt, wf(p + ".to_v.bias", C), o + (size_t)b * L * C, C, L, C, L);
    }
    linear(o, C, wb(p + ".to_out.0.weight", (size_t)C * C), wf(p + ".to_out.0.bias", C), out, C, (int)M, C, C, x, C);
    ar.release(m);
    return out;
  }

  bool run(const void* latents, int io_dtype, void* outp, int out_dtype) {
    const mx_vae_config& c = u->cfg;
    const int nlev = c.n_levels;
    int h = H, wd = W;
    const size_t M0 = (size_t)B * h * wd;
    bf16_t* z0 = alloc<bf16_t>(M0 * kPad);
    if (ok() && !dry && mx::launch_prep_latent(stream, latents, io_dtype, z0, B, c.latent_channels, h * wd, kPad)) fail(mx_last_error());
    bf16_t* z1 = alloc<bf16_t>(M0 * kPad);             // post_quant_conv (1x1; 1 / scaling_factor folded into its weight), padded to 64 channels
    linear(z0, kPad, wb("post_quant_conv.weight", (size_t)kPad * kPad), wf("post_quant_conv.bias", kPad), z1, kPad, (int)M0, kPad, kPad);
    int C = c.block_out_channels[nlev - 1];
    bf16_t* x = alloc<bf16_t>(M0 * C);
    conv(z1, h, wd, kPad, "decoder.conv_in", x, C, 0);
    x = resnet("decoder.mid_block.resnets.0", x, h, wd, C, C);
    x = attention("decoder.mid_block.attentions.0", x, h, wd, C);
    x = resnet("decoder.mid_block.resnets.1", x, h, wd, C, C);
    for (int i = 0; i < nlev && ok(); ++i) {
      const int Cout = c.block_out_channels[nlev - 1 - i];
      for (int j = 0; j < c.layers_per_block + 1 && ok(); ++j) {
        x = resnet("decoder.up_blocks." + std::to_string(i) + ".resnets." + std::to_string(j), x, h, wd, C, Cout);
        C = Cout;
      }
      if (i != nlev - 1) {
        bf16_t* d = alloc<bf16_t>((size_t)B * (2 * h) * (2 * wd) * C);
        conv(x, h, wd, C, "decoder.up_blocks." + std::to_string(i) + ".upsamplers.0.conv", d, C, 1);
        h *= 2; wd *= 2;
        x = d;
      }
    }
    const size_t M = (size_t)B * h * wd;
    bf16_t* n = alloc<bf16_t>(M * C);
    groupnorm(x, n, "decoder.conv_norm_out", h, wd, C, true);
    const int ldo = (c.out_channels + 3) / 4 * 4;
    bf16_t* o = alloc<bf16_t>(M * ldo);
    conv(n, h, wd, C, "decoder.conv_out", o, ldo, 0);
    if (ok() && !dry && mx::launch_nhwc_to_nchw(stream, o, outp, out_dtype, B, c.out_channels, h * wd, ldo)) fail(mx_last_error());
    return ok();
  }
};

int run_impl(mx_vae* u, void* stream, const void* latents, int io_dtype, void* out, int out_dtype, int batch, int H, int W, void* ws, size_t ws_bytes,
             bool dry, bool lookup, size_t* peak) {
  MX_CHECK(u != nullptr, "vae: null handle");
  MX_CHECK(batch > 0 && H > 0 && W > 0, "vae: bad shape");
  if (!dry) {
    MX_CHECK(latents && out && ws && u->blob, "vae: null operand or weights not set");
    MX_CHECK(io_dtype >= 0 && io_dtype <= 2 && out_dtype >= 0 && out_dtype <= 2, "vae: bad dtype");
  }
  Plan p;
  p.u = u; p.stream = (hipStream_t)stream; p.B = batch; p.H = H; p.W = W; p.dry = dry; p.lookup = lookup;
  p.ar.base = (char*)ws; p.ar.cap = ws_bytes; p.ar.top = 0; p.ar.peak = 0; p.ar.dry = dry;
  const bool okr = p.run(latents, io_dtype, out, out_dtype);
  if (peak) *peak = p.ar.peak;
  if (!okr) { mx::set_error(p.err); return 1; }
  return 0;
}
}  // namespace

extern "C" mx_vae* mx_vae_create(const mx_vae_config* c) {
  if (!c || c->n_levels < 1 || c->n_levels > 4 || c->latent_channels <= 0 || c->latent_channels > kPad || c->out_channels <= 0 || c->out_channels > 4) {
    mx::set_error("vae: bad config");
    return nullptr;
  }
  for (int i = 0; i < c->n_levels; ++i)
    if (c->block_out_channels[i] % 64 != 0 || c->block_out_channels[i] % c->norm_num_groups != 0) { mx::set_error("vae: block_out_channels must be multiples of 64 and of the group count"); return nullptr; }
  mx_vae* u = new mx_vae();
  u->cfg = *c;
  return u;
}
extern "C" void mx_vae_destroy(mx_vae* u) { delete u; }
extern "C" int mx_vae_set_weights(mx_vae* u, const void* blob, uint64_t blob_bytes, const mx_weight_entry* table, int n) {
  MX_CHECK(u && blob && table && n > 0, "vae_set_weights: bad arguments");
  u->table.clear();
  for (int i = 0; i < n; ++i) {
    MX_CHECK(table[i].name != nullptr && table[i].offset % 16 == 0 && table[i].offset + table[i].bytes <= blob_bytes, "vae_set_weights: bad entry");
    u->table[table[i].name] = {table[i].offset, table[i].bytes};
  }
  u->blob = (const char*)blob; u->blob_bytes = blob_bytes;
  return 0;
}
extern "C" size_t mx_vae_workspace_bytes(const mx_vae* u, int batch, int H, int W) {
  size_t peak = 0;
  if (run_impl(const_cast<mx_vae*>(u), nullptr, nullptr, MX_BF16, nullptr, MX_F32, batch, H, W, nullptr, 0, true, false, &peak)) return 0;
  return peak + 4096;
}
extern "C" int mx_vae_validate(const mx_vae* u, int batch, int H, int W) {
  MX_CHECK(u && u->blob, "vae_validate: weights not set");
  return run_impl(const_cast<mx_vae*>(u), nullptr, nullptr, MX_BF16, nullptr, MX_F32, batch, H, W, nullptr, 0, true, true, nullptr);
}
extern "C" int mx_vae_decode(mx_vae* u, void* stream, const void* latents, int io_dtype, void* out, int out_dtype, int batch, int H, int W,
                             void* workspace, size_t workspace_bytes) {
  return run_impl(u, stream, latents, io_dtype, out, out_dtype, batch, H, W, workspace, workspace_bytes, false, false, nullptr);
}
