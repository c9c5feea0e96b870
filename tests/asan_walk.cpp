// Host-only walk of every step plan under AddressSanitizer + UBSan (SURVEY.md section 5 "race detection / sanitizers"; GPU sanitizers are not
// available on this pool, so the sanitizer covers what runs on the host: the ~3k lines of arena / cursor / offset arithmetic of the plans).
// Built by `make -C sduss_amd/csrc asan` from the library's own sources compiled --cuda-host-only; nothing here launches a kernel: only the
// dry-run entry points are called (workspace / state sizing, comm plans, grouped-launch tile bookkeeping), at the sizes the benchmark uses.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../include/mxdenoise.h"

#define REQUIRE(cond)                                                                    \
  do {                                                                                   \
    if (!(cond)) { std::fprintf(stderr, "asan_walk: %s failed at line %d: %s\n", #cond, __LINE__, mx_last_error()); return 1; } \
  } while (0)

static int n_exchanges = 0;
static size_t ws_limit = 0;
static int count_gather(void*, void*, const void* send, void* recv, size_t bytes) {
  const size_t so = (size_t)send - 0x1000, ro = (size_t)recv - 0x1000;
  if (so + bytes > ws_limit || ro + 8 * bytes > ws_limit + 8 * bytes) return 1;
  ++n_exchanges;
  return 0;
}

static mx_unet_config sdxl_base() {
  mx_unet_config c; std::memset(&c, 0, sizeof(c));
  c.in_channels = 4; c.out_channels = 4; c.n_levels = 3; c.layers_per_block = 2;
  const int ch[3] = {320, 640, 1280}, tl[3] = {1, 2, 10}, at[3] = {0, 1, 1};
  for (int i = 0; i < 3; ++i) { c.block_out_channels[i] = ch[i]; c.transformer_layers[i] = tl[i]; c.down_has_attn[i] = at[i]; c.num_heads[i] = ch[i] / 64; }
  c.cross_attention_dim = 2048; c.addition_time_embed_dim = 256; c.projection_class_embeddings_input_dim = 2816; c.norm_num_groups = 32;
  c.norm_eps = 1e-5f; c.transformer_norm_eps = 1e-6f; c.layer_norm_eps = 1e-5f;
  return c;
}

int main() {
  // ---- SDXL UNet: every batch / resolution of the predictor table's range, mixed groups, patch-parallel worlds, block-cache state ----
  mx_unet_config uc = sdxl_base();
  mx_unet* u = mx_unet_create(&uc);
  REQUIRE(u != nullptr);
  for (int batch : {1, 2, 3, 8, 16})
    for (int hw : {64, 96, 128}) REQUIRE(mx_unet_workspace_bytes(u, batch, hw, hw, 77) > 0);
  REQUIRE(mx_unet_workspace_bytes(u, 2, 40, 24, 77) > 0);                       // odd, non-square
  REQUIRE(mx_unet_workspace_bytes(u, 2, 30, 32, 77) == 0);                      // not divisible by 2^(levels-1): rejected, not walked
  mx_unet_group g[4];
  std::memset(g, 0, sizeof(g));
  const int res[3] = {64, 96, 128};
  for (int a = 1; a <= 4; a += 3)
    for (int b = 1; b <= 3; b += 2)
      for (int c = 1; c <= 4; c += 3) {
        const int n[3] = {2 * a, 2 * b, 2 * c};
        for (int i = 0; i < 3; ++i) { g[i].batch = n[i]; g[i].H = g[i].W = res[i]; }
        REQUIRE(mx_unet_workspace_bytes_mixed(u, g, 3, 77) > 0);
      }
  g[3].batch = 2; g[3].H = g[3].W = 32;
  REQUIRE(mx_unet_workspace_bytes_mixed(u, g, 4, 77) > 0);
  REQUIRE(mx_unet_workspace_bytes_mixed(u, g, 5, 77) == 0);
  for (int world : {2, 4, 8}) {
    const size_t ws = mx_unet_workspace_bytes_pp(u, 2, 128 / world, 128, 77, world);
    REQUIRE(ws > 0);
    REQUIRE(mx_unet_pp_state_bytes(u, 2, 128 / world, 128, 77, world) > 0);
    mx_pp_comm comm; comm.rank = world - 1; comm.world = world; comm.all_gather = count_gather; comm.ctx = nullptr;
    n_exchanges = 0; ws_limit = ws;
    REQUIRE(mx_unet_pp_comm_plan(u, 2, 128 / world, 128, 77, &comm) == 0);
    REQUIRE(n_exchanges > 100);
  }
  REQUIRE(mx_unet_block_cache_bytes(u, 8, 128, 128) > 0);
  mx_unet_destroy(u);

  // ---- grouped-launch bookkeeping of the GEMM front end: tile choice and statistics slabs over problem lists ----
  {
    mx_gemm_seg s[3]; std::memset(s, 0, sizeof(s));
    const int ms[3] = {512, 1152, 2048};
    for (int i = 0; i < 3; ++i) { s[i].M = ms[i]; s[i].a = (const void*)(uintptr_t)(0x100000 * (i + 1)); s[i].c = (void*)(uintptr_t)(0x9000000 + 0x100000 * i); }
    mx_gemm_desc d; std::memset(&d, 0, sizeof(d));
    d.a = s[0].a; d.w = (const void*)0x1000; d.c = s[0].c; d.N = 1280; d.K = 1280; d.lda = 1280; d.ldc = 1280; d.segs = s; d.n_segs = 3;
    REQUIRE(mx_gemm_stats_slabs(&d) > 0);
    d.N = 10240; d.flags = MX_EPI_GEGLU; d.ldc = 5120;
    (void)mx_gemm_ln_prefers_pass(&d);
  }

  // ---- SD3.5-medium MMDiT ----
  mx_mmdit_config mc; std::memset(&mc, 0, sizeof(mc));
  mc.patch_size = 2; mc.in_channels = 16; mc.out_channels = 16; mc.num_layers = 24; mc.num_attention_heads = 24; mc.joint_attention_dim = 4096;
  mc.pooled_projection_dim = 2048; mc.pos_embed_max_size = 384; mc.norm_eps = 1e-6f;
  for (int i = 0; i < 13; ++i) mc.dual_attention[i] = 1;
  mx_mmdit* m = mx_mmdit_create(&mc);
  REQUIRE(m != nullptr);
  for (int batch : {1, 2, 8})
    for (int hw : {64, 96, 128}) REQUIRE(mx_mmdit_workspace_bytes(m, batch, hw, hw, 333) > 0);
  for (int world : {2, 4, 8}) {
    REQUIRE(mx_mmdit_workspace_bytes_pp(m, 2, 128 / world, 128, 333, world) > 0);
    REQUIRE(mx_mmdit_pp_state_bytes(m, 2, 128 / world, 128, 333, world) > 0);
  }
  REQUIRE(mx_mmdit_block_cache_bytes(m, 8, 128, 128, 333) > 0);
  for (int i = 0; i < 3; ++i) { g[i].batch = 2 * (i + 1); g[i].H = g[i].W = res[i]; }
  REQUIRE(mx_mmdit_workspace_bytes_mixed(m, g, 3, 333) > 0);
  REQUIRE(mx_mmdit_workspace_bytes_mixed(m, g, 2, 333) > 0);
  mx_mmdit_destroy(m);

  // ---- VAE decoder, CLIP, T5 ----
  mx_vae_config vc; std::memset(&vc, 0, sizeof(vc));
  vc.latent_channels = 4; vc.out_channels = 3; vc.n_levels = 4; vc.layers_per_block = 2; vc.norm_num_groups = 32; vc.norm_eps = 1e-6f;
  const int vch[4] = {128, 256, 512, 512};
  for (int i = 0; i < 4; ++i) vc.block_out_channels[i] = vch[i];
  mx_vae* v = mx_vae_create(&vc);
  REQUIRE(v != nullptr);
  for (int hw : {64, 96, 128}) REQUIRE(mx_vae_workspace_bytes(v, 2, hw, hw) > 0);
  mx_vae_destroy(v);
  mx_clip_config cc; std::memset(&cc, 0, sizeof(cc));
  cc.vocab_size = 49408; cc.hidden_size = 1280; cc.intermediate_size = 5120; cc.num_hidden_layers = 32; cc.num_attention_heads = 20;
  cc.max_position_embeddings = 77; cc.hidden_act = 1; cc.projection_dim = 1280; cc.eos_token_id = 2; cc.hidden_layer = -2; cc.layer_norm_eps = 1e-5f;
  mx_clip* c = mx_clip_create(&cc);
  REQUIRE(c != nullptr && mx_clip_workspace_bytes(c, 8) > 0);
  mx_clip_destroy(c);
  mx_t5_config tc; std::memset(&tc, 0, sizeof(tc));
  tc.vocab_size = 32128; tc.d_model = 4096; tc.d_ff = 10240; tc.num_layers = 24; tc.num_heads = 64; tc.layer_norm_epsilon = 1e-6f;
  mx_t5* t = mx_t5_create(&tc);
  REQUIRE(t != nullptr && mx_t5_workspace_bytes(t, 2, 256) > 0);
  mx_t5_destroy(t);
  std::printf("ASAN_WALK_OK\n");
  return 0;
}
