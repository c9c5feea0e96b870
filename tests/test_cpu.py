"""CPU suite (-m "not gpu"): the oracle against itself and its invariants, the host logic, and that the C-ABI library
loads and exports every symbol include/mxdenoise.h declares (no compute calls without a GPU)."""
import ctypes as C
import os
import re

import pytest
import numpy as np
import torch
import torch.nn.functional as F

from oracle import patch_ref, scheduler_ref, sdxl_unet_ref as ref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def tiny():
    cfg = ref.UNetConfig.tiny()
    return cfg, ref.init_params(cfg)


def test_oracle_gn_single_patch_is_group_norm():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 64, 8, 8, generator=g); w = torch.randn(64, generator=g); b = torch.randn(64, generator=g)
    assert torch.allclose(ref.group_norm_patchavg(x, 32, w, b, 1e-5, 8), F.group_norm(x, 32, w, b, 1e-5), atol=1e-5)


def test_oracle_split_concat_roundtrip():
    g = torch.Generator().manual_seed(1)
    samples = {"256": torch.randn(2, 4, 32, 32, generator=g), "512": torch.randn(1, 4, 64, 64, generator=g)}
    pidx, lo, ro, patches, pm = patch_ref.split_sample(samples, 128)
    assert patches.shape == (2 * 4 + 16, 4, 18, 18) and lo == [0, 4, 8, 24] and ro == [0, 2, 3]
    assert pm.tolist() == [1] * 4 + [2] * 4 + [3] * 16
    back = patch_ref.concat_sample(128, patches[:, :, 1:-1, 1:-1], lo)
    for k in samples:
        assert torch.equal(back[k], samples[k])
    # halo cells of split_sample are the true neighbour pixels / zero at the image border
    assert torch.equal(patches[0, :, 1:-1, -1], samples["256"][0, :, 0:16, 16])
    assert patches[0, :, 0, :].abs().max() == 0


def test_oracle_halo_semantics():
    """interior copy, edge rows/cols from neighbours, corners replicated by the left/right sender (cu:210-239)."""
    g = torch.Generator().manual_seed(2)
    img = torch.randn(1, 3, 8, 8, generator=g)
    pidx, lo, ro, patches, pm = patch_ref.split_sample({"64": img}, 32)   # 2x2 patches of 4x4
    x = patches[:, :, 1:-1, 1:-1].contiguous()
    y = patch_ref.mock_groupnorm(x, pidx)
    assert torch.equal(y[:, :, 1:-1, 1:-1], x)
    assert torch.equal(y[0, :, 1:-1, 5], x[1, :, :, 0])      # right halo of patch 0 = first column of patch 1
    assert torch.equal(y[0, :, 5, 1:-1], x[2, :, 0, :])      # bottom halo of patch 0 = first row of patch 2
    assert torch.equal(y[0, :, 5, 5], x[1, :, 3, 0])         # corner: replicated from the RIGHT neighbour, not the diagonal
    assert not torch.equal(y[0, :, 5, 5], x[3, :, 0, 0])
    assert torch.equal(y[0, :, 0, 5], x[1, :, 0, 0])         # image-border junction gets a non-zero corner
    assert y[0, :, 0, 0:5].abs().max() == 0


def test_oracle_sliced_equals_whole_image_with_corner_rule(tiny):
    cfg, P = tiny
    s, t, e, te, ti = ref.make_inputs(cfg, 2, 32)
    for patch_px, gp in ((128, 16), (64, 8)):
        lit = patch_ref.unet_forward_sliced(P, cfg, {"256": s}, t, e, te, ti, patch_size=patch_px)["256"]
        whole = ref.unet_forward(P, cfg, s, t, e, te, ti, gn_patch=gp, sliced_corners=True)
        assert (lit - whole).abs().max() < 5e-5
    one = patch_ref.unet_forward_sliced(P, cfg, {"256": s}, t, e, te, ti, patch_size=256)["256"]
    assert (one - ref.unet_forward(P, cfg, s, t, e, te, ti)).abs().max() < 5e-5


def test_oracle_euler_closed_form():
    ts, sig, init = scheduler_ref.sdxl_euler_tables(50)
    assert ts[0] == 981 and ts[-1] == 1 and len(sig) == 51 and sig[-1] == 0
    assert abs(sig[0].item() - 13.0) < 1.0 and abs(init - (sig[0].item() ** 2 + 1) ** 0.5) < 1e-6
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 4, 8, 8, generator=g); eps = torch.randn(2, 4, 8, 8, generator=g)
    out = scheduler_ref.euler_step(eps, x, sig[[3, 4]], sig[[4, 5]])
    assert torch.allclose(out, x + eps * (sig[[4, 5]] - sig[[3, 4]]).reshape(2, 1, 1, 1), atol=1e-5)


def test_param_inventories_agree():
    from sduss_amd import config
    for a, b in ((config.UNetConfig.tiny(), ref.UNetConfig.tiny()), (config.UNetConfig.sdxl_base(), ref.UNetConfig.sdxl_base())):
        pa, pb = config.param_shapes(a), ref.param_shapes(b)
        assert set(pa.items()) == set(pb.items())
    total = sum(torch.Size(v).numel() for v in config.param_shapes(config.UNetConfig.sdxl_base()).values())
    assert abs(total - 2.567e9) < 2e6     # SDXL-base UNet: 2.57 B parameters


def test_library_exports_every_declared_symbol():
    from sduss_amd import lib
    l = lib.load()
    hdr = open(os.path.join(ROOT, "include", "mxdenoise.h")).read()
    declared = set(re.findall(r"\b(mx_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(lib.SYMBOLS), declared ^ set(lib.SYMBOLS)
    for name in declared:
        assert hasattr(l, name)
    assert l.mx_version() == 1


def test_step_plan_resolves_packed_weights_on_host(tiny):
    """mx_unet_validate walks the C++ step plan with no launches: every packed tensor name/size the plan asks for must
    be what weights.pack produced."""
    from sduss_amd import config, lib, weights
    cfg, P = tiny
    l = lib.load()
    pcfg = config.UNetConfig.tiny()
    pw = weights.PackedWeights(weights.pack(pcfg, P), "cpu")
    cc = lib.UNetConfigC()
    cc.in_channels, cc.out_channels, cc.n_levels, cc.layers_per_block = 4, 4, 3, 2
    for i, v in enumerate(pcfg.block_out_channels):
        cc.block_out_channels[i] = v; cc.down_has_attn[i] = int(pcfg.down_has_attn[i])
        cc.transformer_layers[i] = pcfg.transformer_layers_per_block[i]; cc.num_heads[i] = pcfg.num_heads[i]
    cc.cross_attention_dim, cc.addition_time_embed_dim = pcfg.cross_attention_dim, pcfg.addition_time_embed_dim
    cc.projection_class_embeddings_input_dim, cc.norm_num_groups = pcfg.projection_class_embeddings_input_dim, 32
    h = l.mx_unet_create(C.byref(cc))
    assert h
    assert l.mx_unet_set_weights(h, pw.blob.data_ptr(), pw.blob.numel(), pw.table, len(pw.names)) == 0
    assert l.mx_unet_validate(h, 2, 32, 32, 77) == 0, l.mx_last_error()
    assert l.mx_unet_workspace_bytes(h, 2, 32, 32, 77) > 0
    # a wrong-sized tensor is reported, not ignored
    bad = [(n, t if n != "conv_out.bias" else torch.zeros(8)) for n, t in weights.pack(pcfg, P)]
    pw2 = weights.PackedWeights(bad, "cpu")
    assert l.mx_unet_set_weights(h, pw2.blob.data_ptr(), pw2.blob.numel(), pw2.table, len(pw2.names)) == 0
    assert l.mx_unet_validate(h, 2, 32, 32, 77) != 0 and b"conv_out.bias" in l.mx_last_error()
    l.mx_unet_destroy(h)


def test_mmdit_inventory_and_plan_on_host():
    """SD3.5 MMDiT: the product's tensor inventory equals the oracle's (2.47 B with the positional table at medium size)
    and the C++ step plan resolves every packed tensor (no launches)."""
    from oracle import sd3_mmdit_ref as mref
    from sduss_amd import config, lib, weights
    for a, b in ((config.MMDiTConfig.tiny(), mref.MMDiTConfig.tiny()), (config.MMDiTConfig.sd35_medium(), mref.MMDiTConfig.sd35_medium())):
        assert set(config.mmdit_param_shapes(a).items()) == set(mref.param_shapes(b).items())
    total = sum(torch.Size(v).numel() for v in config.mmdit_param_shapes(config.MMDiTConfig.sd35_medium()).values())
    assert abs(total - 2.47e9) < 1e7
    cfg = config.MMDiTConfig.tiny()
    P = mref.init_params(mref.MMDiTConfig.tiny())
    pw = weights.PackedWeights(weights.pack_mmdit(cfg, P), "cpu")
    l = lib.load()
    cc = lib.MMDiTConfigC()
    cc.patch_size, cc.in_channels, cc.out_channels, cc.num_layers, cc.num_attention_heads = 2, 16, 16, cfg.num_layers, cfg.num_attention_heads
    cc.joint_attention_dim, cc.pooled_projection_dim, cc.pos_embed_max_size, cc.norm_eps = 128, 64, cfg.pos_embed_max_size, 1e-6
    for i in cfg.dual_attention_layers:
        cc.dual_attention[i] = 1
    h = l.mx_mmdit_create(C.byref(cc))
    assert h and l.mx_mmdit_set_weights(h, pw.blob.data_ptr(), pw.blob.numel(), pw.table, len(pw.names)) == 0
    assert l.mx_mmdit_validate(h, 2, 16, 16, 37) == 0, l.mx_last_error()
    assert l.mx_mmdit_validate(h, 1, 64, 64, 37) != 0 and b"positional table" in l.mx_last_error()
    assert l.mx_mmdit_workspace_bytes(h, 2, 16, 16, 37) > 0
    l.mx_mmdit_destroy(h)


def test_oracle_mmdit_token_order_invariance():
    """the sliced branch of the reference only re-chunks the token axis; joint attention is permutation-equivariant over
    image tokens, so shuffling patch positions (with their positional rows) must shuffle the output the same way."""
    from oracle import sd3_mmdit_ref as mref
    cfg = mref.MMDiTConfig.tiny()
    P = mref.init_params(cfg)
    lat, t, e, p = mref.make_inputs(cfg, 1, 16, ctx_len=9)
    tr = {}
    mref.mmdit_forward(P, cfg, lat, t, e, p, trace=tr)
    x0 = tr["embed"]
    perm = torch.randperm(x0.shape[1], generator=torch.Generator().manual_seed(0))
    x, ctx = x0[:, perm], tr["context_embed"]
    y, _ = mref.joint_block(P, "transformer_blocks.0", x, ctx, tr["temb"], cfg, False, True)
    y0, _ = mref.joint_block(P, "transformer_blocks.0", x0, ctx, tr["temb"], cfg, False, True)
    assert torch.allclose(y, y0[:, perm], atol=1e-4)


def test_bad_arguments_raise():
    from sduss_amd import lib
    l = lib.load()
    d = lib.GemmDesc()
    assert l.mx_gemm(None, C.byref(d)) != 0 and b"gemm" in l.mx_last_error()
    with pytest.raises(lib.MxError):
        lib.check(l.mx_attention(None, None, 0, None, 0, None, 0, 0, None, 0, 1, 1, 1, 1, 0.125), "mx_attention")
    assert l.mx_unet_create(None) is None


def test_c_restatement_agrees_with_python_restatement():
    """Two independent restatements of the native op (oracle/gn_halo_ref.c, oracle/patch_ref.py) must agree."""
    import subprocess
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    lib = C.CDLL(os.path.join(ROOT, "oracle", "_build", "libgnhalo_ref.so"))
    g = torch.Generator().manual_seed(4)
    c, cpg = 16, 4
    samples = {"256": torch.randn(1, c, 32, 32, generator=g), "384": torch.randn(2, c, 48, 48, generator=g)}
    pidx, lo, _ro, patches, pm = patch_ref.split_sample(samples, 128)
    x = patches[:, :, 1:-1, 1:-1].contiguous()
    ga = torch.randn(c, generator=g); be = torch.randn(c, generator=g)
    n, _, h, w = x.shape
    fp = C.POINTER(C.c_float); ip = C.POINTER(C.c_int)
    lo_t = torch.tensor(lo, dtype=torch.int32)
    for padding in (1, 0):
        want = patch_ref.groupnorm(x, ga, be, cpg, 1e-5, bool(padding), lo, pm, pidx)
        out = torch.empty_like(want)
        rc = lib.gnhalo_groupnorm(C.cast(x.data_ptr(), fp), C.cast(ga.data_ptr(), fp), C.cast(be.data_ptr(), fp),
                                  C.cast(out.data_ptr(), fp), n, c, h, w, cpg, C.c_double(1e-5), padding,
                                  C.cast(lo_t.data_ptr(), ip), C.cast(pm.data_ptr(), ip), C.cast(pidx.data_ptr(), ip))
        assert rc == 0 and torch.allclose(out, want, atol=2e-5)
    out = torch.empty(n, c, h + 2, w + 2)
    lib.gnhalo_mock(C.cast(x.data_ptr(), fp), C.cast(out.data_ptr(), fp), C.cast(pidx.data_ptr(), ip), n, c, h, w)
    assert torch.equal(out, patch_ref.mock_groupnorm(x, pidx))


def test_geglu_interleave_layout():
    from sduss_amd.weights import _geglu_interleave
    t = torch.arange(8 * 64).float()          # dim 64: hidden rows 0..255, gate rows 256..511
    p = _geglu_interleave(t)
    assert p[:32].tolist() == list(range(0, 32)) and p[32:64].tolist() == list(range(256, 288))
    assert p[64:96].tolist() == list(range(32, 64))


def test_step_cache_follows_host_step_indices():
    """sduss_amd/step_state.py: per-composition sigma / timestep table + device-side step index == the per-step lists the
    reference builds (scheduling_euler_discrete.py:171-175, 213-217), including a rewound request and latents reuse."""
    from types import SimpleNamespace
    from sduss_amd.pipeline import euler_tables
    from sduss_amd.step_state import StepCache
    t50, s50, _ = euler_tables(50)
    t30, s30, _ = euler_tables(30)
    reqs = [SimpleNamespace(request_id=i, timesteps=t, sigmas=s, step_index=k, latents=torch.full((1, 4, 2, 2), float(i)))
            for i, (t, s, k) in enumerate(((t50, s50, 0), (t30, s30, 7), (t50, s50, 49)))]
    cache = StepCache("cpu")
    built = []
    e = cache.entry(("k",), reqs, lambda: built.append(1) or ("cond",))
    assert cache.entry(("k",), reqs, lambda: built.append(1) or ("x",)) is e and built == [1]
    for it in range(3):
        if it == 2:
            reqs[2].step_index = 0            # the caller rewinds a request (bench.py keeps its batch full this way)
        sig, sig_next, ts = cache.step_scalars(e, reqs)
        for i, r in enumerate(reqs):
            assert sig[i].item() == r.sigmas[r.step_index] and sig_next[i].item() == r.sigmas[r.step_index + 1]
            assert ts[i].item() == r.timesteps[r.step_index]
        lat = cache.latents(e, reqs)
        assert [lat[i, 0, 0, 0].item() for i in range(3)] == [0.0, 1.0, 2.0]
        if it > 0:
            assert lat.data_ptr() == prev.data_ptr(), "latents buffer must be reused once the requests hold views of it"
        prev = lat
        for i, r in enumerate(reqs):
            r.latents = lat[i:i + 1]
            if it < 1 or i < 2:
                r.step_index += 1
        if it == 0:
            reqs[2].step_index = 49           # stays at its last step (done() would retire it in the real loop)
    reqs[0].step_index = 50
    with pytest.raises(IndexError):
        cache.step_scalars(e, reqs)


def _validate_unet(l, pcfg, pw, batch, hw):
    from sduss_amd import lib
    cc = lib.UNetConfigC()
    cc.in_channels, cc.out_channels, cc.n_levels, cc.layers_per_block = pcfg.in_channels, pcfg.out_channels, len(pcfg.block_out_channels), pcfg.layers_per_block
    for i, v in enumerate(pcfg.block_out_channels):
        cc.block_out_channels[i] = v; cc.down_has_attn[i] = int(pcfg.down_has_attn[i])
        cc.transformer_layers[i] = pcfg.transformer_layers_per_block[i]; cc.num_heads[i] = pcfg.num_heads[i]
    cc.cross_attention_dim, cc.addition_time_embed_dim = pcfg.cross_attention_dim, pcfg.addition_time_embed_dim
    cc.projection_class_embeddings_input_dim, cc.norm_num_groups = pcfg.projection_class_embeddings_input_dim, pcfg.norm_num_groups
    h = l.mx_unet_create(C.byref(cc))
    assert h and l.mx_unet_set_weights(h, pw.blob.data_ptr(), pw.blob.numel(), pw.table, len(pw.names)) == 0
    rc = l.mx_unet_validate(h, batch, hw, hw, 77)
    msg = l.mx_last_error()
    l.mx_unet_destroy(h)
    return rc, msg


def test_hf_layout_directories_load_into_the_step_plans(tmp_path):
    """The calls INTEGRATION.md tells a maintainer to make: an HF-layout ``unet/`` and ``transformer/`` folder
    (config.json in diffusers' key set + diffusion_pytorch_model.safetensors, model_loader.py:64-66) goes through
    from_hf_json / load_safetensors_dir / pack into mx_*_validate.  A stray fp16 shard is ignored."""
    import json
    from safetensors.torch import save_file
    from oracle import sd3_mmdit_ref as mref, sdxl_unet_ref as uref
    from sduss_amd import config, lib, weights
    l = lib.load()
    # ---- unet/ ----
    ocfg = uref.UNetConfig.tiny()
    P = uref.init_params(ocfg)
    pc = config.UNetConfig.tiny()
    ud = tmp_path / "unet"; ud.mkdir()
    hf = {"_class_name": "UNet2DConditionModel", "_diffusers_version": "0.32.1", "in_channels": pc.in_channels, "out_channels": pc.out_channels,
          "block_out_channels": list(pc.block_out_channels), "layers_per_block": pc.layers_per_block,
          "down_block_types": ["CrossAttnDownBlock2D" if a else "DownBlock2D" for a in pc.down_has_attn],
          "up_block_types": ["CrossAttnUpBlock2D" if a else "UpBlock2D" for a in reversed(pc.down_has_attn)],
          "attention_head_dim": list(pc.num_heads), "transformer_layers_per_block": list(pc.transformer_layers_per_block),
          "cross_attention_dim": pc.cross_attention_dim, "addition_embed_type": "text_time", "addition_time_embed_dim": pc.addition_time_embed_dim,
          "projection_class_embeddings_input_dim": pc.projection_class_embeddings_input_dim, "norm_num_groups": pc.norm_num_groups,
          "norm_eps": 1e-5, "use_linear_projection": True, "sample_size": 128, "act_fn": "silu"}
    (ud / "config.json").write_text(json.dumps(hf))
    save_file({k: v.to(torch.float16).contiguous() for k, v in P.items()}, str(ud / "diffusion_pytorch_model.safetensors"))
    save_file({"junk": torch.zeros(1)}, str(ud / "diffusion_pytorch_model.fp16.safetensors"))
    cfg, params = weights.load_safetensors_dir(str(ud))
    assert cfg == pc and set(params) == set(P)
    rc, msg = _validate_unet(l, cfg, weights.PackedWeights(weights.pack(cfg, params), "cpu"), 2, 32)
    assert rc == 0, msg
    # ---- transformer/ ----
    mo = mref.MMDiTConfig.tiny()
    PM = mref.init_params(mo)
    mc = config.MMDiTConfig.tiny()
    td = tmp_path / "transformer"; td.mkdir()
    hf = {"_class_name": "SD3Transformer2DModel", "sample_size": mc.sample_size, "patch_size": mc.patch_size, "in_channels": mc.in_channels,
          "out_channels": mc.out_channels, "num_layers": mc.num_layers, "attention_head_dim": mc.attention_head_dim,
          "num_attention_heads": mc.num_attention_heads, "joint_attention_dim": mc.joint_attention_dim,
          "caption_projection_dim": mc.caption_projection_dim, "pooled_projection_dim": mc.pooled_projection_dim,
          "pos_embed_max_size": mc.pos_embed_max_size, "dual_attention_layers": list(mc.dual_attention_layers), "qk_norm": "rms_norm"}
    (td / "config.json").write_text(json.dumps(hf))
    save_file({k: v.to(torch.float16).contiguous() for k, v in PM.items()}, str(td / "diffusion_pytorch_model.safetensors"))
    cfg, params = weights.load_mmdit_safetensors_dir(str(td))
    assert cfg == mc and set(params) == set(PM)
    pw = weights.PackedWeights(weights.pack_mmdit(cfg, params), "cpu")
    cc = lib.MMDiTConfigC()
    cc.patch_size, cc.in_channels, cc.out_channels, cc.num_layers, cc.num_attention_heads = cfg.patch_size, cfg.in_channels, cfg.out_channels, cfg.num_layers, cfg.num_attention_heads
    cc.joint_attention_dim, cc.pooled_projection_dim, cc.pos_embed_max_size, cc.norm_eps = cfg.joint_attention_dim, cfg.pooled_projection_dim, cfg.pos_embed_max_size, 1e-6
    for i in cfg.dual_attention_layers:
        cc.dual_attention[i] = 1
    h = l.mx_mmdit_create(C.byref(cc))
    assert h and l.mx_mmdit_set_weights(h, pw.blob.data_ptr(), pw.blob.numel(), pw.table, len(pw.names)) == 0
    assert l.mx_mmdit_validate(h, 2, 16, 16, 37) == 0, l.mx_last_error()
    l.mx_mmdit_destroy(h)


def test_vae_plan_resolves_packed_weights_on_host():
    """mx_vae_validate walks the decoder's step plan on the host: every packed tensor name / size it asks for is what pack_vae produced;
    the inventory equals the oracle's; a missing tensor is reported."""
    from oracle import vae_ref
    from sduss_amd import lib
    from sduss_amd.vae import VAEConfig, pack_vae
    from sduss_amd import weights
    l = lib.load()
    for cfg, ocfg in ((VAEConfig.tiny(), vae_ref.VAEConfig.tiny()), (VAEConfig.sdxl(), vae_ref.VAEConfig.sdxl())):
        shapes = vae_ref.param_shapes(ocfg)
        P = {k: torch.zeros(v) for k, v in shapes.items()}
        pw = weights.PackedWeights(pack_vae(cfg, P), "cpu")
        cc = lib.VAEConfigC()
        cc.latent_channels, cc.out_channels, cc.n_levels = cfg.latent_channels, cfg.out_channels, len(cfg.block_out_channels)
        for i, v in enumerate(cfg.block_out_channels):
            cc.block_out_channels[i] = v
        cc.layers_per_block, cc.norm_num_groups, cc.norm_eps = cfg.layers_per_block, cfg.norm_num_groups, cfg.norm_eps
        h = l.mx_vae_create(C.byref(cc))
        assert h and l.mx_vae_set_weights(h, pw.blob.data_ptr(), pw.blob.numel(), pw.table, len(pw.names)) == 0
        assert l.mx_vae_validate(h, 1, 16, 16) == 0, l.mx_last_error()
        assert l.mx_vae_workspace_bytes(h, 1, 16, 16) > 0
        bad = [e for e in pack_vae(cfg, P) if e[0] != "decoder.mid_block.attentions.0.to_k.bias"]
        pw2 = weights.PackedWeights(bad, "cpu")
        assert l.mx_vae_set_weights(h, pw2.blob.data_ptr(), pw2.blob.numel(), pw2.table, len(pw2.names)) == 0
        assert l.mx_vae_validate(h, 1, 16, 16) != 0 and b"to_k.bias" in l.mx_last_error()
        l.mx_vae_destroy(h)
    # SDXL VAE decoder: 49.5 M parameters
    total = sum(torch.Size(v).numel() for k, v in vae_ref.param_shapes(vae_ref.VAEConfig.sdxl()).items())
    assert 49.0e6 < total < 50.0e6, total


def test_layernorm_fold_algebra_and_tile_policy():
    """weights.fold_layernorm: LayerNorm(x) W^T + b == rstd (x W'^T - mean colsum) + b' (checked in fp64 with the rounded W' on both sides);
    the host-side tile policy of the fold: launches that carry ln_stats / stats_out never take the 256 x 256 kernel (built without the
    hooks), mx_gemm_stats_slabs / mx_gemm_ln_prefers_pass answer from the same chooser the launch uses."""
    from sduss_amd import lib
    from sduss_amd.weights import fold_layernorm
    g = torch.Generator().manual_seed(3)
    m, c, n = 37, 128, 96
    x = torch.randn(m, c, generator=g, dtype=torch.float64) * 2 + 1.5
    w = torch.randn(n, c, generator=g); bias = torch.randn(n, generator=g); gamma = 1 + 0.2 * torch.randn(c, generator=g); beta = 0.1 * torch.randn(c, generator=g)
    wf, colsum, bf_ = fold_layernorm(w, bias, gamma, beta)
    assert wf.dtype == torch.bfloat16 and colsum.dtype == torch.float32 and bf_.dtype == torch.float32
    mean = x.mean(dim=1, keepdim=True); var = x.var(dim=1, unbiased=False, keepdim=True); rstd = (var + 1e-5).rsqrt()
    w_eff = wf.double() / gamma.double()[None, :]                       # the weight the folded form realises
    want = ((x - mean) * rstd * gamma.double() + beta.double()) @ w_eff.t() + bias.double() + (w.double() - w_eff) @ beta.double()
    got = rstd * (x @ wf.double().t() - mean * colsum.double()[None, :]) + bf_.double()[None, :]
    assert (got - want).abs().max().item() < 1e-5 * want.abs().max().item()

    l = lib.load()
    def desc(m, n, k, flags=0, aligned=True):
        d = lib.GemmDesc()
        d.a, d.w, d.c = 4096, 8192, 16384 if aligned else 16386
        d.M, d.N, d.K, d.lda, d.ldc, d.flags = m, n, k, k, (n // 2 if flags & lib.EPI_GEGLU else n), flags
        return d
    # headline batch: GEGLU and QKV-sized launches would take the 256 x 256 kernel -> normalisation pass; to_q hides the statistics
    assert l.mx_gemm_ln_prefers_pass(C.byref(desc(8192, 10240, 1280, lib.EPI_GEGLU))) == 1
    assert l.mx_gemm_ln_prefers_pass(C.byref(desc(8192, 1280, 1280))) == 0
    assert l.mx_gemm_ln_prefers_pass(C.byref(desc(2048, 3840, 1280))) in (0, 1)
    # producers: one slab per wave column panel (80 columns of a 160-wide tile, 64 of a 128-wide one); none from the generic kernel
    assert l.mx_gemm_stats_slabs(C.byref(desc(8192, 1280, 1280))) == 16
    assert l.mx_gemm_stats_slabs(C.byref(desc(2048, 1280, 5120))) == 20      # (a 5 % split-K gain is below the 25 % margin: unsplit 128 x 128 tiles)
    assert l.mx_gemm_stats_slabs(C.byref(desc(1152, 1280, 5120))) == 16      # round 4: split-K moves a 768 px request's ff.net.2 to 128 x 160 tiles in 3 K slices (80-column panels)
    assert l.mx_gemm_splitk(C.byref(desc(512, 1280, 5120)), 0) >= 2 and l.mx_gemm_splitk(C.byref(desc(2048, 1280, 1280)), 0) == 1
    assert l.mx_gemm_stats_slabs(C.byref(desc(64, 1280, 1280))) == 0             # M < 128: generic kernel
    assert l.mx_gemm_stats_slabs(C.byref(desc(8192, 1280, 1280, lib.EPI_GEGLU))) == 0
    # a shape whose plain launch takes 256 x 256 tiles still yields statistics: asking for them moves it to a 256 / 128-row tile
    assert l.mx_gemm_ln_prefers_pass(C.byref(desc(32768, 5120, 640))) == 1 and l.mx_gemm_stats_slabs(C.byref(desc(32768, 5120, 640))) > 0
    # round 4, the shape queries of the late additions (host only, the same chooser the launch uses):
    # finalised row statistics need a 256-row tile of the producer (headline batch: yes; one request: 128-row tiles, no)
    def producer(m, k):
        d = desc(m, 1280, k); d.residual, d.ldr = 32768, 1280
        return d
    assert l.mx_gemm_ln_final_supported(C.byref(producer(8192, 1280))) == 1 and l.mx_gemm_ln_final_supported(C.byref(producer(8192, 5120))) == 1
    assert l.mx_gemm_ln_final_supported(C.byref(producer(2048, 1280))) == 0 and l.mx_gemm_ln_final_supported(C.byref(desc(8192, 10240, 1280, lib.EPI_GEGLU))) == 0
    # tail split: one 1024 px request's GEGLU is 1.25 rounds of 256 x 256 tiles -> two launches; the headline batch (5 whole rounds) and QKV stay one
    assert l.mx_gemm_launches(C.byref(desc(2048, 10240, 1280, lib.EPI_GEGLU))) == 2 and l.mx_gemm_launches(C.byref(desc(4096, 10240, 1280, lib.EPI_GEGLU))) == 2
    assert l.mx_gemm_launches(C.byref(desc(8192, 10240, 1280, lib.EPI_GEGLU))) == 1 and l.mx_gemm_launches(C.byref(desc(8192, 1280, 1280))) == 1
    # GroupNorm partial sums: a chip-filling conv with the time-embedding row bias can leave them; with a residual, or on 128-row tiles, it cannot
    def conv(b, hw, cin, cout, residual=False):
        d = lib.GemmDesc()
        d.a, d.w, d.c, d.bias, d.rowbias = 4096, 8192, 16384, 32768, 65536
        d.M, d.N, d.K, d.ldc, d.ldr, d.ldrb, d.rows_per_batch = b * hw * hw, cout, 9 * cin, cout, cout, cout, hw * hw
        d.B, d.Hin, d.Win, d.Cin, d.Hout, d.Wout, d.stride = b, hw, hw, cin, hw, hw, 1
        if residual: d.residual = 131072
        return d
    assert l.mx_gemm_gn_partials_supported(C.byref(conv(8, 32, 1280, 1280)), 1) == 1 and l.mx_gemm_gn_partials_supported(C.byref(conv(8, 128, 320, 320)), 1) == 1
    assert l.mx_gemm_gn_partials_supported(C.byref(conv(8, 32, 1280, 1280, residual=True)), 1) == 0 and l.mx_gemm_gn_partials_supported(C.byref(conv(1, 32, 1280, 1280)), 1) == 0
    # round 5: which kernel family serves a launch (mx_gemm_form): the M <= 16 weight stream, conv_out's small-N kernel, and the tile families
    GENERIC, T256, T128, P256, SMALL_M, SMALL_N = range(6)
    form = lambda d, c=0: l.mx_gemm_form(C.byref(d), c)
    assert form(desc(8, 1280, 2816, lib.EPI_SILU)) == SMALL_M and form(desc(16, 13760, 1280, lib.EPI_OUT_F32)) == SMALL_M and form(desc(1, 64, 64)) == SMALL_M
    assert form(desc(17, 1280, 2816)) == GENERIC                      # one row past the form's reach
    assert form(desc(8, 442368, 1536, lib.EPI_OUT_F32)) == SMALL_M    # the MMDiT's stacked AdaLN modulation (1.4 GB of weights): the long-stream form (activations staged in LDS)
    assert form(desc(16, 65536, 4096, lib.EPI_OUT_F32)) == GENERIC    # a long stream whose M x K activations exceed 64 KB of LDS: the tile kernel
    assert form(desc(8, 1280, 1280, lib.EPI_GEGLU)) == GENERIC and form(desc(8, 1288, 1280)) == GENERIC      # gated epilogue / N % 16 != 0
    assert form(desc(8192, 1280, 1280)) == T256 and form(desc(2048, 1280, 1280)) == T128 and form(desc(8192, 10240, 1280, lib.EPI_GEGLU)) == P256
    def conv_out(b, hw, cin, cout, **kw):
        d = conv(b, hw, cin, cout)
        d.rowbias, d.ldrb, d.rows_per_batch = None, 0, 0
        for k, v in kw.items(): setattr(d, k, v)
        return d
    assert form(conv_out(8, 128, 320, 4), 1) == SMALL_N and form(conv_out(2, 128, 320, 4, corner_patch=32), 1) == SMALL_N
    assert form(conv_out(8, 128, 320, 16), 1) != SMALL_N               # 16 x 2880 weights + the staged chunk exceed 64 KB of LDS
    assert form(conv_out(8, 128, 320, 4, stride=2, Hout=64, Wout=64, M=8 * 64 * 64), 1) != SMALL_N and form(conv(8, 128, 320, 4), 1) != SMALL_N      # stride 2; a row bias
    assert form(conv_out(8, 128, 320, 320), 1) == T256
    assert form(conv_out(8, 128, 64, 320, cin_valid=8), 1) == 6 and form(conv_out(8, 128, 64, 320), 1) == T256      # conv_in: only when the caller names the valid channels
    assert form(conv_out(8, 128, 64, 128, cin_valid=8), 1) != 6 and form(conv_out(8, 128, 64, 320, cin_valid=16), 1) != 6                  # N % 80; more than 8 channels


def test_clip_plan_resolves_packed_weights_on_host():
    """mx_clip_validate walks the text-encoder plan on the host against what pack_clip produces from a transformers-named state dict"""
    from sduss_amd import lib, weights
    from sduss_amd.clip import CLIPTextConfig, pack_clip
    l = lib.load()
    for cfg in (CLIPTextConfig.tiny(), CLIPTextConfig.tiny(projection_dim=64, hidden_act="gelu")):
        h, i = cfg.hidden_size, cfg.intermediate_size
        P = {"text_model.embeddings.token_embedding.weight": torch.zeros(cfg.vocab_size, h), "text_model.embeddings.position_embedding.weight": torch.zeros(77, h),
             "text_model.final_layer_norm.weight": torch.zeros(h), "text_model.final_layer_norm.bias": torch.zeros(h)}
        for k in range(cfg.num_hidden_layers):
            p = f"text_model.encoder.layers.{k}"
            for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
                P[f"{p}.self_attn.{n}.weight"] = torch.zeros(h, h); P[f"{p}.self_attn.{n}.bias"] = torch.zeros(h)
            for n in ("layer_norm1", "layer_norm2"):
                P[f"{p}.{n}.weight"] = torch.zeros(h); P[f"{p}.{n}.bias"] = torch.zeros(h)
            P[f"{p}.mlp.fc1.weight"] = torch.zeros(i, h); P[f"{p}.mlp.fc1.bias"] = torch.zeros(i)
            P[f"{p}.mlp.fc2.weight"] = torch.zeros(h, i); P[f"{p}.mlp.fc2.bias"] = torch.zeros(h)
        if cfg.projection_dim:
            P["text_projection.weight"] = torch.zeros(cfg.projection_dim, h)
        pw = weights.PackedWeights(pack_clip(cfg, P), "cpu")
        cc = lib.CLIPConfigC()
        cc.vocab_size, cc.hidden_size, cc.intermediate_size, cc.num_hidden_layers = cfg.vocab_size, h, i, cfg.num_hidden_layers
        cc.num_attention_heads, cc.max_position_embeddings, cc.hidden_act = cfg.num_attention_heads, 77, 0 if cfg.hidden_act == "quick_gelu" else 1
        cc.projection_dim, cc.eos_token_id, cc.hidden_layer, cc.layer_norm_eps = cfg.projection_dim, 2, -2, 1e-5
        hnd = l.mx_clip_create(C.byref(cc))
        assert hnd and l.mx_clip_set_weights(hnd, pw.blob.data_ptr(), pw.blob.numel(), pw.table, len(pw.names)) == 0
        assert l.mx_clip_validate(hnd, 2) == 0, l.mx_last_error()
        assert l.mx_clip_workspace_bytes(hnd, 2) > 0
        l.mx_clip_destroy(hnd)
    cc.num_attention_heads = 3                                   # heads of 64 only
    assert not l.mx_clip_create(C.byref(cc)) and b"bad config" in l.mx_last_error()


def test_t5_plan_resolves_packed_weights_on_host():
    """mx_t5_validate walks the T5 encoder plan on the host against what pack_t5 produces from a transformers-named state dict; the position
    bias table follows the published bucket rule (checked against transformers' own function)"""
    from sduss_amd import lib, weights
    from sduss_amd.t5 import T5Config, pack_t5, position_bias, relative_position_bucket
    cfg = T5Config.tiny()
    d, f, inner = cfg.d_model, cfg.d_ff, cfg.num_heads * 64
    P = {"shared.weight": torch.zeros(cfg.vocab_size, d), "encoder.final_layer_norm.weight": torch.ones(d),
         "encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight": torch.arange(32 * cfg.num_heads, dtype=torch.float32).reshape(32, cfg.num_heads)}
    for k in range(cfg.num_layers):
        a = f"encoder.block.{k}.layer.0"
        for n in ("q", "k", "v"):
            P[f"{a}.SelfAttention.{n}.weight"] = torch.zeros(inner, d)
        P[f"{a}.SelfAttention.o.weight"] = torch.zeros(d, inner); P[f"{a}.layer_norm.weight"] = torch.ones(d)
        m = f"encoder.block.{k}.layer.1"
        P[f"{m}.DenseReluDense.wi_0.weight"] = torch.zeros(f, d); P[f"{m}.DenseReluDense.wi_1.weight"] = torch.zeros(f, d)
        P[f"{m}.DenseReluDense.wo.weight"] = torch.zeros(d, f); P[f"{m}.layer_norm.weight"] = torch.ones(d)
    pw = weights.PackedWeights(pack_t5(cfg, P, (256, 64)), "cpu")
    l = lib.load()
    cc = lib.T5ConfigC()
    cc.vocab_size, cc.d_model, cc.d_ff, cc.num_layers, cc.num_heads, cc.layer_norm_epsilon = cfg.vocab_size, d, f, cfg.num_layers, cfg.num_heads, 1e-6
    h = l.mx_t5_create(C.byref(cc))
    assert h and l.mx_t5_set_weights(h, pw.blob.data_ptr(), pw.blob.numel(), pw.table, len(pw.names)) == 0
    assert l.mx_t5_validate(h, 2, 256) == 0, l.mx_last_error()
    assert l.mx_t5_validate(h, 1, 64) == 0, l.mx_last_error()
    assert l.mx_t5_validate(h, 1, 128) != 0 and b"position_bias.128" in l.mx_last_error()       # a length that was not prepared
    assert l.mx_t5_workspace_bytes(h, 2, 256) > 0
    l.mx_t5_destroy(h)
    from transformers.models.t5 import modeling_t5
    pos = torch.arange(300); rel = pos[None, :] - pos[:, None]
    assert torch.equal(relative_position_bucket(rel, 32, 128), modeling_t5.T5Attention._relative_position_bucket(rel, bidirectional=True, num_buckets=32, max_distance=128))
    pb = position_bias(cfg, P["encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight"], 100)
    assert pb.shape == (cfg.num_heads, 100, 128) and float(pb[:, :, 100:].abs().max()) == 0.0
    assert abs(float(pb[1, 10, 10]) - 1.4426950408889634 * 1.0) < 1e-5          # bucket 0 (offset 0), head 1 -> weight[0, 1] = 1


def test_block_skip_decisions_follow_the_reference_counters():
    """cache_manager.py:134-136: a sample that reused a block four times in a row runs it; the counter resets on a run."""
    from sduss_amd.block_cache import BlockSkipCache, ThresholdPredictor, decide, MSE_UNCACHED
    run, prev = decide(np.array([0, 0, 1, 0]), np.array([4, 3, 2, 0]))
    assert run.tolist() == [True, False, True, False] and prev.tolist() == [0, 4, 0, 1]
    # five steps of "never run": reuse x4 then the forced run
    prev = np.zeros(1, dtype=np.int64)
    seen = []
    for _ in range(11):
        run, prev = decide(np.array([0]), prev)
        seen.append(bool(run[0]))
    assert seen == [False] * 4 + [True] + [False] * 4 + [True, False]
    # the callback builds the reference's feature rows and treats an uncached sample as a fresh counter
    rows = []

    class Spy:
        def predict(self, f):
            rows.append(np.array(f)); return np.zeros(len(f))
    bc = BlockSkipCache(Spy())
    fp = C.POINTER(C.c_float)                      # what the library hands to the callback
    ts = C.cast((C.c_float * 2)(981.0, 961.0), fp)
    mse = C.cast((C.c_float * 8)(*([0.5, 0.1, 0.2, 0.3] + [MSE_UNCACHED] * 4)), fp)
    out = (C.c_ubyte * 2)(9, 9)
    assert bc._predict(None, 5, 1, 2, 4, ts, mse, out) == 0
    assert rows[0].shape == (2, 6) and rows[0][0].tolist()[:3] == [5.0, 981.0, 0.5] and list(out) == [0, 0]
    assert ThresholdPredictor(0.25).predict(rows[0]).tolist() == [1, 1]
    assert ThresholdPredictor(0.6).predict(rows[0][:1]).tolist() == [0]

    class Broken:
        def predict(self, f):
            raise RuntimeError("no model")
    bc = BlockSkipCache(Broken())
    assert bc._predict(None, 0, 0, 2, 1, ts, mse, out) == 1 and isinstance(bc.error, RuntimeError)


def test_block_cache_state_size_is_the_sum_of_block_inputs_and_outputs(tiny):
    """mx_unet_block_cache_bytes walks the plan without launches: it must equal the inputs and outputs of the seven blocks."""
    from sduss_amd import config, lib
    l = lib.load()
    pcfg = config.UNetConfig.tiny()
    cc = lib.UNetConfigC()
    cc.in_channels, cc.out_channels, cc.n_levels, cc.layers_per_block = 4, 4, 3, 2
    for i, v in enumerate(pcfg.block_out_channels):
        cc.block_out_channels[i] = v; cc.down_has_attn[i] = int(pcfg.down_has_attn[i])
        cc.transformer_layers[i] = pcfg.transformer_layers_per_block[i]; cc.num_heads[i] = pcfg.num_heads[i]
    cc.cross_attention_dim, cc.addition_time_embed_dim = pcfg.cross_attention_dim, pcfg.addition_time_embed_dim
    cc.projection_class_embeddings_input_dim, cc.norm_num_groups = pcfg.projection_class_embeddings_input_dim, 32
    h = l.mx_unet_create(C.byref(cc))
    B, H = 3, 24
    r256 = lambda n: (n + 255) // 256 * 256
    ten = lambda hw, ch: r256(B * hw * hw * ch * 2)
    ch = list(pcfg.block_out_channels)
    want = r256(4 * B * 64 * 8 + 2 * B * 4)        # comparison scratch + the slot table + the selection table of a partially reused block
    hw, cin, skips = H, ch[0], [(H, ch[0])]
    for i in range(3):
        want += ten(hw, cin)
        for _ in range(2):
            want += ten(hw, ch[i]); skips.append((hw, ch[i]))
        if i != 2:
            hw //= 2; want += ten(hw, ch[i]); skips.append((hw, ch[i]))
        cin = ch[i]
    want += 2 * ten(hw, cin)
    for i in range(3):
        want += ten(hw, cin) + sum(ten(*skips[-1 - k]) for k in range(3))
        del skips[-3:]
        cin = ch[2 - i]
        if i != 2:
            hw *= 2
        want += ten(hw, cin)
    assert l.mx_unet_block_cache_bytes(h, B, H, H) == want + 256
    assert l.mx_unet_block_cache_bytes(h, B, 25, 24) == 0
    l.mx_unet_destroy(h)


def test_mmdit_block_cache_state_size():
    from sduss_amd import config, lib
    from sduss_amd.transformer_sd3 import mmdit_config_c
    l = lib.load()
    pcfg = config.MMDiTConfig.tiny()
    h = l.mx_mmdit_create(C.byref(mmdit_config_c(pcfg)))
    assert h
    B, H, Lt = 3, 24, 77
    d, L = pcfg.num_attention_heads * 64, (H // pcfg.patch_size) ** 2
    r256 = lambda n: (n + 255) // 256 * 256
    want = r256(B * 64 * 8 + B * 4) + pcfg.num_layers * (2 * r256(B * L * d * 2) + r256(B * Lt * d * 2))
    assert l.mx_mmdit_block_cache_bytes(h, B, H, H, Lt) == want
    assert l.mx_mmdit_block_cache_bytes(h, B, H + 1, H, Lt) == 0
    l.mx_mmdit_destroy(h)


def test_compiled_forest_equals_sklearn_predict():
    """CompiledForest / mx_forest_predict: the same answers as RandomForestClassifier.predict on the predictor's feature rows (block index,
    timestep, input differences spanning 1e-6 .. the uncached marker)"""
    from sklearn.ensemble import RandomForestClassifier
    from sduss_amd.block_cache import CompiledForest, MSE_UNCACHED
    rng = np.random.default_rng(0)
    for n_feat in (3, 6):
        X = np.column_stack([rng.integers(0, 7, 4000).astype(np.float64), rng.uniform(0, 1000, 4000)] +
                            [10.0 ** rng.uniform(-6, 1, 4000) for _ in range(n_feat - 2)])
        y = ((X[:, 2:].max(axis=1) > 0.01 * (1 + X[:, 0])) ^ (rng.uniform(size=4000) < 0.05)).astype(np.int64)
        rf = RandomForestClassifier(n_estimators=32, max_depth=8, random_state=0).fit(X, y)
        cf = CompiledForest(rf)
        T = np.column_stack([rng.integers(0, 7, 2000).astype(np.float64), rng.uniform(0, 1000, 2000)] +
                            [10.0 ** rng.uniform(-7, 2, 2000) for _ in range(n_feat - 2)])
        T[:50, 2:] = MSE_UNCACHED
        T[50:100, 2] = 0.0
        assert np.array_equal(cf.predict(T), rf.predict(T))
    with pytest.raises(AssertionError):
        cf.predict(np.zeros((2, 4)))


def test_block_skip_bookkeeping_equals_the_reference_restatement():
    """BlockSkipCache._predict against oracle/cache_ref.py (cache_manager.py:101-161 line by line) on random predictor answers: same feature
    rows, same run masks, step after step, for a fixed batch composition (the product keys its state by the composition; a change refills it,
    where the reference keeps the ids that stay) -- down-type and up-type blocks, both forced-run settings."""
    from oracle.cache_ref import CacheManagerRef, MAX
    from sduss_amd.block_cache import BlockSkipCache, MSE_UNCACHED
    assert np.float32(MAX) == np.float32(MSE_UNCACHED)

    class Scripted:
        def __init__(self, rng):
            self.rng, self.last = rng, None

        def predict(self, f):
            self.last = np.array(f, dtype=np.float64)
            return (self.rng.uniform(size=len(f)) < 0.35).astype(np.int64)
    fp = C.POINTER(C.c_float)
    for forced in (4, 2):
        for nres in (0, 3):
            rng_a, rng_b, data = np.random.default_rng(7), np.random.default_rng(7), np.random.default_rng(1)
            pa, pb = Scripted(rng_a), Scripted(rng_b)
            refm = CacheManagerRef(pa, forced_after=forced)
            bc = BlockSkipCache(pb, forced_after=forced)
            ids = ["r3", "r9", "r4"]
            n, block = len(ids), 5
            for step in range(40):
                ts = data.uniform(0, 1000, n).astype(np.float32)
                mse = data.uniform(0, 2, (n, 1 + nres)).astype(np.float32)
                want, feat = refm.get_mask(ids, mse[:, 0], block, ts, res_mse=mse[:, 1:] if nres else None)
                sent = mse.copy() if step > 0 else np.full_like(mse, MSE_UNCACHED)       # what the library sends for an uncached block
                out = (C.c_ubyte * n)()
                assert bc._predict(None, block, int(nres > 0), n, 1 + nres, C.cast(ts.ctypes.data, fp), C.cast(sent.ctypes.data, fp), out) == 0
                assert np.array_equal(np.array(list(out)) > 0, want), (forced, nres, step)
                assert np.allclose(pb.last, feat.astype(np.float32).astype(np.float64), rtol=1e-6), (forced, nres, step)


def test_block_cache_slot_table_follows_the_requests(tiny):
    """BlockSkipCache._bind_rows on the host (state tensor on the CPU, sizes from the library's dry walk): distinct rows, a request keeps its row
    while it stays, is forgotten when it leaves (cache_manager.py:131: the dictionaries are rebuilt from the ids of each call), a returning or new
    request starts invalid, the counters follow the ids, a larger batch grows the state and forgets everything, another latent size too."""
    from sduss_amd import config, lib
    from sduss_amd.block_cache import BlockSkipCache
    l = lib.load()
    pcfg = config.UNetConfig.tiny()
    cc = lib.UNetConfigC()
    cc.in_channels, cc.out_channels, cc.n_levels, cc.layers_per_block = 4, 4, 3, 2
    for i, v in enumerate(pcfg.block_out_channels):
        cc.block_out_channels[i] = v; cc.down_has_attn[i] = int(pcfg.down_has_attn[i])
        cc.transformer_layers[i] = pcfg.transformer_layers_per_block[i]; cc.num_heads[i] = pcfg.num_heads[i]
    cc.cross_attention_dim, cc.addition_time_embed_dim = pcfg.cross_attention_dim, pcfg.addition_time_embed_dim
    cc.projection_class_embeddings_input_dim, cc.norm_num_groups = pcfg.projection_class_embeddings_input_dim, 32
    h = l.mx_unet_create(C.byref(cc))

    class Model:
        _lib, _handle, device = l, h, torch.device("cpu")

    class Zero:
        def predict(self, f):
            return np.zeros(len(f))
    bc = BlockSkipCache(Zero(), forced_after=3)

    def bind(ids, hw=16, commit=True):
        bc.bind(Model, len(ids), hw, hw, 0, row_ids=ids)
        out = list(bc._slots_arr), list(bc._valid_arr)
        if commit:
            bc.after_forward()                                  # the forward succeeded: the staged request -> row table becomes the table
        return out
    s1, v1 = bind(["a", "b", "c"])
    assert len(set(s1)) == 3 and v1 == [0, 0, 0] and bc.desc.n_slots == 8
    assert bc.state.numel() == l.mx_unet_block_cache_bytes(h, 8, 16, 16)
    bc.previous = {0: {"a": 1, "b": 2, "c": 0}}
    s2, v2 = bind(["c", "a"])                                   # b left; order changed
    assert s2 == [s1[2], s1[0]] and v2 == [1, 1] and bc.previous[0] == {"a": 1, "c": 0}
    s3, v3 = bind(["a", "b", "d"])                              # b comes back (forgotten), d is new, c left
    assert s3[0] == s1[0] and v3 == [1, 0, 0] and len(set(s3)) == 3 and bc.previous[0] == {"a": 1}
    state_before = bc.state
    s4, v4 = bind([f"r{i}" for i in range(9)])                  # more requests than rows: the state grows, nothing cached survives
    assert bc.desc.n_slots == 18 and v4 == [0] * 9 and bc.state is not state_before and bc.previous == {}
    s5, v5 = bind(["r0", "r1"], hw=24)                          # another latent size
    assert v5 == [0, 0] and bc.desc.n_slots == 8
    with pytest.raises(AssertionError):
        bind(["x", "x"])
    # a forward that fails part-way never reaches after_forward(): the requests it introduced must NOT be valid at the next bind (their rows
    # were not stored for the later blocks), and because some blocks' rows were overwritten and others not, nothing else survives either
    s6, v6 = bind(["r0", "r1", "n0"], hw=24)
    assert v6 == [1, 1, 0]
    s7, v7 = bind(["r0", "r1", "n0", "n1"], hw=24, commit=False)   # the forward of this bind "fails"
    assert v7 == [1, 1, 1, 0]
    s8, v8 = bind(["r0", "n1"], hw=24)
    assert v8 == [0, 0] and bc.previous == {}
    s9, v9 = bind(["r0", "n1"], hw=24)
    assert v9 == [1, 1] and s9 == s8
    # without row ids the descriptor goes back to the batch-composition rule
    bc.bind(Model, 2, 16, 16, 5)
    assert not bc.desc.slots and bc.desc.n_slots == 0 and bc.desc.batch_key == 5
    l.mx_unet_destroy(h)


def test_oracle_inventories_equal_the_published_checkpoints():
    """The wiring pin that needs no weights: the oracle's parameter inventory, built from the HF config values alone, must be the published
    SDXL-base-1.0 UNet (1 680 tensors, 2 567 463 684 parameters: the "2.6 B UNet" of the SDXL report) and the published SD3.5-medium
    transformer (909 tensors, 2 469 663 936 parameters: "2.5 B").  A mis-wired block (a missing shortcut conv, a wrong transformer depth,
    a GEGLU of the wrong width, a dual-attention block too many) changes these numbers."""
    import math
    from oracle import sd3_mmdit_ref
    sh = ref.param_shapes(ref.UNetConfig.sdxl_base())
    assert len(sh) == 1680 and sum(math.prod(v) for v in sh.values()) == 2_567_463_684
    sh3 = sd3_mmdit_ref.param_shapes(sd3_mmdit_ref.MMDiTConfig.sd35_medium())
    assert len(sh3) == 909 and sum(math.prod(v) for v in sh3.values()) == 2_469_663_936
    # per-family sub-totals of the SDXL UNet (they are what the FLOP split of SURVEY 8d is derived from)
    conv = sum(math.prod(v) for k, v in sh.items() if len(v) == 4)
    attn = sum(math.prod(v) for k, v in sh.items() if ".attn1." in k or ".attn2." in k)
    ff = sum(math.prod(v) for k, v in sh.items() if ".ff.net." in k)
    assert conv + attn + ff > 0.97 * 2_567_463_684 and ff > attn > 0.2 * ff


def test_oracle_chain_batched_steps_equal_each_request_alone(tiny):
    """oracle/chain_ref.denoising_step on a batch of requests at DIFFERENT step indices with DIFFERENT step counts (continuous batching,
    scheduling_euler_discrete.py:171-175, 213-217) must give every request exactly what it gets when stepped alone: samples are independent,
    the per-request sigma / timestep rows are the only coupling and they are per-row."""
    from oracle import chain_ref
    cfg, P = tiny
    g = torch.Generator().manual_seed(3)

    def make(rid, steps, hw):
        ts, sig, init = scheduler_ref.sdxl_euler_tables(steps)
        pe, ne = torch.randn(1, 77, cfg.cross_attention_dim, generator=g), torch.randn(1, 77, cfg.cross_attention_dim, generator=g)
        pp, npp = torch.randn(1, cfg.text_embed_dim, generator=g), torch.randn(1, cfg.text_embed_dim, generator=g)
        tid = torch.tensor([[hw * 8.0, hw * 8.0, 0.0, 0.0, hw * 8.0, hw * 8.0]])
        lat = torch.randn(1, cfg.in_channels, hw, hw, generator=g) * init
        return chain_ref.ChainRequest(rid, hw * 8, steps, lat, (pe, pp, tid), (ne, npp, tid), ts, sig)

    import copy
    model = lambda x, t, e, te, ti: ref.unet_forward(P, cfg, x, t, e, te, ti)
    a = [make(0, 6, 16), make(1, 9, 16), make(2, 7, 8)]
    alone = copy.deepcopy(a)
    joins = [0, 2, 1]
    with torch.inference_mode():
        gstep = 0
        while not all(r.done() for r in a):
            act = [r for i, r in enumerate(a) if joins[i] <= gstep and not r.done()]
            d = {}
            for r in act:
                d.setdefault(str(r.resolution), []).append(r)
            chain_ref.denoising_step(d, model, "sdxl", 5.0, store_dtype=None)
            gstep += 1
        for r in alone:
            while not r.done():
                chain_ref.denoising_step({str(r.resolution): [r]}, model, "sdxl", 5.0, store_dtype=None)
    assert gstep == 11                       # request 1 joins at 2 and takes 9 steps
    for x, y in zip(a, alone):
        assert x.step_index == y.step_index == x.num_inference_steps
        assert torch.allclose(x.latents, y.latents, atol=2e-3, rtol=0), (x.latents - y.latents).abs().max()   # fp32 throughout (store_dtype None): batching changes only the summation order inside the BLAS calls
    # the flow-match tables restated for the chain equal the step mirror's
    from sduss_amd.pipeline_sd3 import flow_match_tables
    for n in (20, 28, 50):
        ts, sig = chain_ref.sd3_flow_tables(n)
        pts, psig = flow_match_tables(n)
        assert np.array_equal(ts.numpy(), pts) and np.array_equal(sig.numpy(), psig)


def test_host_side_plans_under_address_and_ub_sanitizers():
    """SURVEY.md section 5 (race detection / sanitizers): GPU sanitizers are not available on this pool, so the sanitizer covers what runs on
    the host -- the step plans' arena / cursor / offset arithmetic.  `make asan` compiles every source --cuda-host-only with AddressSanitizer +
    UBSan and links tests/asan_walk.cpp, which walks the dry-run entry points (workspace / state sizing at the benchmark's sizes, mixed
    groups, patch-parallel comm plans at world 2 / 4 / 8, block-cache state, grouped-GEMM tile bookkeeping) without launching a kernel."""
    import subprocess
    csrc = os.path.join(ROOT, "sduss_amd", "csrc")
    r = subprocess.run(["make", "-C", csrc, "-j4", "asan"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    exe = os.path.join(ROOT, "build", "asan", "asan_walk")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0"))
    assert r.returncode == 0 and "ASAN_WALK_OK" in r.stdout, (r.stdout[-2000:] + r.stderr[-4000:])


def test_slo_tables_for_mi355x_have_the_reference_formats():
    """SURVEY 8f rank 3, second half: profiles/esymred_mi355x.json must load the way sduss/worker/scheduler/esymred_utils.py:14-43 loads
    configs/esymred.json (STANDALONE denoising / postprocessing per model and resolution -> deadlines = (denoising [+ postprocessing]) * SLO),
    and profiles/exec_time_mi355x/sm_util_<model>_<res>.csv the way policy/ESyMReD.py:105-118 reads the last column ("post time" per batch size)."""
    import json
    data = json.load(open(os.path.join(ROOT, "profiles", "esymred_mi355x.json")))
    hp, standalone, discard = data["Hyper_Parameter"], data["STANDALONE"], data["DISCARD_SLACK"]
    assert discard == 500 and hp["postprocessing_ratio"] == 0.9 and set(standalone) == {"sdxl", "sd3"}
    slo = 5
    for model in standalone:
        den, post = standalone[model]["denoising"], standalone[model]["postprocessing"]
        assert list(den) == list(post) == ["512", "768", "1024"]
        ddl_post = {r: (den[r] + post[r]) * slo for r in den}          # esymred_utils.py:27-35
        ddl_den = {r: den[r] * slo for r in den}                        # :37-43
        assert 0 < ddl_den["512"] < ddl_den["768"] < ddl_den["1024"] < ddl_post["1024"]
        assert den["1024"] < 3.7 if model == "sdxl" else den["1024"] < 5.92      # faster than the H100 values the reference ships (esymred.json:24-38)
        for res in (512, 768, 1024):
            path = os.path.join(ROOT, "profiles", "exec_time_mi355x", f"sm_util_{model}_{res}.csv")
            rows = []
            with open(path) as f:
                first = True
                for line in f:                                          # ESyMReD.py:109-118
                    if first:
                        first = False
                        continue
                    rows.append(float(line.strip().split(",")[-1]))
            assert len(rows) == 8 and all(b > a > 0 for a, b in zip(rows, rows[1:])), (model, res, rows)
            unet = [float(l.split(",")[1]) for l in open(path).read().strip().splitlines()[1:]]
            assert abs(unet[0] - den[str(res)]) < 0.02 * den[str(res)] + 1e-3         # batch-1 row == the STANDALONE entry


def test_patch_unit_cache_sizing_walks_on_host(tiny):
    """mx_unet_patch_cache_bytes / mx_unet_workspace_bytes_cached_mixed walk the step plan in its patch-unit cache mode with no launches
    (the mixed-resolution batch of the GPU test: one 128 px and two 256 px requests, 64-px patches): the state holds every cached op's tensor
    per request row -- far more than the per-sample block cache -- and the workspace holds the compact patch batches."""
    from sduss_amd import config, lib
    l = lib.load()
    pcfg = config.UNetConfig.tiny()
    cc = lib.UNetConfigC()
    cc.in_channels, cc.out_channels, cc.n_levels, cc.layers_per_block = 4, 4, 3, 2
    for i, v in enumerate(pcfg.block_out_channels):
        cc.block_out_channels[i] = v; cc.down_has_attn[i] = int(pcfg.down_has_attn[i])
        cc.transformer_layers[i] = pcfg.transformer_layers_per_block[i]; cc.num_heads[i] = pcfg.num_heads[i]
    cc.cross_attention_dim, cc.addition_time_embed_dim = pcfg.cross_attention_dim, pcfg.addition_time_embed_dim
    cc.projection_class_embeddings_input_dim, cc.norm_num_groups = pcfg.projection_class_embeddings_input_dim, 32
    h = l.mx_unet_create(C.byref(cc))
    assert h
    state = l.mx_unet_patch_cache_bytes(h, 8, 32, 32, 8)
    assert state > 0, l.mx_last_error()
    assert state > l.mx_unet_block_cache_bytes(h, 8, 32, 32)
    assert l.mx_unet_patch_cache_bytes(h, 16, 32, 32, 8) > 1.9 * state - (1 << 20)       # linear in the request rows
    assert l.mx_unet_patch_cache_bytes(h, 8, 32, 32, 5) == 0                               # rows must be whole patches
    groups = (lib.UNetGroup * 2)()
    groups[0].batch, groups[0].H, groups[0].W = 1, 16, 16
    groups[1].batch, groups[1].H, groups[1].W = 2, 32, 32
    ws = l.mx_unet_workspace_bytes_cached_mixed(h, groups, 2, 77, 8)
    assert ws > 0, l.mx_last_error()
    assert ws >= l.mx_unet_workspace_bytes_mixed(h, groups, 2, 77)
    assert l.mx_unet_workspace_bytes_cached_mixed(h, groups, 2, 77, 0) == 0 and b"is_sliced" in l.mx_last_error()
    l.mx_unet_destroy(h)


def test_attn_tail_capability_query_and_patch_skip_counters_on_host():
    """Round 5, host-only parts.  (1) mx_attn_tail_supported is a pure host query: yes for the two transformer widths of the headline batch (and only where the
    three linears take 256 x 160 tiles and a panel of 256 rows lies inside one sample), the sync buffer is sized per 256-row panel, and the step plans do not prefer
    the chained launch unless MX_ATTN_TAIL=1 (measured 2 % slower per step: DESIGN.md section 4).  (2) PatchSkipCache keeps its per-patch reuse counters as arrays
    aligned with the forward's patch order: they follow the patches through a change of composition exactly as the reference's per-key dictionaries do
    (cache_manager.py:128-131, 150-153)."""
    import ctypes as C
    import numpy as np
    from sduss_amd import lib
    from sduss_amd.block_cache import PatchSkipCache, ThresholdPredictor
    l = lib.load()

    def desc(b, heads, L):
        c, m = heads * 64, b * L
        td = lib.AttnTailDesc()
        for d, a, w, out, res in ((td.out1, 0x10000, 0x20000, 0x30000, 0x30000), (td.to_q, 0x30000, 0x40000, 0x50000, 0), (td.out2, 0x60000, 0x70000, 0x30000, 0x30000)):
            d.a, d.w, d.c, d.bias, d.M, d.N, d.K, d.lda, d.ldc = a, w, out, 0x5000, m, c, c, c, c
            if res:
                d.residual, d.ldr = res, c
        td.out1.stats_out = 0x80000
        td.to_q.ln_stats, td.to_q.ln_colsum, td.to_q.ln_slabs, td.to_q.ln_eps = 0x80000, 0x90000, l.mx_gemm_stats_slabs(C.byref(td.out1)), 1e-5
        td.out2.stats_out = 0xa0000
        td.k, td.ldk, td.vt, td.ldvt, td.vt_batch_stride = 0xb0000, c, 0xc0000, 80, c * 80
        td.B, td.heads, td.L, td.ctx_len, td.sync = b, heads, L, 77, 0xd0000
        return td
    assert l.mx_attn_tail_supported(C.byref(desc(8, 20, 1024))) == 1 and l.mx_attn_tail_supported(C.byref(desc(8, 10, 4096))) == 1
    assert l.mx_attn_tail_supported(C.byref(desc(2, 20, 1024))) == 0          # one request: 128-row tiles
    assert l.mx_attn_tail_supported(C.byref(desc(8, 20, 384))) == 0           # a 256-row panel would straddle two samples
    bad = desc(8, 20, 1024); bad.out2.stats_out = bad.out1.stats_out
    assert l.mx_attn_tail_supported(C.byref(bad)) == 0                        # the two statistics buffers must differ
    assert l.mx_attn_tail_sync_bytes(8192) >= (8 * 32 + 32 + 32 * 4) * 4 and l.mx_attn_tail_sync_bytes(0) == 0
    assert l.mx_attn_tail_preferred() == (1 if os.environ.get("MX_ATTN_TAIL") == "1" else 0)

    pc = PatchSkipCache(ThresholdPredictor(0.5), forced_after=4)
    pc._keys = ["a-0-0", "a-0-1", "b-0-0"]
    pc.previous = {3: {"a-0-0": 2, "b-0-0": 4, "gone": 1}}
    assert pc.previous == {3: {"a-0-0": 2, "a-0-1": 0, "b-0-0": 4}}
    n = 3
    ts = (C.c_float * n)(500.0, 500.0, 500.0)
    mse = (C.c_float * n)(0.1, 0.9, 0.1)                                      # patch 1 exceeds the threshold; patch 2 has reused four times: forced
    out = (C.c_ubyte * n)()
    ptr = lambda a, t: C.cast(a, C.POINTER(t))             # the library hands the callback plain pointers
    rc = pc._predict(None, 3, 0, n, 1, ptr(ts, C.c_float), ptr(mse, C.c_float), ptr(out, C.c_ubyte))
    assert rc == 0 and pc.error is None, repr(pc.error)
    assert list(out) == [0, 1, 1]
    assert pc.previous[3] == {"a-0-0": 3, "a-0-1": 0, "b-0-0": 0}
