"""Weight loading and packing for the MI355X step plan.

The reference rebuilds every Conv2d/Linear/GroupNorm as fp16 ``Split*`` modules and fuses to_k/to_v into one
``to_kv`` at load (sduss/model_executor/modules/unet.py:31-100, attention.py:33-49).  Here the HF state dict is
packed ONCE into a single device blob in the layouts the gfx950 kernels read:

  conv  O,I,3,3        -> bf16 [O, 9*I]  tap-major (k = (ky*3+kx)*I + c)   NHWC implicit GEMM; conv_in's I padded to 64
  linear [N, K]        -> bf16 [N, K]    as is
  attn1 to_q/to_k/to_v -> bf16 [3C, C]   one fused QKV GEMM
  attn2 to_k/to_v      -> bf16 [L*2C, ctx] for ALL L cross-attention layers of one width, [K_l ; V_l] per layer
  ff.net.0.proj (GEGLU)-> rows interleaved in groups of 32 hidden | 32 gate so that the GEMM epilogue finds the
                          pair in the same lane (gemm_bf16.hip)
  time_emb_proj        -> bf16 [sum C_out, T] for all resnets in execution order (one GEMM per step)
  norm1 / norm2 / norm3 of a BasicTransformerBlock -> folded into attn1.to_qkv / attn2.to_q / ff.net.0.proj (fold_layernorm):
                          weight * gamma (bf16), ".colsum" (fp32 row sums of the rounded product), ".bias" = bias + W beta (fp32)
  biases / GroupNorm affine -> fp32
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional, Tuple

import torch

from . import lib as _lib
from .config import UNetConfig, param_shapes, resnet_names, transformer_names

_ALIGN = 256
CONV_IN_PAD = 64


def synthetic_params(cfg: UNetConfig, device="cpu", seed: int = 10086) -> Dict[str, torch.Tensor]:
    """Random-init weights of the architecture (no checkpoint on the box): N(0,1)*fan_in^-1/2 matrices,
    gains ~1, small biases; seed 10086 = the reference's default (sduss/engine/arg_utils.py:20)."""
    g = torch.Generator(device=device).manual_seed(seed)
    out = {}
    for name, shape in sorted(param_shapes(cfg).items()):
        if name.endswith(".weight") and len(shape) >= 2:
            fan_in = 1
            for d in shape[1:]:
                fan_in *= d
            t = torch.randn(shape, generator=g, device=device) * fan_in ** -0.5
        elif name.endswith(".weight"):
            t = 1.0 + 0.1 * torch.randn(shape, generator=g, device=device)
        else:
            t = 0.05 * torch.randn(shape, generator=g, device=device)
        out[name] = t.to(torch.bfloat16)
    return out


def load_safetensors_dir(unet_dir: str) -> Tuple[UNetConfig, Dict[str, torch.Tensor]]:
    """Reads the HF layout the reference loads from (model_loader.py:64-66): <dir>/config.json +
    diffusion_pytorch_model*.safetensors."""
    from safetensors.torch import load_file
    cfg = UNetConfig.from_hf_json(os.path.join(unet_dir, "config.json"))
    params: Dict[str, torch.Tensor] = {}
    files = sorted(f for f in os.listdir(unet_dir) if f.endswith(".safetensors"))
    if not files:
        raise FileNotFoundError(f"no .safetensors under {unet_dir}")
    pick = [f for f in files if "fp16" not in f] or files
    for f in pick:
        params.update(load_file(os.path.join(unet_dir, f)))
    return cfg, params


def load_mmdit_safetensors_dir(transformer_dir: str):
    """SD3.5: <dir>/config.json + diffusion_pytorch_model*.safetensors of the HF ``transformer/`` sub-folder."""
    from safetensors.torch import load_file
    from .config import MMDiTConfig
    cfg = MMDiTConfig.from_hf_json(os.path.join(transformer_dir, "config.json"))
    files = sorted(f for f in os.listdir(transformer_dir) if f.endswith(".safetensors"))
    if not files:
        raise FileNotFoundError(f"no .safetensors under {transformer_dir}")
    params: Dict[str, torch.Tensor] = {}
    for f in [f for f in files if "fp16" not in f] or files:
        params.update(load_file(os.path.join(transformer_dir, f)))
    return cfg, params


def _conv_pack(w: torch.Tensor, pad_in: int = 0) -> torch.Tensor:
    o, i, kh, kw = w.shape
    w = w.permute(0, 2, 3, 1)  # O, kh, kw, I
    if pad_in and pad_in > i:
        w = torch.nn.functional.pad(w, (0, pad_in - i))
    return w.reshape(o, -1).contiguous()


def fold_layernorm(w: torch.Tensor, bias: Optional[torch.Tensor], gamma: torch.Tensor, beta: torch.Tensor):
    """LayerNorm(x; gamma, beta) W^T + bias == rstd (x W'^T - mean colsum) + bias' with W' = W * gamma (bf16, what the matrix core reads),
    colsum = row sums of the ROUNDED W' (fp32), bias' = bias + W beta (fp32): the operands of mx_gemm_desc.ln_stats (include/mxdenoise.h).
    w [N, K], gamma / beta [K] -> (W' bf16 [N, K], colsum fp32 [N], bias' fp32 [N])."""
    w32 = w.to(torch.float32)
    wf = (w32 * gamma.to(torch.float32)[None, :]).to(torch.bfloat16)
    colsum = wf.to(torch.float32).sum(dim=1)
    b = w32 @ beta.to(torch.float32)
    if bias is not None:
        b = b + bias.to(torch.float32)
    return wf.contiguous(), colsum.contiguous(), b.contiguous()


def params_as_held(cfg: UNetConfig, P: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """HF-named parameters with the values the device actually multiplies by.  Folding norm1 / norm2 / norm3 rounds the PRODUCT W * gamma to
    bf16 once -- the rounding any checkpoint weight gets -- instead of W alone, so for parameters that were bf16-exact to begin with (the
    synthetic ones of the parity tests) the folded linears see a 2^-9 relative perturbation the un-folded path would not.  A parity test
    of the kernels' arithmetic hands the oracle these weights (W' / gamma in fp32), the same way it hands it bf16-rounded weights."""
    out = dict(P)
    for p, _dim, _h, layers in transformer_names(cfg):
        for k in range(layers):
            b = f"{p}.transformer_blocks.{k}"
            for names, nn in (((f"{b}.attn1.to_q.weight", f"{b}.attn1.to_k.weight", f"{b}.attn1.to_v.weight"), "norm1"),
                              ((f"{b}.attn2.to_q.weight",), "norm2"), ((f"{b}.ff.net.0.proj.weight",), "norm3")):
                g = P[f"{b}.{nn}.weight"].to(torch.float32)
                safe = torch.where(g.abs() > 1e-6, g, torch.ones_like(g))
                for n in names:
                    w = P[n].to(torch.float32)
                    held = (w * g[None, :]).to(torch.bfloat16).to(torch.float32) / safe[None, :]
                    out[n] = torch.where(g.abs()[None, :] > 1e-6, held, w).to(P[n].dtype)
    return out


def _geglu_interleave(t: torch.Tensor) -> torch.Tensor:
    """rows [hidden(4C) ; gate(4C)] -> groups of [32 hidden | 32 gate]."""
    half = t.shape[0] // 2
    h = t[:half].reshape(half // 32, 32, *t.shape[1:])
    g = t[half:].reshape(half // 32, 32, *t.shape[1:])
    return torch.cat([h, g], dim=1).reshape(t.shape).contiguous()


def pack(cfg: UNetConfig, P: Dict[str, torch.Tensor]) -> List[Tuple[str, torch.Tensor]]:
    """HF-named params -> ordered list of (packed name, tensor in its blob dtype)."""
    missing = [k for k in param_shapes(cfg) if k not in P]
    if missing:
        raise KeyError(f"state dict lacks {len(missing)} tensors, e.g. {missing[:3]}")
    bf, f32 = torch.bfloat16, torch.float32
    out: List[Tuple[str, torch.Tensor]] = []

    def mat(name, t):
        out.append((name, t.to(bf).contiguous()))

    def vec(name, t):
        out.append((name, t.to(f32).contiguous()))

    mat("conv_in.weight", _conv_pack(P["conv_in.weight"], CONV_IN_PAD))
    vec("conv_in.bias", P["conv_in.bias"])
    for nm in ("time_embedding", "add_embedding"):
        for l in ("linear_1", "linear_2"):
            mat(f"{nm}.{l}.weight", P[f"{nm}.{l}.weight"])
            vec(f"{nm}.{l}.bias", P[f"{nm}.{l}.bias"])
    res = resnet_names(cfg)
    mat("temb_proj_all.weight", torch.cat([P[f"{p}.time_emb_proj.weight"] for p, _, _ in res], dim=0))
    vec("temb_proj_all.bias", torch.cat([P[f"{p}.time_emb_proj.bias"] for p, _, _ in res], dim=0))
    for p, cin, cout in res:
        vec(f"{p}.norm1.weight", P[f"{p}.norm1.weight"]); vec(f"{p}.norm1.bias", P[f"{p}.norm1.bias"])
        mat(f"{p}.conv1.weight", _conv_pack(P[f"{p}.conv1.weight"])); vec(f"{p}.conv1.bias", P[f"{p}.conv1.bias"])
        vec(f"{p}.norm2.weight", P[f"{p}.norm2.weight"]); vec(f"{p}.norm2.bias", P[f"{p}.norm2.bias"])
        mat(f"{p}.conv2.weight", _conv_pack(P[f"{p}.conv2.weight"])); vec(f"{p}.conv2.bias", P[f"{p}.conv2.bias"])
        if cin != cout:
            mat(f"{p}.conv_shortcut.weight", P[f"{p}.conv_shortcut.weight"].reshape(cout, cin))
            vec(f"{p}.conv_shortcut.bias", P[f"{p}.conv_shortcut.bias"])
    kv_by_dim: Dict[int, List[torch.Tensor]] = {}
    for p, dim, _h, layers in transformer_names(cfg):
        vec(f"{p}.norm.weight", P[f"{p}.norm.weight"]); vec(f"{p}.norm.bias", P[f"{p}.norm.bias"])
        for l in ("proj_in", "proj_out"):
            mat(f"{p}.{l}.weight", P[f"{p}.{l}.weight"].reshape(dim, dim)); vec(f"{p}.{l}.bias", P[f"{p}.{l}.bias"])
        for k in range(layers):
            b = f"{p}.transformer_blocks.{k}"
            # norm1 / norm2 / norm3 are folded into the one linear each of them feeds (fold_layernorm)
            def folded(name, w, bias, nn):
                wf, cs, bf_ = fold_layernorm(w, bias, P[f"{b}.{nn}.weight"], P[f"{b}.{nn}.bias"])
                mat(f"{name}.weight", wf); vec(f"{name}.colsum", cs); vec(f"{name}.bias", bf_)
            folded(f"{b}.attn1.to_qkv", torch.cat([P[f"{b}.attn1.to_q.weight"], P[f"{b}.attn1.to_k.weight"], P[f"{b}.attn1.to_v.weight"]], dim=0),
                   None, "norm1")
            mat(f"{b}.attn1.to_out.0.weight", P[f"{b}.attn1.to_out.0.weight"]); vec(f"{b}.attn1.to_out.0.bias", P[f"{b}.attn1.to_out.0.bias"])
            folded(f"{b}.attn2.to_q", P[f"{b}.attn2.to_q.weight"], None, "norm2")
            kv_by_dim.setdefault(dim, []).append(torch.cat([P[f"{b}.attn2.to_k.weight"], P[f"{b}.attn2.to_v.weight"]], dim=0))
            mat(f"{b}.attn2.to_out.0.weight", P[f"{b}.attn2.to_out.0.weight"]); vec(f"{b}.attn2.to_out.0.bias", P[f"{b}.attn2.to_out.0.bias"])
            folded(f"{b}.ff.net.0.proj", _geglu_interleave(P[f"{b}.ff.net.0.proj.weight"]), _geglu_interleave(P[f"{b}.ff.net.0.proj.bias"]), "norm3")
            mat(f"{b}.ff.net.2.weight", P[f"{b}.ff.net.2.weight"]); vec(f"{b}.ff.net.2.bias", P[f"{b}.ff.net.2.bias"])
    for dim, lst in kv_by_dim.items():
        mat(f"attn2_kv_all.{dim}.weight", torch.cat(lst, dim=0))
    n = len(cfg.block_out_channels)
    for i in range(n - 1):
        for p in (f"down_blocks.{i}.downsamplers.0.conv", f"up_blocks.{i}.upsamplers.0.conv"):
            mat(f"{p}.weight", _conv_pack(P[f"{p}.weight"])); vec(f"{p}.bias", P[f"{p}.bias"])
    vec("conv_norm_out.weight", P["conv_norm_out.weight"]); vec("conv_norm_out.bias", P["conv_norm_out.bias"])
    co = P["conv_out.weight"].shape[0]
    pad = (-co) % 4
    w = _conv_pack(P["conv_out.weight"]); bias = P["conv_out.bias"]
    if pad:
        w = torch.nn.functional.pad(w, (0, 0, 0, pad)); bias = torch.nn.functional.pad(bias, (0, pad))
    mat("conv_out.weight", w); vec("conv_out.bias", bias)
    return out


class PackedWeights:
    """One device blob + the (name, offset, bytes) table mx_unet_set_weights takes."""

    def __init__(self, entries: List[Tuple[str, torch.Tensor]], device):
        total = 0
        offs = []
        for _n, t in entries:
            total = (total + _ALIGN - 1) // _ALIGN * _ALIGN
            offs.append(total)
            total += t.numel() * t.element_size()
        self.blob = torch.empty(total + _ALIGN, dtype=torch.uint8, device=device)
        self.names = []
        for (name, t), off in zip(entries, offs):
            nb = t.numel() * t.element_size()
            self.blob[off:off + nb].copy_(t.reshape(-1).view(torch.uint8).to(device), non_blocking=False)
            self.names.append((name, off, nb))
        self.nbytes = total
        self._keep = [n.encode() for n, _, _ in self.names]
        arr = (_lib.WeightEntry * len(self.names))()
        for i, (n, off, nb) in enumerate(self.names):
            arr[i].name = self._keep[i]
            arr[i].offset = off
            arr[i].bytes = nb
        self.table = arr


# ------------------------------------------------------------------------------------------------------------------
# SD3.5 MMDiT
# ------------------------------------------------------------------------------------------------------------------
def synthetic_mmdit_params(cfg, device="cpu", seed: int = 10086) -> Dict[str, torch.Tensor]:
    """Random-init MMDiT weights (N(0,1)*fan_in^-1/2; AdaLN projections damped so activations stay O(1))."""
    from .config import mmdit_param_shapes
    g = torch.Generator(device=device).manual_seed(seed)
    out = {}
    for name, shape in sorted(mmdit_param_shapes(cfg).items()):
        if name == "pos_embed.pos_embed":
            t = 0.5 * torch.randn(shape, generator=g, device=device)
        elif name.endswith(("norm_q.weight", "norm_k.weight", "norm_added_q.weight", "norm_added_k.weight")):
            t = 1.0 + 0.1 * torch.randn(shape, generator=g, device=device)
        elif name.endswith(".weight"):
            fan_in = 1
            for d in shape[1:]:
                fan_in *= d
            gain = 0.3 if (".norm1" in name or "norm_out" in name) else 1.0
            t = torch.randn(shape, generator=g, device=device) * gain * fan_in ** -0.5
        else:
            t = 0.05 * torch.randn(shape, generator=g, device=device)
        out[name] = t.to(torch.bfloat16)
    return out


def pack_mmdit(cfg, P: Dict[str, torch.Tensor]) -> List[Tuple[str, torch.Tensor]]:
    """HF-named MMDiT params -> packed tensors (fused q|k|v projections, one stacked AdaLN projection, conv as GEMM)."""
    from .config import mmdit_param_shapes
    missing = [k for k in mmdit_param_shapes(cfg) if k not in P]
    if missing:
        raise KeyError(f"state dict lacks {len(missing)} tensors, e.g. {missing[:3]}")
    bf, f32 = torch.bfloat16, torch.float32
    out: List[Tuple[str, torch.Tensor]] = []

    def mat(name, t):
        out.append((name, t.to(bf).contiguous()))

    def vec(name, t):
        out.append((name, t.to(f32).contiguous()))

    def lin(dst, src):
        mat(f"{dst}.weight", P[f"{src}.weight"]); vec(f"{dst}.bias", P[f"{src}.bias"])

    d = cfg.dim
    mat("pos_embed.table", P["pos_embed.pos_embed"].reshape(-1, d))
    mat("pos_embed.proj.weight", _conv_pack(P["pos_embed.proj.weight"]))     # [d, (dy*ps+dx)*C + c]
    vec("pos_embed.proj.bias", P["pos_embed.proj.bias"])
    for n in ("time_text_embed.timestep_embedder.linear_1", "time_text_embed.timestep_embedder.linear_2",
              "time_text_embed.text_embedder.linear_1", "time_text_embed.text_embedder.linear_2", "context_embedder"):
        lin(n, n)
    ada_w, ada_b = [], []
    for i in range(cfg.num_layers):
        b = f"transformer_blocks.{i}"
        for n in (f"{b}.norm1.linear", f"{b}.norm1_context.linear"):
            ada_w.append(P[f"{n}.weight"]); ada_b.append(P[f"{n}.bias"])
    ada_w.append(P["norm_out.linear.weight"]); ada_b.append(P["norm_out.linear.bias"])
    mat("adaln_all.weight", torch.cat(ada_w, dim=0)); vec("adaln_all.bias", torch.cat(ada_b, dim=0))
    for i in range(cfg.num_layers):
        b = f"transformer_blocks.{i}"
        last = i == cfg.num_layers - 1
        dual = i in cfg.dual_attention_layers
        mat(f"{b}.attn.to_qkv.weight", torch.cat([P[f"{b}.attn.{n}.weight"] for n in ("to_q", "to_k", "to_v")], dim=0))
        vec(f"{b}.attn.to_qkv.bias", torch.cat([P[f"{b}.attn.{n}.bias"] for n in ("to_q", "to_k", "to_v")], dim=0))
        mat(f"{b}.attn.add_qkv.weight", torch.cat([P[f"{b}.attn.{n}.weight"] for n in ("add_q_proj", "add_k_proj", "add_v_proj")], dim=0))
        vec(f"{b}.attn.add_qkv.bias", torch.cat([P[f"{b}.attn.{n}.bias"] for n in ("add_q_proj", "add_k_proj", "add_v_proj")], dim=0))
        for n in ("norm_q", "norm_k", "norm_added_q", "norm_added_k"):
            vec(f"{b}.attn.{n}.weight", P[f"{b}.attn.{n}.weight"])
        lin(f"{b}.attn.to_out.0", f"{b}.attn.to_out.0")
        if not last:
            lin(f"{b}.attn.to_add_out", f"{b}.attn.to_add_out")
        if dual:
            mat(f"{b}.attn2.to_qkv.weight", torch.cat([P[f"{b}.attn2.{n}.weight"] for n in ("to_q", "to_k", "to_v")], dim=0))
            vec(f"{b}.attn2.to_qkv.bias", torch.cat([P[f"{b}.attn2.{n}.bias"] for n in ("to_q", "to_k", "to_v")], dim=0))
            for n in ("norm_q", "norm_k"):
                vec(f"{b}.attn2.{n}.weight", P[f"{b}.attn2.{n}.weight"])
            lin(f"{b}.attn2.to_out.0", f"{b}.attn2.to_out.0")
        lin(f"{b}.ff.net.0.proj", f"{b}.ff.net.0.proj"); lin(f"{b}.ff.net.2", f"{b}.ff.net.2")
        if not last:
            lin(f"{b}.ff_context.net.0.proj", f"{b}.ff_context.net.0.proj"); lin(f"{b}.ff_context.net.2", f"{b}.ff_context.net.2")
    lin("proj_out", "proj_out")
    return out
