"""Model configuration of the denoiser slot: the HF ``unet/config.json`` keys the forward depends on
(SURVEY.md section 8c lists them) and the inventory of HF state-dict tensors they imply."""
from __future__ import annotations

import json
from dataclasses import dataclass
from typing import Dict, Tuple


@dataclass
class UNetConfig:
    in_channels: int = 4
    out_channels: int = 4
    block_out_channels: Tuple[int, ...] = (320, 640, 1280)
    layers_per_block: int = 2
    down_has_attn: Tuple[bool, ...] = (False, True, True)   # CrossAttnDownBlock2D?
    transformer_layers_per_block: Tuple[int, ...] = (1, 2, 10)
    num_heads: Tuple[int, ...] = (5, 10, 20)                # HF "attention_head_dim" (a misnomer for SDXL)
    cross_attention_dim: int = 2048
    addition_time_embed_dim: int = 256
    projection_class_embeddings_input_dim: int = 2816
    norm_num_groups: int = 32
    norm_eps: float = 1e-5
    transformer_norm_eps: float = 1e-6
    layer_norm_eps: float = 1e-5
    # read by the pipeline (pipeline_stable_diffusion_xl_esymred.py:151,237)
    time_cond_proj_dim = None

    @property
    def time_embed_dim(self) -> int:
        return self.block_out_channels[0] * 4

    @property
    def text_embed_dim(self) -> int:
        return self.projection_class_embeddings_input_dim - 6 * self.addition_time_embed_dim

    @staticmethod
    def sdxl_base() -> "UNetConfig":
        """stabilityai/stable-diffusion-xl-base-1.0 unet/config.json."""
        return UNetConfig()

    @staticmethod
    def tiny() -> "UNetConfig":
        return UNetConfig(block_out_channels=(64, 128, 256), transformer_layers_per_block=(1, 1, 2),
                          num_heads=(1, 2, 4), cross_attention_dim=128, addition_time_embed_dim=32,
                          projection_class_embeddings_input_dim=64 + 6 * 32)

    @staticmethod
    def from_hf_json(path: str) -> "UNetConfig":
        with open(path) as f:
            c = json.load(f)
        n = len(c["block_out_channels"])
        heads = c["attention_head_dim"]
        heads = [heads] * n if isinstance(heads, int) else heads
        tl = c.get("transformer_layers_per_block", 1)
        tl = [tl] * n if isinstance(tl, int) else tl
        return UNetConfig(
            in_channels=c["in_channels"], out_channels=c["out_channels"],
            block_out_channels=tuple(c["block_out_channels"]), layers_per_block=c["layers_per_block"],
            down_has_attn=tuple(t.startswith("CrossAttn") for t in c["down_block_types"]),
            transformer_layers_per_block=tuple(tl), num_heads=tuple(heads),
            cross_attention_dim=c["cross_attention_dim"], addition_time_embed_dim=c["addition_time_embed_dim"],
            projection_class_embeddings_input_dim=c["projection_class_embeddings_input_dim"],
            norm_num_groups=c["norm_num_groups"], norm_eps=c["norm_eps"])


def resnet_names(cfg: UNetConfig):
    """(prefix, c_in, c_out) of every ResnetBlock2D in execution order (down, mid, up)."""
    ch = cfg.block_out_channels
    n = len(ch)
    out = []
    cur = ch[0]
    for i in range(n):
        for j in range(cfg.layers_per_block):
            out.append((f"down_blocks.{i}.resnets.{j}", cur, ch[i]))
            cur = ch[i]
    out.append(("mid_block.resnets.0", ch[-1], ch[-1]))
    out.append(("mid_block.resnets.1", ch[-1], ch[-1]))
    rev = list(reversed(ch))
    prev = rev[0]
    for i in range(n):
        oc = rev[i]
        ic = rev[min(i + 1, n - 1)]
        for j in range(cfg.layers_per_block + 1):
            skip = ic if j == cfg.layers_per_block else oc
            rin = prev if j == 0 else oc
            out.append((f"up_blocks.{i}.resnets.{j}", rin + skip, oc))
        prev = oc
    return out


def transformer_names(cfg: UNetConfig):
    """(prefix, dim, heads, layers) of every Transformer2DModel in execution order."""
    ch = cfg.block_out_channels
    n = len(ch)
    out = []
    for i in range(n):
        if cfg.down_has_attn[i]:
            for j in range(cfg.layers_per_block):
                out.append((f"down_blocks.{i}.attentions.{j}", ch[i], cfg.num_heads[i], cfg.transformer_layers_per_block[i]))
    out.append(("mid_block.attentions.0", ch[-1], cfg.num_heads[-1], cfg.transformer_layers_per_block[-1]))
    for i in range(n):
        lv = n - 1 - i
        if cfg.down_has_attn[lv]:
            for j in range(cfg.layers_per_block + 1):
                out.append((f"up_blocks.{i}.attentions.{j}", ch[lv], cfg.num_heads[lv], cfg.transformer_layers_per_block[lv]))
    return out


def param_shapes(cfg: UNetConfig) -> Dict[str, Tuple[int, ...]]:
    """{HF diffusers state-dict key: shape} -- what ``unet/diffusion_pytorch_model.safetensors`` holds."""
    ch = cfg.block_out_channels
    t = cfg.time_embed_dim
    s: Dict[str, Tuple[int, ...]] = {}
    s["conv_in.weight"] = (ch[0], cfg.in_channels, 3, 3)
    s["conv_in.bias"] = (ch[0],)
    for nm, k in (("time_embedding", ch[0]), ("add_embedding", cfg.projection_class_embeddings_input_dim)):
        s[f"{nm}.linear_1.weight"] = (t, k); s[f"{nm}.linear_1.bias"] = (t,)
        s[f"{nm}.linear_2.weight"] = (t, t); s[f"{nm}.linear_2.bias"] = (t,)
    for p, cin, cout in resnet_names(cfg):
        s[f"{p}.norm1.weight"] = (cin,); s[f"{p}.norm1.bias"] = (cin,)
        s[f"{p}.conv1.weight"] = (cout, cin, 3, 3); s[f"{p}.conv1.bias"] = (cout,)
        s[f"{p}.time_emb_proj.weight"] = (cout, t); s[f"{p}.time_emb_proj.bias"] = (cout,)
        s[f"{p}.norm2.weight"] = (cout,); s[f"{p}.norm2.bias"] = (cout,)
        s[f"{p}.conv2.weight"] = (cout, cout, 3, 3); s[f"{p}.conv2.bias"] = (cout,)
        if cin != cout:
            s[f"{p}.conv_shortcut.weight"] = (cout, cin, 1, 1); s[f"{p}.conv_shortcut.bias"] = (cout,)
    ctx = cfg.cross_attention_dim
    for p, dim, _h, layers in transformer_names(cfg):
        s[f"{p}.norm.weight"] = (dim,); s[f"{p}.norm.bias"] = (dim,)
        s[f"{p}.proj_in.weight"] = (dim, dim); s[f"{p}.proj_in.bias"] = (dim,)
        s[f"{p}.proj_out.weight"] = (dim, dim); s[f"{p}.proj_out.bias"] = (dim,)
        for k in range(layers):
            b = f"{p}.transformer_blocks.{k}"
            for nn in ("norm1", "norm2", "norm3"):
                s[f"{b}.{nn}.weight"] = (dim,); s[f"{b}.{nn}.bias"] = (dim,)
            for a, kd in (("attn1", dim), ("attn2", ctx)):
                s[f"{b}.{a}.to_q.weight"] = (dim, dim)
                s[f"{b}.{a}.to_k.weight"] = (dim, kd)
                s[f"{b}.{a}.to_v.weight"] = (dim, kd)
                s[f"{b}.{a}.to_out.0.weight"] = (dim, dim); s[f"{b}.{a}.to_out.0.bias"] = (dim,)
            s[f"{b}.ff.net.0.proj.weight"] = (8 * dim, dim); s[f"{b}.ff.net.0.proj.bias"] = (8 * dim,)
            s[f"{b}.ff.net.2.weight"] = (dim, 4 * dim); s[f"{b}.ff.net.2.bias"] = (dim,)
    n = len(ch)
    for i in range(n - 1):
        s[f"down_blocks.{i}.downsamplers.0.conv.weight"] = (ch[i], ch[i], 3, 3)
        s[f"down_blocks.{i}.downsamplers.0.conv.bias"] = (ch[i],)
        c = ch[n - 1 - i]
        s[f"up_blocks.{i}.upsamplers.0.conv.weight"] = (c, c, 3, 3)
        s[f"up_blocks.{i}.upsamplers.0.conv.bias"] = (c,)
    s["conv_norm_out.weight"] = (ch[0],); s["conv_norm_out.bias"] = (ch[0],)
    s["conv_out.weight"] = (cfg.out_channels, ch[0], 3, 3); s["conv_out.bias"] = (cfg.out_channels,)
    return s


# ------------------------------------------------------------------------------------------------------------------
# SD3.5 MMDiT (transformer/config.json of stabilityai/stable-diffusion-3.5-medium)
# ------------------------------------------------------------------------------------------------------------------
@dataclass
class MMDiTConfig:
    sample_size: int = 128
    patch_size: int = 2
    in_channels: int = 16
    out_channels: int = 16
    num_layers: int = 24
    attention_head_dim: int = 64
    num_attention_heads: int = 24
    joint_attention_dim: int = 4096
    caption_projection_dim: int = 1536
    pooled_projection_dim: int = 2048
    pos_embed_max_size: int = 384
    dual_attention_layers: Tuple[int, ...] = tuple(range(13))
    norm_eps: float = 1e-6

    @property
    def dim(self) -> int:
        return self.attention_head_dim * self.num_attention_heads

    @staticmethod
    def sd35_medium() -> "MMDiTConfig":
        return MMDiTConfig()

    @staticmethod
    def tiny() -> "MMDiTConfig":
        return MMDiTConfig(sample_size=16, num_layers=4, num_attention_heads=2, joint_attention_dim=128,
                           caption_projection_dim=128, pooled_projection_dim=64, pos_embed_max_size=24,
                           dual_attention_layers=(0, 1))

    @staticmethod
    def from_hf_json(path: str) -> "MMDiTConfig":
        with open(path) as f:
            c = json.load(f)
        return MMDiTConfig(sample_size=c["sample_size"], patch_size=c["patch_size"], in_channels=c["in_channels"],
                           out_channels=c.get("out_channels", c["in_channels"]), num_layers=c["num_layers"],
                           attention_head_dim=c["attention_head_dim"], num_attention_heads=c["num_attention_heads"],
                           joint_attention_dim=c["joint_attention_dim"], caption_projection_dim=c["caption_projection_dim"],
                           pooled_projection_dim=c["pooled_projection_dim"], pos_embed_max_size=c["pos_embed_max_size"],
                           dual_attention_layers=tuple(c.get("dual_attention_layers", ())))


def mmdit_param_shapes(cfg: MMDiTConfig) -> Dict[str, Tuple[int, ...]]:
    """{HF state-dict key: shape} of transformer/diffusion_pytorch_model.safetensors."""
    d = cfg.dim
    s: Dict[str, Tuple[int, ...]] = {}

    def lin(name, n, k):
        s[f"{name}.weight"] = (n, k)
        s[f"{name}.bias"] = (n,)

    s["pos_embed.pos_embed"] = (1, cfg.pos_embed_max_size ** 2, d)
    s["pos_embed.proj.weight"] = (d, cfg.in_channels, cfg.patch_size, cfg.patch_size)
    s["pos_embed.proj.bias"] = (d,)
    lin("time_text_embed.timestep_embedder.linear_1", d, 256)
    lin("time_text_embed.timestep_embedder.linear_2", d, d)
    lin("time_text_embed.text_embedder.linear_1", d, cfg.pooled_projection_dim)
    lin("time_text_embed.text_embedder.linear_2", d, d)
    lin("context_embedder", d, cfg.joint_attention_dim)
    for i in range(cfg.num_layers):
        b = f"transformer_blocks.{i}"
        last = i == cfg.num_layers - 1
        dual = i in cfg.dual_attention_layers
        lin(f"{b}.norm1.linear", (9 if dual else 6) * d, d)
        lin(f"{b}.norm1_context.linear", (2 if last else 6) * d, d)
        for nm in ("to_q", "to_k", "to_v", "add_q_proj", "add_k_proj", "add_v_proj"):
            lin(f"{b}.attn.{nm}", d, d)
        for nm in ("norm_q", "norm_k", "norm_added_q", "norm_added_k"):
            s[f"{b}.attn.{nm}.weight"] = (cfg.attention_head_dim,)
        lin(f"{b}.attn.to_out.0", d, d)
        if not last:
            lin(f"{b}.attn.to_add_out", d, d)
        if dual:
            for nm in ("to_q", "to_k", "to_v"):
                lin(f"{b}.attn2.{nm}", d, d)
            for nm in ("norm_q", "norm_k"):
                s[f"{b}.attn2.{nm}.weight"] = (cfg.attention_head_dim,)
            lin(f"{b}.attn2.to_out.0", d, d)
        lin(f"{b}.ff.net.0.proj", 4 * d, d)
        lin(f"{b}.ff.net.2", d, 4 * d)
        if not last:
            lin(f"{b}.ff_context.net.0.proj", 4 * d, d)
            lin(f"{b}.ff_context.net.2", d, 4 * d)
    lin("norm_out.linear", 2 * d, d)
    lin("proj_out", cfg.patch_size ** 2 * cfg.out_channels, d)
    return s
