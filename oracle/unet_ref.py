"""ORACLE (test infrastructure only) -- fp32/fp64 PyTorch restatement of
``diffusers==0.20.0`` ``UNet2DConditionModel.forward``.

PARITY UNPINNED: the reference (p1atdev/LECO) keeps this arithmetic in the
un-vendored third-party dependency ``diffusers==0.20.0`` (requirements.txt:1),
which is absent from /root/reference and from this image.  This file restates
the *published* algorithm; it is anchored only by
  * the reference's own call sites (train_util.py:156-160, :239-244),
  * the public parameter counts 859 520 964 (SD1.5) / 865 910 724 (SD2.1) /
    2 567 463 684 (SDXL), which ``tests/test_oracle_golden.py`` reproduces,
  * the LoRA module census 192 / 278 / 722 produced by the reference's own
    ``lora.py`` when applied to this tree (tests/test_reference_crosscheck.py).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline
leg may import this module.  The product path (leco_amd/) never does.

The module tree deliberately uses the diffusers class names
(``Transformer2DModel``, ``ResnetBlock2D``, ``Downsample2D``, ``Upsample2D``)
and plain ``torch.nn.Linear`` / ``torch.nn.Conv2d`` leaves with the diffusers
attribute names, because the reference discovers LoRA targets by
``__class__.__name__`` string match (lora.py:169-199).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Optional, Sequence, Tuple, Union

import torch
import torch.nn as nn
import torch.nn.functional as F


# ----------------------------------------------------------------------------
# configs (public unet/config.json values)
# ----------------------------------------------------------------------------
@dataclass
class UNetConfig:
    in_channels: int = 4
    out_channels: int = 4
    block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280)
    down_block_types: Tuple[str, ...] = (
        "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D")
    up_block_types: Tuple[str, ...] = (
        "UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D")
    layers_per_block: int = 2
    transformer_layers_per_block: Union[int, Tuple[int, ...]] = 1
    # diffusers' misnamed field: this is the NUMBER OF HEADS per level
    attention_head_dim: Union[int, Tuple[int, ...]] = 8
    cross_attention_dim: int = 768
    use_linear_projection: bool = False
    norm_num_groups: int = 32
    norm_eps: float = 1e-5
    sample_size: int = 64
    addition_embed_type: Optional[str] = None  # "text_time" for SDXL
    addition_time_embed_dim: int = 256
    projection_class_embeddings_input_dim: int = 2816

    def heads(self, level: int) -> int:
        a = self.attention_head_dim
        return a if isinstance(a, int) else a[level]

    def depth(self, level: int) -> int:
        a = self.transformer_layers_per_block
        return a if isinstance(a, int) else a[level]


def sd15_config() -> UNetConfig:
    return UNetConfig()


def sd21_config() -> UNetConfig:
    return UNetConfig(attention_head_dim=(5, 10, 20, 20), cross_attention_dim=1024,
                      use_linear_projection=True, sample_size=96)


def sdxl_config() -> UNetConfig:
    return UNetConfig(
        block_out_channels=(320, 640, 1280),
        down_block_types=("DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D"),
        up_block_types=("CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "UpBlock2D"),
        transformer_layers_per_block=(1, 2, 10), attention_head_dim=(5, 10, 20),
        cross_attention_dim=2048, use_linear_projection=True, sample_size=128,
        addition_embed_type="text_time")


def tiny_config(linear_proj: bool = False, xl: bool = False) -> UNetConfig:
    """Small SD1.x-shaped net (4 levels, 3 cross-attn + 1 plain) for CPU tests."""
    if xl:
        return UNetConfig(
            block_out_channels=(64, 128, 128),
            down_block_types=("DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D"),
            up_block_types=("CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "UpBlock2D"),
            layers_per_block=1, transformer_layers_per_block=(1, 1, 2),
            attention_head_dim=(2, 2, 2), cross_attention_dim=64,
            use_linear_projection=True, sample_size=16, addition_embed_type="text_time",
            addition_time_embed_dim=32, projection_class_embeddings_input_dim=6 * 32 + 64)
    return UNetConfig(block_out_channels=(64, 128, 128, 128), layers_per_block=1,
                      attention_head_dim=2, cross_attention_dim=64,
                      use_linear_projection=linear_proj, sample_size=16)


# ----------------------------------------------------------------------------
# building blocks (attribute names == diffusers state-dict keys)
# ----------------------------------------------------------------------------
def timestep_sinusoid(t: torch.Tensor, dim: int) -> torch.Tensor:
    """diffusers ``get_timestep_embedding(flip_sin_to_cos=True, freq_shift=0)``:
    [cos(t*f) | sin(t*f)], f_i = exp(-ln(10000) * i / (dim/2)) computed in fp32."""
    half = dim // 2
    exponent = -math.log(10000.0) * torch.arange(half, dtype=torch.float32, device=t.device) / half
    arg = t.float()[:, None] * torch.exp(exponent)[None, :]
    return torch.cat([torch.cos(arg), torch.sin(arg)], dim=-1)


class TimestepEmbedding(nn.Module):
    def __init__(self, in_dim: int, dim: int):
        super().__init__()
        self.linear_1 = nn.Linear(in_dim, dim)
        self.linear_2 = nn.Linear(dim, dim)

    def forward(self, x):
        return self.linear_2(F.silu(self.linear_1(x)))


class ResnetBlock2D(nn.Module):
    def __init__(self, cin: int, cout: int, temb_dim: int, groups: int, eps: float):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=eps)
        self.conv1 = nn.Conv2d(cin, cout, 3, 1, 1)
        self.time_emb_proj = nn.Linear(temb_dim, cout)
        self.norm2 = nn.GroupNorm(groups, cout, eps=eps)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1, 1, 0) if cin != cout else None

    def forward(self, x, temb):
        h = self.conv1(F.silu(self.norm1(x)))
        h = h + self.time_emb_proj(F.silu(temb))[:, :, None, None]
        h = self.conv2(F.silu(self.norm2(h)))
        if self.conv_shortcut is not None:
            x = self.conv_shortcut(x)
        return x + h


class Downsample2D(nn.Module):
    def __init__(self, c: int):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, 2, 1)

    def forward(self, x):
        return self.conv(x)


class Upsample2D(nn.Module):
    def __init__(self, c: int):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, 1, 1)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))


class Attention(nn.Module):
    def __init__(self, dim: int, ctx_dim: Optional[int], heads: int):
        super().__init__()
        self.heads = heads
        ctx_dim = dim if ctx_dim is None else ctx_dim
        self.to_q = nn.Linear(dim, dim, bias=False)
        self.to_k = nn.Linear(ctx_dim, dim, bias=False)
        self.to_v = nn.Linear(ctx_dim, dim, bias=False)
        self.to_out = nn.ModuleList([nn.Linear(dim, dim), nn.Dropout(0.0)])

    def forward(self, x, ctx=None):
        ctx = x if ctx is None else ctx
        B, S, C = x.shape
        h, d = self.heads, C // self.heads
        q = self.to_q(x).view(B, S, h, d).transpose(1, 2)
        k = self.to_k(ctx).view(B, -1, h, d).transpose(1, 2)
        v = self.to_v(ctx).view(B, -1, h, d).transpose(1, 2)
        p = torch.softmax((q @ k.transpose(-1, -2)) * (d ** -0.5), dim=-1)
        o = (p @ v).transpose(1, 2).reshape(B, S, C)
        return self.to_out[1](self.to_out[0](o))


class GEGLU(nn.Module):
    def __init__(self, dim: int, inner: int):
        super().__init__()
        self.proj = nn.Linear(dim, inner * 2)

    def forward(self, x):
        a, g = self.proj(x).chunk(2, dim=-1)
        return a * F.gelu(g)  # erf GELU


class FeedForward(nn.Module):
    def __init__(self, dim: int):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(dim, dim * 4), nn.Dropout(0.0), nn.Linear(dim * 4, dim)])

    def forward(self, x):
        for m in self.net:
            x = m(x)
        return x


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim: int, heads: int, ctx_dim: int):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.attn1 = Attention(dim, None, heads)
        self.norm2 = nn.LayerNorm(dim)
        self.attn2 = Attention(dim, ctx_dim, heads)
        self.norm3 = nn.LayerNorm(dim)
        self.ff = FeedForward(dim)

    def forward(self, h, ctx):
        h = self.attn1(self.norm1(h)) + h
        h = self.attn2(self.norm2(h), ctx) + h
        h = self.ff(self.norm3(h)) + h
        return h


class Transformer2DModel(nn.Module):
    def __init__(self, dim: int, heads: int, ctx_dim: int, depth: int, groups: int, linear_proj: bool):
        super().__init__()
        self.use_linear_projection = linear_proj
        self.norm = nn.GroupNorm(groups, dim, eps=1e-6)
        self.proj_in = nn.Linear(dim, dim) if linear_proj else nn.Conv2d(dim, dim, 1, 1, 0)
        self.transformer_blocks = nn.ModuleList(
            [BasicTransformerBlock(dim, heads, ctx_dim) for _ in range(depth)])
        self.proj_out = nn.Linear(dim, dim) if linear_proj else nn.Conv2d(dim, dim, 1, 1, 0)

    def forward(self, x, ctx):
        B, C, H, W = x.shape
        res = x
        h = self.norm(x)
        if self.use_linear_projection:
            h = self.proj_in(h.permute(0, 2, 3, 1).reshape(B, H * W, C))
        else:
            h = self.proj_in(h).permute(0, 2, 3, 1).reshape(B, H * W, C)
        for blk in self.transformer_blocks:
            h = blk(h, ctx)
        if self.use_linear_projection:
            h = self.proj_out(h).reshape(B, H, W, C).permute(0, 3, 1, 2)
        else:
            h = self.proj_out(h.reshape(B, H, W, C).permute(0, 3, 1, 2))
        return h + res


class _DownBlock(nn.Module):
    def __init__(self, cfg: UNetConfig, level: int, cin: int, cout: int, temb: int, attn: bool, last: bool):
        super().__init__()
        if attn:
            self.attentions = nn.ModuleList([
                Transformer2DModel(cout, cfg.heads(level), cfg.cross_attention_dim, cfg.depth(level),
                                   cfg.norm_num_groups, cfg.use_linear_projection)
                for _ in range(cfg.layers_per_block)])
        self.resnets = nn.ModuleList([
            ResnetBlock2D(cin if j == 0 else cout, cout, temb, cfg.norm_num_groups, cfg.norm_eps)
            for j in range(cfg.layers_per_block)])
        self.downsamplers = None if last else nn.ModuleList([Downsample2D(cout)])
        self.has_attn = attn

    def forward(self, h, temb, ctx):
        outs = []
        for j, res in enumerate(self.resnets):
            h = res(h, temb)
            if self.has_attn:
                h = self.attentions[j](h, ctx)
            outs.append(h)
        if self.downsamplers is not None:
            h = self.downsamplers[0](h)
            outs.append(h)
        return h, outs


class CrossAttnDownBlock2D(_DownBlock):
    pass


class DownBlock2D(_DownBlock):
    pass


class UNetMidBlock2DCrossAttn(nn.Module):
    def __init__(self, cfg: UNetConfig, c: int, temb: int):
        super().__init__()
        lvl = len(cfg.block_out_channels) - 1
        self.attentions = nn.ModuleList([
            Transformer2DModel(c, cfg.heads(lvl), cfg.cross_attention_dim, cfg.depth(lvl),
                               cfg.norm_num_groups, cfg.use_linear_projection)])
        self.resnets = nn.ModuleList([
            ResnetBlock2D(c, c, temb, cfg.norm_num_groups, cfg.norm_eps) for _ in range(2)])

    def forward(self, h, temb, ctx):
        h = self.resnets[0](h, temb)
        h = self.attentions[0](h, ctx)
        return self.resnets[1](h, temb)


class _UpBlock(nn.Module):
    def __init__(self, cfg: UNetConfig, level: int, prev: int, cout: int, cin_skip_last: int,
                 temb: int, attn: bool, last: bool):
        super().__init__()
        n = cfg.layers_per_block + 1
        if attn:
            self.attentions = nn.ModuleList([
                Transformer2DModel(cout, cfg.heads(level), cfg.cross_attention_dim, cfg.depth(level),
                                   cfg.norm_num_groups, cfg.use_linear_projection)
                for _ in range(n)])
        res = []
        for j in range(n):
            skip = cin_skip_last if j == n - 1 else cout
            rin = prev if j == 0 else cout
            res.append(ResnetBlock2D(rin + skip, cout, temb, cfg.norm_num_groups, cfg.norm_eps))
        self.resnets = nn.ModuleList(res)
        self.upsamplers = None if last else nn.ModuleList([Upsample2D(cout)])
        self.has_attn = attn

    def forward(self, h, skips, temb, ctx):
        for j, res in enumerate(self.resnets):
            h = torch.cat([h, skips.pop()], dim=1)
            h = res(h, temb)
            if self.has_attn:
                h = self.attentions[j](h, ctx)
        if self.upsamplers is not None:
            h = self.upsamplers[0](h)
        return h


class CrossAttnUpBlock2D(_UpBlock):
    pass


class UpBlock2D(_UpBlock):
    pass


class _Out:
    def __init__(self, sample):
        self.sample = sample


class UNet2DConditionModel(nn.Module):
    """Call signature as used by the reference: ``unet(x, t, encoder_hidden_states=...,
    [added_cond_kwargs=...]).sample`` (train_util.py:156-160, 239-244)."""

    def __init__(self, cfg: UNetConfig):
        super().__init__()
        self.cfg = cfg
        ch = cfg.block_out_channels
        temb = ch[0] * 4
        self.conv_in = nn.Conv2d(cfg.in_channels, ch[0], 3, 1, 1)
        self.time_embedding = TimestepEmbedding(ch[0], temb)
        if cfg.addition_embed_type == "text_time":
            self.add_embedding = TimestepEmbedding(cfg.projection_class_embeddings_input_dim, temb)
        self.down_blocks = nn.ModuleList()
        self.up_blocks = nn.ModuleList()
        cout = ch[0]
        for i, t in enumerate(cfg.down_block_types):
            cin, cout = cout, ch[i]
            cls = CrossAttnDownBlock2D if t.startswith("CrossAttn") else DownBlock2D
            self.down_blocks.append(cls(cfg, i, cin, cout, temb, t.startswith("CrossAttn"),
                                        i == len(ch) - 1))
        self.mid_block = UNetMidBlock2DCrossAttn(cfg, ch[-1], temb)
        rev = list(reversed(ch))
        cout = rev[0]
        for i, t in enumerate(cfg.up_block_types):
            prev, cout = cout, rev[i]
            skip_last = rev[min(i + 1, len(ch) - 1)]
            cls = CrossAttnUpBlock2D if t.startswith("CrossAttn") else UpBlock2D
            self.up_blocks.append(cls(cfg, len(ch) - 1 - i, prev, cout, skip_last, temb,
                                      t.startswith("CrossAttn"), i == len(ch) - 1))
        self.conv_norm_out = nn.GroupNorm(cfg.norm_num_groups, ch[0], eps=cfg.norm_eps)
        self.conv_out = nn.Conv2d(ch[0], cfg.out_channels, 3, 1, 1)

    def forward(self, sample, timestep, encoder_hidden_states=None, added_cond_kwargs=None):
        cfg = self.cfg
        B = sample.shape[0]
        t = torch.as_tensor(timestep, device=sample.device)
        if t.ndim == 0:
            t = t[None]
        t = t.expand(B)
        emb = self.time_embedding(timestep_sinusoid(t, cfg.block_out_channels[0]).to(sample.dtype))
        if cfg.addition_embed_type == "text_time":
            te = added_cond_kwargs["text_embeds"]
            ids = added_cond_kwargs["time_ids"]
            tid = timestep_sinusoid(ids.flatten(), cfg.addition_time_embed_dim).reshape(B, -1)
            add = torch.cat([te, tid.to(te.dtype)], dim=-1).to(emb.dtype)
            emb = emb + self.add_embedding(add)
        h = self.conv_in(sample)
        skips = [h]
        for blk in self.down_blocks:
            h, outs = blk(h, emb, encoder_hidden_states)
            skips += outs
        h = self.mid_block(h, emb, encoder_hidden_states)
        for blk in self.up_blocks:
            h = blk(h, skips, emb, encoder_hidden_states)
        h = self.conv_out(F.silu(self.conv_norm_out(h)))
        return _Out(h)

    # reference call sites use these (train_lora.py:67-70); no-ops here
    def enable_xformers_memory_efficient_attention(self):
        return None


def init_synthetic_(unet: nn.Module, seed: int = 1234, gain_out: float = 1.0) -> nn.Module:
    """Seeded PyTorch-default inits; norm gamma=1, beta=0.  Deterministic
    on CPU for a given torch version; the weights are created on CPU then moved."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in unet.named_parameters():
            if p.ndim >= 2:
                fan_in = p[0].numel()
                bound = 1.0 / math.sqrt(fan_in)
                p.copy_((torch.rand(p.shape, generator=g) * 2 - 1) * bound)
            elif "norm" in name:
                p.fill_(1.0 if name.endswith("weight") else 0.0)
            else:
                p.copy_((torch.rand(p.shape, generator=g) * 2 - 1) * 0.02)
    return unet
