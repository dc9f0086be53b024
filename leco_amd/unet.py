"""UNet2DConditionModel on the MI355X hot path.

This module replaces ``diffusers.UNet2DConditionModel.forward`` and its autograd backward --
the reference's call sites are ``train_util.py:156-160`` / ``:239-244`` (forward) and
``train_lora.py:279`` (``loss.backward()``) -- with a *static launch plan* of the hand-written
HIP kernels in ``libleco_hip.so``:

* The module tree keeps the diffusers class / attribute names (``Transformer2DModel``,
  ``ResnetBlock2D``, ``Downsample2D``, ``Upsample2D``; leaves are plain ``torch.nn.Linear`` /
  ``torch.nn.Conv2d``) because the reference discovers LoRA targets and derives the saved key
  names from them (``lora.py:169-199``), and so that diffusers-format state dicts load as is.
  The leaves only *hold* weights; nothing here calls ``leaf.forward``.
* For a given (batch, h, w) the engine builds ONE plan: every activation gets its own
  pre-allocated channels-last bf16 buffer (288 GB of HBM: nothing is recomputed or re-used, so
  the forward of the LoRA-on "target" pass doubles as the saved-activation set of the backward),
  and every layer appends its kernel launches (``ops.Op``) to a forward list and pushes a
  closure on a tape; unrolling the tape in reverse emits the backward list (dgrad everywhere
  downstream of the first LoRA site, wgrad only for LoRA down/up).  A plan is replayed eagerly
  or as one hipGraph launch.
* Weight operands are re-laid once for the MFMA kernels: ``W[N][K]`` (conv: ``[N][kh][kw][Cin]``),
  q|k|v fused to one ``[3C][C]`` GEMM, k|v of cross-attention fused, all 22 ``time_emb_proj``
  fused into one GEMM; a transposed / flipped copy serves the dgrad GEMMs.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple, Union

import torch
import torch.nn as nn

from . import hip, ops
from .hip import (ACT_GEGLU, ACT_NONE, ACT_SILU, A_CONV3_S1, A_CONV3_S2, A_CONV3_TR2, A_CONV3_UP2, A_PLAIN, gemm_args)

bf16 = torch.bfloat16


# =============================================================================================
# configuration (public unet/config.json values of the three model families)
# =============================================================================================
@dataclass
class UNetConfig:
    in_channels: int = 4
    out_channels: int = 4
    block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280)
    down_block_types: Tuple[str, ...] = ("CrossAttnDownBlock2D",) * 3 + ("DownBlock2D",)
    up_block_types: Tuple[str, ...] = ("UpBlock2D",) + ("CrossAttnUpBlock2D",) * 3
    layers_per_block: int = 2
    transformer_layers_per_block: Union[int, Tuple[int, ...]] = 1
    attention_head_dim: Union[int, Tuple[int, ...]] = 8  # diffusers' name; it is the head COUNT
    cross_attention_dim: int = 768
    use_linear_projection: bool = False
    norm_num_groups: int = 32
    norm_eps: float = 1e-5
    sample_size: int = 64
    addition_embed_type: Optional[str] = None
    addition_time_embed_dim: int = 256
    projection_class_embeddings_input_dim: int = 2816

    def heads(self, level: int) -> int:
        a = self.attention_head_dim
        return a if isinstance(a, int) else a[level]

    def depth(self, level: int) -> int:
        a = self.transformer_layers_per_block
        return a if isinstance(a, int) else a[level]

    @classmethod
    def from_dict(cls, d: dict) -> "UNetConfig":
        keys = {f for f in cls.__dataclass_fields__}
        kw = {k: (tuple(v) if isinstance(v, list) else v) for k, v in d.items() if k in keys}
        return cls(**kw)


def sd15_config() -> UNetConfig:
    return UNetConfig()


def sd21_config() -> UNetConfig:
    return UNetConfig(attention_head_dim=(5, 10, 20, 20), cross_attention_dim=1024, use_linear_projection=True,
                      sample_size=96)


def sdxl_config() -> UNetConfig:
    return UNetConfig(block_out_channels=(320, 640, 1280),
                      down_block_types=("DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D"),
                      up_block_types=("CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "UpBlock2D"),
                      transformer_layers_per_block=(1, 2, 10), attention_head_dim=(5, 10, 20),
                      cross_attention_dim=2048, use_linear_projection=True, sample_size=128,
                      addition_embed_type="text_time")


# =============================================================================================
# weight-holding module tree (names == diffusers state-dict keys)
# =============================================================================================
class TimestepEmbedding(nn.Module):
    def __init__(self, in_dim, dim):
        super().__init__()
        self.linear_1 = nn.Linear(in_dim, dim)
        self.linear_2 = nn.Linear(dim, dim)


class ResnetBlock2D(nn.Module):
    def __init__(self, cin, cout, temb, groups, eps):
        super().__init__()
        self.in_channels, self.out_channels = cin, cout
        self.norm1 = nn.GroupNorm(groups, cin, eps=eps)
        self.conv1 = nn.Conv2d(cin, cout, 3, 1, 1)
        self.time_emb_proj = nn.Linear(temb, cout)
        self.norm2 = nn.GroupNorm(groups, cout, eps=eps)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1, 1, 0) if cin != cout else None


class Downsample2D(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, 2, 1)


class Upsample2D(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, 1, 1)


class Attention(nn.Module):
    def __init__(self, dim, ctx_dim, heads):
        super().__init__()
        self.heads = heads
        ctx_dim = dim if ctx_dim is None else ctx_dim
        self.to_q = nn.Linear(dim, dim, bias=False)
        self.to_k = nn.Linear(ctx_dim, dim, bias=False)
        self.to_v = nn.Linear(ctx_dim, dim, bias=False)
        self.to_out = nn.ModuleList([nn.Linear(dim, dim), nn.Dropout(0.0)])


class GEGLU(nn.Module):
    def __init__(self, dim, inner):
        super().__init__()
        self.proj = nn.Linear(dim, inner * 2)


class FeedForward(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(dim, dim * 4), nn.Dropout(0.0), nn.Linear(dim * 4, dim)])


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim, heads, ctx_dim):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.attn1 = Attention(dim, None, heads)
        self.norm2 = nn.LayerNorm(dim)
        self.attn2 = Attention(dim, ctx_dim, heads)
        self.norm3 = nn.LayerNorm(dim)
        self.ff = FeedForward(dim)


class Transformer2DModel(nn.Module):
    def __init__(self, dim, heads, ctx_dim, depth, groups, linear_proj):
        super().__init__()
        self.use_linear_projection = linear_proj
        self.norm = nn.GroupNorm(groups, dim, eps=1e-6)
        self.proj_in = nn.Linear(dim, dim) if linear_proj else nn.Conv2d(dim, dim, 1, 1, 0)
        self.transformer_blocks = nn.ModuleList([BasicTransformerBlock(dim, heads, ctx_dim) for _ in range(depth)])
        self.proj_out = nn.Linear(dim, dim) if linear_proj else nn.Conv2d(dim, dim, 1, 1, 0)


class _DownBlock(nn.Module):
    def __init__(self, cfg, level, cin, cout, temb, attn, last):
        super().__init__()
        self.has_attn = attn
        if attn:
            self.attentions = nn.ModuleList([
                Transformer2DModel(cout, cfg.heads(level), cfg.cross_attention_dim, cfg.depth(level),
                                   cfg.norm_num_groups, cfg.use_linear_projection)
                for _ in range(cfg.layers_per_block)])
        self.resnets = nn.ModuleList([
            ResnetBlock2D(cin if j == 0 else cout, cout, temb, cfg.norm_num_groups, cfg.norm_eps)
            for j in range(cfg.layers_per_block)])
        self.downsamplers = None if last else nn.ModuleList([Downsample2D(cout)])


class CrossAttnDownBlock2D(_DownBlock):
    pass


class DownBlock2D(_DownBlock):
    pass


class UNetMidBlock2DCrossAttn(nn.Module):
    def __init__(self, cfg, c, temb):
        super().__init__()
        lvl = len(cfg.block_out_channels) - 1
        self.attentions = nn.ModuleList([
            Transformer2DModel(c, cfg.heads(lvl), cfg.cross_attention_dim, cfg.depth(lvl), cfg.norm_num_groups,
                               cfg.use_linear_projection)])
        self.resnets = nn.ModuleList([ResnetBlock2D(c, c, temb, cfg.norm_num_groups, cfg.norm_eps) for _ in range(2)])


class _UpBlock(nn.Module):
    def __init__(self, cfg, level, prev, cout, skip_last, temb, attn, last):
        super().__init__()
        self.has_attn = attn
        n = cfg.layers_per_block + 1
        if attn:
            self.attentions = nn.ModuleList([
                Transformer2DModel(cout, cfg.heads(level), cfg.cross_attention_dim, cfg.depth(level),
                                   cfg.norm_num_groups, cfg.use_linear_projection) for _ in range(n)])
        res = []
        for j in range(n):
            skip = skip_last if j == n - 1 else cout
            rin = prev if j == 0 else cout
            res.append(ResnetBlock2D(rin + skip, cout, temb, cfg.norm_num_groups, cfg.norm_eps))
        self.resnets = nn.ModuleList(res)
        self.upsamplers = None if last else nn.ModuleList([Upsample2D(cout)])


class CrossAttnUpBlock2D(_UpBlock):
    pass


class UpBlock2D(_UpBlock):
    pass


class UNetOutput:
    def __init__(self, sample):
        self.sample = sample


# =============================================================================================
# plan-time tensors and the build-time autograd tape
# =============================================================================================
class TRef:
    """A row-major [rows][cols] activation view (row stride ``ld``; bf16, or fp32 in the fp32 compute mode) inside a
    device buffer."""
    __slots__ = ("t", "ptr", "ld", "rows", "cols", "rg", "gparts", "name", "cstats")

    def __init__(self, t: torch.Tensor, rows: int, cols: int, ld: Optional[int] = None, offset: int = 0,
                 rg: bool = False, name: str = ""):
        self.t = t
        self.ptr = t.data_ptr() + offset * t.element_size()
        self.ld = cols if ld is None else ld
        self.rows, self.cols = rows, cols
        self.rg = rg            # requires grad (some LoRA site is upstream)
        self.gparts: List["TRef"] = []
        self.name = name
        self.cstats: Optional[int] = None   # device address of fp32 [B][cols][2] {sum, sumsq} left by the producer, or None

    def cols_view(self, c0: int, c1: int) -> "TRef":
        v = TRef(self.t, self.rows, c1 - c0, self.ld, 0, self.rg, self.name)
        v.ptr = self.ptr + c0 * self.t.element_size()
        return v


class LoraSiteState:
    """Packed MFMA operands of one GEMM site that has LoRA modules attached."""

    def __init__(self, site: "GemmSite", mods: list, r: int, dev):
        self.mods = mods  # per group: LoRAModule or None
        self.r = r
        g = len(mods)
        self.R = g * r
        self.R16 = (self.R + 15) // 16 * 16
        # columns of the packed up / down images: one 32- or 64-wide K-extension step, or (ranks whose stacked
        # columns exceed 64) several 64-wide steps, the first riding in the main GEMM, the rest chained behind it
        if site.conv3:
            # the low-rank image T = conv3x3(x, down) is a whole number of 64-channel chunks (the conv gathers walk 64-channel
            # chunks); ranks above 64 chain further 64-column slices of T . (scale up)^T behind the main convolution
            self.Rp = (self.R + 63) // 64 * 64
        else:
            self.Rp = (self.R + 31) // 32 * 32 if self.R <= 64 else (self.R + 63) // 64 * 64
        K, N = site.lora_k, site.n
        dt = site.eng.adt
        # dn_s / up_t get Rp rows (rows beyond R16 stay zero) so they can be GEMM weight operands
        self.dn_s = torch.zeros(self.Rp, K, dtype=dt, device=dev)
        self.up_p = torch.zeros(N, self.Rp, dtype=dt, device=dev)
        self.up_t = torch.zeros(self.Rp, N, dtype=dt, device=dev)
        self.dn_p = torch.zeros(K, self.Rp, dtype=dt, device=dev)
        # GEGLU input projections keep a second, row-interleaved copy of up_p for the fused-GEGLU epilogue (bf16 path)
        self.up_pg = torch.zeros(N, self.Rp, dtype=dt, device=dev) if (site.geglu_ok and not site.eng.f32) else None


class GemmSite:
    """One contraction of the UNet (possibly several fused leaves) with its packed operands."""

    def __init__(self, eng: "Engine", name: str, leaves: Sequence[Tuple[str, nn.Module]], conv3: bool = False,
                 stride2: bool = False, up2: bool = False, extra_bias: Optional[torch.Tensor] = None):
        self.eng, self.name = eng, name
        self.leaf_names = [n for n, _ in leaves]
        self.conv3, self.stride2, self.up2 = conv3, stride2, up2
        dev = eng.device
        ws, bs = [], []
        for _, m in leaves:
            w = m.weight.detach()
            if w.ndim == 4 and w.shape[2] == 3:
                w = w.permute(0, 2, 3, 1).reshape(w.shape[0], -1)  # [N][kh][kw][Cin]
            elif w.ndim == 4:
                w = w.reshape(w.shape[0], -1)
            ws.append(w)
            bs.append(None if m.bias is None else m.bias.detach().float())
        self.group_n = ws[0].shape[0]
        self.w = torch.cat(ws, 0).to(device=dev, dtype=eng.adt).contiguous()
        self.n, self.k = self.w.shape
        self.cin = self.k // 9 if conv3 else self.k
        self.lora_k = self.k
        if any(b is not None for b in bs):
            self.bias = torch.cat([b if b is not None else torch.zeros(w.shape[0]) for b, w in zip(bs, ws)]).to(dev)
        else:
            self.bias = None
        self._wt = None
        self._wg = None
        self.lora: Optional[LoraSiteState] = None
        # GEGLU.proj sites (N = 2F, F % 64 == 0) can run with the GEGLU fused into the GEMM epilogue
        self.geglu_ok = (not conv3) and name.endswith("ff.net.0.proj") and self.n % 128 == 0

    @staticmethod
    def geglu_perm(n: int, device) -> torch.Tensor:
        """source row of interleaved row i: blocks of 128 = 64 value rows then their 64 gate rows (LECO_ACT_GEGLU)."""
        i = torch.arange(n, device=device)
        j, r = i // 128, i % 128
        return torch.where(r < 64, j * 64 + r, n // 2 + j * 64 + (r - 64))

    @property
    def w_geglu(self) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
        """(weights, bias) with the LECO_ACT_GEGLU row interleave."""
        if self._wg is None:
            perm = self.geglu_perm(self.n, self.w.device)
            self._wg = (self.w[perm].contiguous(), None if self.bias is None else self.bias[perm].contiguous())
        return self._wg

    @property
    def wt(self) -> torch.Tensor:
        """dgrad operand: W^T [K][N] (conv: spatially flipped, in/out swapped, [Cin][kh][kw][Cout])."""
        if self._wt is None:
            if self.conv3:
                w4 = self.w.reshape(self.n, 3, 3, self.cin)
                self._wt = w4.flip(1, 2).permute(3, 1, 2, 0).reshape(self.cin, 9 * self.n).contiguous()
            else:
                self._wt = self.w.t().contiguous()
        return self._wt


# =============================================================================================
# the engine
# =============================================================================================
class Plan:
    def __init__(self):
        # named launch lists; callers may add their own (e.g. the fused denoising pass)
        self.lists: Dict[str, List[ops.Op]] = {"fwd_on": [], "fwd_off": [], "bwd": []}
        self.fwd: Dict[bool, List[ops.Op]] = {True: self.lists["fwd_on"], False: self.lists["fwd_off"]}
        self.bwd: List[ops.Op] = self.lists["bwd"]
        self.bufs: Dict[str, torch.Tensor] = {}
        self.graphs: Dict[str, C.c_void_p] = {}
        self.key: Optional[tuple] = None      # its key in Engine.plans (set by Engine.plan)
        self.ctx_src = None                   # the tensor `ctx` was last filled from (FusedStep skips identical refills)

    def set_ctx(self, t: torch.Tensor, src=None) -> None:
        """The ONE writer of the prompt-context buffer.  ``src``: an identity token for the contents (FusedStep passes the
        cached embedding tensor of the prompt pair and skips the copy while the plan still holds it); None = anonymous
        contents, the next tokened call copies again.  Plans are shared through ``Engine.plans`` (keyed by shape), so every
        writer has to go through here -- a direct ``plan.ctx.copy_`` would leave a stale token behind."""
        if src is not None and self.ctx_src is src:
            return
        self.ctx.copy_(t)
        self.ctx_src = src


class Engine:
    def __init__(self, unet: "UNet2DConditionModel", device: torch.device):
        self.unet, self.cfg, self.device = unet, unet.cfg, device
        # compute mode = the dtype the model holds when its engine is built: bf16 MFMA path, or -- for models kept in
        # torch.float32, what `train.precision: float32` selects (config_util.py:75-83, train_lora.py:54-67) -- the fp32
        # mode of csrc/f32.hip: fp32 activations / weights / LoRA operands, exact fp32 contractions, no fused fast paths
        self.f32 = unet.conv_in.weight.dtype == torch.float32
        self.adt = torch.float32 if self.f32 else bf16
        self.esz = 4 if self.f32 else 2
        self.sites: Dict[str, GemmSite] = {}      # by site name
        self.leaf_site: Dict[str, Tuple[GemmSite, int]] = {}  # leaf qualified name -> (site, group)
        self.plans: Dict[tuple, Plan] = {}
        self.network = None  # LoRANetwork (set by attach_lora)
        self.temb_lora_sites: Dict[str, GemmSite] = {}
        self.use_graphs = False
        # LECO_DETERMINISTIC=1: bitwise reproducible steps -- the LoRA weight gradients (the only fp32 atomics of a step)
        # are accumulated through per-slab partials in the workspace instead; plans built afterwards pick it up
        self.deterministic = ops.deterministic_default()
        # shared fp32 scratch for split-K partial slabs (all launches are stream-ordered)
        self.workspace = torch.empty(32 * 1024 * 1024, dtype=torch.float32, device=device)
        self._pack()

    # ---- weight packing ------------------------------------------------------------------------
    def _site(self, name, leaves, **kw) -> GemmSite:
        s = GemmSite(self, name, leaves, **kw)
        self.sites[name] = s
        for g, (ln, _) in enumerate(leaves):
            self.leaf_site[ln] = (s, g)
        return s

    def _f32(self, t) -> torch.Tensor:
        return t.detach().float().to(self.device).contiguous()

    def _pack(self):
        u = self.unet
        named = dict(u.named_modules())
        self.named = named
        # fused time-embedding projection of every ResnetBlock2D (+ conv1 bias folded in)
        self.resnets = [(n, m) for n, m in named.items() if isinstance(m, ResnetBlock2D)]
        self.temb_off: Dict[str, int] = {}
        off = 0
        for n, m in self.resnets:
            self.temb_off[n] = off
            off += m.out_channels
        self.temb_total = off
        s = self._site("time_emb_proj_all", [(n + ".time_emb_proj", m.time_emb_proj) for n, m in self.resnets])
        s.bias = torch.cat([(m.time_emb_proj.bias.detach().float() + m.conv1.bias.detach().float())
                            for _, m in self.resnets]).to(self.device)
        self._site("time_embedding.linear_1", [("time_embedding.linear_1", u.time_embedding.linear_1)])
        self._site("time_embedding.linear_2", [("time_embedding.linear_2", u.time_embedding.linear_2)])
        if self.cfg.addition_embed_type == "text_time":
            self._site("add_embedding.linear_1", [("add_embedding.linear_1", u.add_embedding.linear_1)])
            self._site("add_embedding.linear_2", [("add_embedding.linear_2", u.add_embedding.linear_2)])
        self.conv_in_w = self._f32(u.conv_in.weight.detach().permute(1, 2, 3, 0))  # [Cin][3][3][Cout]
        self.conv_in_b = self._f32(u.conv_in.bias)
        self.conv_out_w = u.conv_out.weight.detach().permute(0, 2, 3, 1).contiguous().to(self.device, self.adt)
        self.conv_out_b = self._f32(u.conv_out.bias)
        self.norm_p: Dict[str, Tuple[torch.Tensor, torch.Tensor]] = {}
        for n, m in named.items():
            if isinstance(m, (nn.GroupNorm, nn.LayerNorm)):
                self.norm_p[n] = (self._f32(m.weight), self._f32(m.bias))
            if isinstance(m, ResnetBlock2D):
                c1 = self._site(n + ".conv1", [(n + ".conv1", m.conv1)], conv3=True)
                c1.bias = None  # folded into the fused time-embedding bias
                self._site(n + ".conv2", [(n + ".conv2", m.conv2)], conv3=True)
                if m.conv_shortcut is not None:
                    self._site(n + ".conv_shortcut", [(n + ".conv_shortcut", m.conv_shortcut)])
            elif isinstance(m, Downsample2D):
                self._site(n + ".conv", [(n + ".conv", m.conv)], conv3=True, stride2=True)
            elif isinstance(m, Upsample2D):
                self._site(n + ".conv", [(n + ".conv", m.conv)], conv3=True, up2=True)
            elif isinstance(m, Transformer2DModel):
                self._site(n + ".proj_in", [(n + ".proj_in", m.proj_in)])
                self._site(n + ".proj_out", [(n + ".proj_out", m.proj_out)])
            elif isinstance(m, BasicTransformerBlock):
                a1, a2 = m.attn1, m.attn2
                self._site(n + ".attn1.qkv", [(n + ".attn1.to_q", a1.to_q), (n + ".attn1.to_k", a1.to_k),
                                              (n + ".attn1.to_v", a1.to_v)])
                self._site(n + ".attn1.to_out.0", [(n + ".attn1.to_out.0", a1.to_out[0])])
                self._site(n + ".attn2.to_q", [(n + ".attn2.to_q", a2.to_q)])
                self._site(n + ".attn2.kv", [(n + ".attn2.to_k", a2.to_k), (n + ".attn2.to_v", a2.to_v)])
                self._site(n + ".attn2.to_out.0", [(n + ".attn2.to_out.0", a2.to_out[0])])
                self._site(n + ".ff.net.0.proj", [(n + ".ff.net.0.proj", m.ff.net[0].proj)])
                self._site(n + ".ff.net.2", [(n + ".ff.net.2", m.ff.net[2])])

    # ---- LoRA ------------------------------------------------------------------------------------
    def attach_lora(self, network) -> None:
        """Bind a LoRANetwork: every LoRA module is matched to (site, group) by its leaf name."""
        self.network = network
        self.temb_lora_sites: Dict[str, GemmSite] = {}
        per_site: Dict[str, list] = {}
        for lora in network.unet_loras:
            if lora.leaf_name not in self.leaf_site:
                raise KeyError(f"LoRA target {lora.leaf_name} has no GEMM site")
            site, g = self.leaf_site[lora.leaf_name]
            if site.name == "time_emb_proj_all":
                # c3lier: a LoRA on ResnetBlock2D.time_emb_proj.  The fused 22-way projection cannot carry 22
                # independent low-rank terms in one 64-wide K-extension, so those resnets get their own small GEMM.
                rname = lora.leaf_name[:-len(".time_emb_proj")]
                m = self.named[rname]
                site = self._site(lora.leaf_name, [(lora.leaf_name, m.time_emb_proj)])
                site.bias = (m.time_emb_proj.bias.detach().float() + m.conv1.bias.detach().float()).to(self.device)
                self.temb_lora_sites[rname] = site
                g = 0
            per_site.setdefault(site.name, [None] * len(site.leaf_names))[g] = lora
        self.lora_sites: List[GemmSite] = []
        for name, mods in per_site.items():
            site = self.sites[name]
            r = next(m for m in mods if m is not None).lora_dim
            site.lora = LoraSiteState(site, mods, r, self.device)
            self.lora_sites.append(site)
        self.plans.clear()
        self._build_pack_table()

    def _build_pack_table(self):
        net = self.network
        sites = (hip.LoraSite * len(self.lora_sites))()
        self._zero_dummy = {}
        for i, site in enumerate(self.lora_sites):
            st, d = site.lora, sites[i]
            for g, mod in enumerate(st.mods):
                if mod is None:  # group without a LoRA module: zero operands
                    key = (st.r, site.k, site.group_n)
                    if key not in self._zero_dummy:
                        self._zero_dummy[key] = (torch.zeros(st.r, site.k, dtype=self.adt, device=self.device),
                                                 torch.zeros(site.group_n, st.r, dtype=self.adt, device=self.device))
                    dn, up = self._zero_dummy[key]
                    d.down[g], d.up[g] = dn.data_ptr(), up.data_ptr()
                else:
                    # bf16 path: the bf16 shadow of the slab; fp32 mode: the fp32 master slab itself
                    src = net.slab if self.f32 else net.shadow
                    d.down[g] = src.data_ptr() + mod.down_off * self.esz
                    d.up[g] = src.data_ptr() + mod.up_off * self.esz
            d.groups, d.r, d.k, d.n = len(st.mods), st.r, site.k, site.n
            d.scale = 0.0  # filled by refresh_lora
            d.taps = 9 if site.conv3 else 1
            d.dn_s, d.up_p, d.up_t, d.dn_p = st.dn_s.data_ptr(), st.up_p.data_ptr(), st.up_t.data_ptr(), st.dn_p.data_ptr()
            d.up_pg = st.up_pg.data_ptr() if st.up_pg is not None else None
            d.rp = st.Rp
        self._pack_host = sites
        self._pack_dev = torch.zeros(C.sizeof(sites), dtype=torch.uint8, device=self.device)
        self._pack_scale = None

    def refresh_lora(self, multiplier: float) -> None:
        """Re-pack all LoRA operand images from the bf16 shadow (after an optimizer step)."""
        net = self.network
        if self._pack_scale != multiplier:
            for i, site in enumerate(self.lora_sites):
                mod = next(m for m in site.lora.mods if m is not None)
                self._pack_host[i].scale = float(multiplier * mod.scale)
            raw = torch.frombuffer(bytearray(bytes(self._pack_host)), dtype=torch.uint8)
            self._pack_dev.copy_(raw)
            self._pack_scale = multiplier
        with ops.f32_mode(self.f32):
            ops.lora_pack(self._pack_dev, len(self.lora_sites)).run()
        net._packed_version = net.version

    # ---- plan construction ---------------------------------------------------------------------
    def plan(self, B: int, h: int, w: int, need_bwd: bool = True, ws_slot: int = 0, share: int = 1, tag: Optional[str] = None) -> Plan:
        """``need_bwd=False`` builds a forward-only plan (no gradient buffers): used for the batched
        LoRA-off passes, which never run a backward.  ``ws_slot`` > 0 gives the plan's split-K launches a workspace of
        their own (the only mutable buffer plans share), so that its lists may run CONCURRENTLY with another plan's on a
        second stream (`FusedStep`: the batched frozen pass beside the LoRA-on target pass).  ``share`` > 1 (forward-only
        plans): the caller guarantees that the B input latents are ``share`` back-to-back copies of B / share samples at one
        timestep (``predict_noise``'s ``cat([latents] * 2)``, train_util.py:151) -- everything in front of the first use of
        the prompt embeddings is then computed once per distinct sample (`PlanBuilder.shared_prefix`)."""
        key = (B, h, w, need_bwd) if ws_slot == 0 else (B, h, w, need_bwd, ws_slot)
        if share > 1:
            assert not need_bwd and B % share == 0
            key = key + ("share", share)
        if tag:      # a private copy of an otherwise identical plan (FusedStep's de-duplicated passes: their batch sizes collide
            key = key + ("tag", tag)      # with the faithful plans of another resident (bs, h, w) bucket, and buckets are evicted alone)
        if key not in self.plans:
            with ops.f32_mode(self.f32):
                self.plans[key] = PlanBuilder(self, B, h, w, need_bwd, ws=self.workspace_slot(ws_slot), share=share,
                                              ).build()
                self.plans[key].key = key
        return self.plans[key]

    def workspace_slot(self, slot: int) -> torch.Tensor:
        if slot == 0:
            return self.workspace
        extra = self.__dict__.setdefault("_workspaces", {})
        if slot not in extra:
            extra[slot] = torch.empty_like(self.workspace)
        return extra[slot]

    def drop_plan(self, key: tuple) -> None:
        """Release one plan: its captured hipGraphs, then (by dropping the references) its activation buffers.
        `dynamic_resolution` prompts (train_lora.py:160-170) visit up to 25 (h, w) buckets of several GB each."""
        plan = self.plans.pop(key, None)
        if plan is None:
            return
        if plan.graphs:
            if self.device.type == "cuda":
                torch.cuda.synchronize(self.device)
            lib = _graph_api()
            for g in plan.graphs.values():
                lib.leco_graph_destroy(g)
            plan.graphs.clear()
        plan.lists.clear()
        plan.bufs.clear()

    def release(self) -> None:
        """Drop every plan (instantiated hipGraphs, activation buffers).  The engine stays usable: plans are rebuilt on
        demand."""
        for key in list(self.plans):
            self.drop_plan(key)


class PlanBuilder:
    def __init__(self, eng: Engine, B: int, h: int, w: int, need_bwd: bool = True, ws: Optional[torch.Tensor] = None,
                 share: int = 1):
        self.eng, self.cfg, self.dev = eng, eng.cfg, eng.device
        self.share = share
        self.Bfull = B          # (self.B is lowered to B / share while the batch-shared prefix is built)
        self.ws = eng.workspace if ws is None else ws      # split-K slabs of this plan's launches
        self.B, self.h, self.w = B, h, w
        self.need_bwd = need_bwd
        self.plan = Plan()
        self.f_on: List[ops.Op] = self.plan.fwd[True]
        self.f_off: List[ops.Op] = self.plan.fwd[False]
        self.tape: List = []
        self.nbuf = 0
        self.wgrad_problems: List[dict] = []     # LoRA weight gradients of the backward: one grouped launch at its end
        self.wgrad_keep: List = []
        # GroupNorm statistics from the producers (leco_gemm_args.col_stats): every tensor that feeds a GroupNorm -- the
        # outputs of the resnet / down / upsampler convolutions, of Transformer2DModel.proj_out and of conv_in -- gets a
        # fp32 [B][C][2] slice of ONE arena, zeroed by one memset at the head of each forward list; the GroupNorm itself is
        # then a single apply pass (leco_groupnorm_apply_stats).  Off in deterministic mode (the statistics are accumulated
        # with fp32 atomics) and in the fp32 mode.  LECO_GN_FUSED: "auto" (default) = only tensors whose GroupNorm would
        # otherwise take the three-launch path (few large slices: the 64^2 level at UNet batch 4) -- measured, that is where
        # it pays (-13 us per GroupNorm); "1" = every tensor (2 % SLOWER on the whole step: the one-launch GroupNorm of the
        # smaller levels is launch-latency bound and the statistics walk costs its producers ~1 us); "0" = off.
        import os
        self.gn_mode = os.environ.get("LECO_GN_FUSED", "auto")
        self.gn_fused = (not eng.f32) and (not eng.deterministic) and self.gn_mode != "0"
        self.stat_used = 0
        self.stat_arena = None
        # statistics are kept per ATOM of adjacent channels: the largest unit that divides every GroupNorm group and every
        # concat split of the network (SD: 320 / 32 = 10)
        self.stat_atom = max(1, self.cfg.block_out_channels[0] // self.cfg.norm_num_groups)
        if self.gn_fused:
            chans = sum(st.n for nm, st in eng.sites.items() if st.conv3 or nm.endswith(".proj_out")) + self.cfg.block_out_channels[0]
            self.stat_arena = torch.zeros(2 * B * chans + 64, dtype=torch.float32, device=self.dev)

    # ---- helpers -----------------------------------------------------------------------------
    def buf(self, name, shape, dtype=None, zero=False) -> torch.Tensor:
        t = (torch.zeros if zero else torch.empty)(shape, dtype=dtype or self.eng.adt, device=self.dev)
        key = name
        while key in self.plan.bufs:
            self.nbuf += 1
            key = f"{name}#{self.nbuf}"
        self.plan.bufs[key] = t
        return t

    def act(self, name, rows, cols, rg=False) -> TRef:
        return TRef(self.buf(name, (rows, cols)), rows, cols, rg=rg, name=name)

    def both(self, op: ops.Op):
        self.f_on.append(op)
        self.f_off.append(op)

    def stat_slice(self, cols: int, hw: int, batch: Optional[int] = None) -> Optional[int]:
        """Device address of a fresh [B][cols / atom][2] slice of the statistics arena (None when the fusion is off or does
        not pay for a tensor of this shape)."""
        B = self.B if batch is None else batch
        if not self.gn_fused or cols % self.stat_atom:
            return None
        if self.gn_mode == "auto" and hip.lib().leco_groupnorm_single_launch(B, hw, cols, self.cfg.norm_num_groups):
            return None
        n = 2 * B * (cols // self.stat_atom)
        assert self.stat_used + n <= self.stat_arena.numel(), "GroupNorm statistics arena too small"
        p = self.stat_arena.data_ptr() + 4 * self.stat_used
        self.stat_used += n
        return p

    def total_grad(self, t: TRef, out: List[ops.Op]) -> Optional[TRef]:
        """Sum of the gradient contributions recorded for ``t`` (None if there are none)."""
        parts = t.gparts
        if not parts:
            return None
        if len(parts) == 1:
            return parts[0]
        acc = parts[0]
        i = 1
        while i < len(parts):
            b = parts[i]
            c = parts[i + 1] if i + 1 < len(parts) else None
            dst = self.act("g_sum." + t.name, t.rows, t.cols)
            out.append(ops.add(acc.ptr, acc.ld, b.ptr, b.ld, c.ptr if c else None, c.ld if c else 0, dst.ptr, dst.ld,
                               t.rows, t.cols))
            acc = dst
            i += 2
        return acc

    # ---- GEMM site forward / backward -----------------------------------------------------------
    def gemm_fwd(self, site: GemmSite, x: Union[TRef, Tuple[TRef, TRef]], name: str, *, conv=None, amode=A_PLAIN,
                 rows: int, residual: Optional[TRef] = None, rowbias=None, rows_per_group=0, ld_rowbias=0,
                 act=ACT_NONE, out: Optional[TRef] = None, out_f32: Optional[torch.Tensor] = None,
                 bias="site", ldc32_override: int = 0, geglu: bool = False, stats_hw: int = 0) -> TRef:
        if geglu:
            return self._gemm_fwd_geglu(site, x, name, rows)
        xs = x if isinstance(x, tuple) else (x,)
        rg_in = any(t.rg for t in xs) or (residual is not None and residual.rg)
        lora = site.lora
        y = out if out is not None else (self.act(name, rows, site.n) if out_f32 is None else None)
        bias_t = site.bias if bias == "site" else bias
        common = dict(m=rows, n=site.n, k=site.k, a_mode=amode, conv=conv, bias=bias_t, rowbias=rowbias,
                      rows_per_group=rows_per_group, ld_rowbias=ld_rowbias,
                      residual=residual.ptr if residual is not None else None,
                      ldr=residual.ld if residual is not None else 0, act=act, out_f32=out_f32,
                      ldc32=(ldc32_override or (out_f32.shape[-1] if out_f32 is not None else 0)))
        if len(xs) == 2:
            common.update(a1=xs[1].ptr, lda1=xs[1].ld, k_split=xs[0].cols)
        # GroupNorm statistics of the output from this launch's epilogue (not when further LoRA slices accumulate into y)
        if stats_hw and y is not None and (lora is None or lora.Rp <= 64):
            y.cstats = self.stat_slice(site.n, stats_hw)
            if y.cstats is not None:
                common.update(col_stats=y.cstats, stats_rows=stats_hw, stats_atom=self.stat_atom)
        if (y is not None and y.cstats is None and out_f32 is None and self.xgemm_ok(site, xs, amode, rows, rowbias, act)
                and bias == "site"):
            return self._gemm_fwd_x(site, xs[0], y, rows, residual, rg_in)
        a0, lda0 = xs[0].ptr, xs[0].ld
        yptr = y.ptr if y is not None else None
        ldc = y.ld if y is not None else site.n
        g_off = gemm_args(a0, site.w, yptr, lda=lda0, ldc=ldc, **common)
        self.f_off.append(ops.gemm(g_off, keep=(site, xs, residual, y), ws=self.ws))
        T = None
        self._last_T = None
        if lora is not None:
            T = self._last_T = self.act(name + ".loraT", rows, lora.Rp)
            if amode == A_PLAIN and lora.Rp == 32 and not self.eng.f32:
                # down-projection fused into the main GEMM's K sweep (T is still written: lora_up wgrad)
                g_on = gemm_args(a0, site.w, yptr, lda=lda0, ldc=ldc, w_ext=lora.up_p, ext_k=32, ld_wext=32,
                                 t_w=lora.dn_s, t_rows=lora.R16, t_out=T.ptr, ld_tout=T.ld, **common)
                self.f_on.append(ops.gemm(g_on, keep=(site, lora, xs, residual, y, T), ws=self.ws))
            else:
                kw = dict(m=rows, n=lora.Rp, k=site.k, a_mode=amode, conv=conv)
                if len(xs) == 2:
                    kw.update(a1=xs[1].ptr, lda1=xs[1].ld, k_split=xs[0].cols)
                g_t = gemm_args(a0, lora.dn_s, T.ptr, lda=lda0, ldc=T.ld, **kw)
                self.f_on.append(ops.gemm(g_t, keep=(lora, xs, T), ws=self.ws))
                e0 = min(lora.Rp, 64)
                chain_after = y is not None and act == ACT_NONE
                on_common = common
                if lora.Rp > 64 and not chain_after:
                    # fp32-output or activated sites (c3lier time_emb_proj): the slices beyond the first 64 columns are summed
                    # FIRST into a bf16 image that the main launch adds through its residual operand (inside the activation)
                    assert residual is None, "a LoRA rank above 64 on an activated / fp32-output site that also has a residual"
                    pre = self.act(name + ".loraHi", rows, site.n)
                    for c in range(64, lora.Rp, 64):
                        g_c = gemm_args(T.ptr + self.eng.esz * c, lora.up_p.data_ptr() + self.eng.esz * c, pre.ptr, m=rows, n=site.n,
                                        k=64, lda=T.ld, ldw=lora.Rp, ldc=pre.ld, residual=pre.ptr if c > 64 else None, ldr=pre.ld)
                        self.f_on.append(ops.gemm(g_c, keep=(lora, T, pre), ws=self.ws))
                    on_common = dict(common, residual=pre.ptr, ldr=pre.ld)
                g_on = gemm_args(a0, site.w, yptr, lda=lda0, ldc=ldc, a_ext=T.ptr, w_ext=lora.up_p, ext_k=e0,
                                 ld_aext=T.ld, ld_wext=lora.Rp, **on_common)
                self.f_on.append(ops.gemm(g_on, keep=(site, lora, xs, residual, y), ws=self.ws))
                for c in range(64, lora.Rp if chain_after else 0, 64):   # further 64-column slices of T . (scale up)^T, accumulated into y
                    g_c = gemm_args(T.ptr + self.eng.esz * c, lora.up_p.data_ptr() + self.eng.esz * c, y.ptr, m=rows, n=site.n, k=64, lda=T.ld,
                                    ldw=lora.Rp, ldc=y.ld, residual=y.ptr, ldr=y.ld)
                    self.f_on.append(ops.gemm(g_c, keep=(lora, T, y), ws=self.ws))
        else:
            self.f_on.append(ops.gemm(g_off, ws=self.ws))
        if y is not None:
            y.rg = rg_in or lora is not None
            if y.rg:
                self.tape.append(lambda: self.gemm_bwd(site, xs, y, T, conv, amode, rows, residual))
        return y

    # ---- A-stationary GEMM (csrc/xgemm.hip) for the short-K / small-M Linears of the forward-only plans ------------------
    def xgemm_ok(self, site: GemmSite, xs, amode, rows: int, rowbias, act) -> bool:
        """The launch can go to `leco_xgemm`: forward-only bf16 plan, plain single-source operand, no per-sample bias /
        activation, a shape the kernel covers and (LoRA sites) the <= 32-column fused down-projection images.
        ``LECO_XGEMM=0`` keeps the LDS-ring kernel; ``LECO_XGEMM_MAX_M`` bounds the row count it is used for.  Measured on
        MI355X (profiles/r05_bench_xgemm_var*.txt): the A-stationary kernel wins only where the ring GEMM cannot fill its
        tiles -- M = 256 (the 8^2 level at UNet batch 4): 7.5 vs 10.1 us; at M = 1024 it ties (13.9 - 15.8 vs 14.7 us) and
        from M = 3072 on it loses 1.3 - 1.5x: weight fragments streamed global -> VGPR, 16 bytes per lane, top out near
        330 TFLOP/s, below what the LDS ring reaches once its tiles are full.  Hence the default bound of 256 rows."""
        import os
        if self.need_bwd or self.eng.f32 or os.environ.get("LECO_XGEMM", "1") in ("", "0"):
            return False
        if amode != A_PLAIN or len(xs) != 1 or rowbias is not None or act != ACT_NONE or site.conv3:
            return False
        if rows > int(os.environ.get("LECO_XGEMM_MAX_M", "256")) or not ops.xgemm_supported(rows, site.n, site.k):
            return False
        lo = site.lora
        return lo is None or (lo.Rp == 32 and lo.dn_s is not None and lo.up_p is not None)

    def _gemm_fwd_x(self, site: GemmSite, x: TRef, y: TRef, rows: int, residual: Optional[TRef], rg_in: bool) -> TRef:
        for lora_on, lst in ((True, self.f_on), (False, self.f_off)):
            lin, kp = self._xlin(site, lora_on)
            lst.append(ops.xgemm(x.ptr, x.ld, lin, y.ptr, y.ld, rows, site.n, site.k,
                                 residual=residual.ptr if residual is not None else None,
                                 ldr=residual.ld if residual is not None else 0, keep=(kp, x, y, residual)))
        self._last_T = None
        y.rg = rg_in or site.lora is not None
        return y

    def _gemm_fwd_geglu(self, site: GemmSite, x: TRef, name: str, rows: int) -> TRef:
        """GEGLU.proj with value * gelu(gate) fused into the epilogue (forward-only plans): output [rows][N/2]."""
        assert site.geglu_ok and not self.need_bwd
        wg, bg = site.w_geglu
        y = self.act(name, rows, site.n // 2)
        common = dict(m=rows, n=site.n, k=site.k, bias=bg, act=ACT_GEGLU, lda=x.ld, ldc=y.ld)
        self.f_off.append(ops.gemm(gemm_args(x.ptr, wg, y.ptr, **common), keep=(site, x, y, wg, bg), ws=self.ws))
        lora = site.lora
        if lora is not None and lora.Rp == 32 and lora.up_pg is not None:
            T = self.act(name + ".loraT", rows, 32)
            g_on = gemm_args(x.ptr, wg, y.ptr, w_ext=lora.up_pg, ext_k=32, ld_wext=32, t_w=lora.dn_s, t_rows=lora.R16,
                             t_out=T.ptr, ld_tout=T.ld, **common)
            self.f_on.append(ops.gemm(g_on, keep=(site, lora, x, y, T, wg, bg), ws=self.ws))
        elif lora is not None:
            raise RuntimeError("fused GEGLU needs the rank-<=32 fused down-projection path")
        else:
            self.f_on.append(self.f_off[-1])
        return y

    def lora_bwd(self, site: GemmSite, xs, dy: TRef, T: TRef, conv, amode, rows, name: str, fuse_u: bool = False) -> TRef:
        """U = dY * up (per group) and the LoRA weight gradients of one site; returns U [rows][Rp].  ``fuse_u``: the
        caller's dgrad GEMM forms U inside its own K sweep (`t_w` = up_t, the mirror image of the forward's fused
        down-projection) and writes it to the returned buffer -- the wgrad launches must then be appended AFTER it
        (`lora_wgrads`)."""
        out, lora = self.plan.bwd, site.lora
        U = self.act("g." + name + ".loraU", rows, lora.Rp)
        if not fuse_u:
            out.append(ops.gemm(gemm_args(dy.ptr, lora.up_t, U.ptr, m=rows, n=lora.Rp, k=site.n, lda=dy.ld, ldc=U.ld),
                                keep=(lora, dy, U)))
            self.lora_wgrads(site, xs, dy, T, U, conv, amode, rows)
        return U

    def lora_wgrads(self, site: GemmSite, xs, dy: TRef, T: TRef, U: TRef, conv, amode, rows) -> None:
        """The LoRA weight gradients of one site.  Default: recorded as problems of the ONE grouped launch that `build`
        appends at the end of the backward (every operand buffer of a plan stays intact until then); deterministic mode:
        one atomic-free launch pair per problem, in place."""
        out, lora, net = self.plan.bwd, site.lora, self.eng.network
        gn, r = site.group_n, lora.r
        cin_total = sum(t.cols for t in xs)
        # atomic-free wgrad accumulation: deterministic mode, and always in the fp32 mode (one fixed-order kernel per problem)
        det = self.eng.workspace if (self.eng.deterministic or self.eng.f32) else None
        esz = self.eng.esz
        det_bytes = 0 if det is None else det.numel() * det.element_size()

        def emit(p, ldp, q, ldq, g, g_sj, g_sc, cols, s, cv=None, keep=(), r=r):
            if r > 16 and not self.eng.f32:   # the bf16 wgrad kernel keeps <= 16 rank columns in registers: one problem per 16-column slice
                for j0 in range(0, r, 16):
                    emit(p + esz * j0, ldp, q, ldq, g + 4 * j0 * g_sj, g_sj, g_sc, cols, s, cv, keep, min(16, r - j0))
                return
            if det is None:
                pr = dict(p=p, ldp=ldp, q=q, ldq=ldq, g=g, g_sj=g_sj, g_sc=g_sc, m=rows, r=r, cols=cols, scale=s)
                if cv is not None:
                    pr.update(a_mode=cv[0], h_out=cv[1], w_out=cv[2], h_in=cv[3], w_in=cv[4], kh=cv[5], kw=cv[6])
                self.wgrad_problems.append(pr)
                self.wgrad_keep.append(keep)
            elif cv is None:
                out.append(ops.Op("leco_lora_wgrad", (p, ldp, q, ldq, g, g_sj, g_sc, rows, r, cols, s, ops.ptr(det), det_bytes),
                                  keep=(keep, det)))
            else:
                out.append(ops.Op("leco_lora_wgrad_conv", (p, ldp, q, ldq, g, g_sj, g_sc, rows, r, cols, s, *cv,
                                                           ops.ptr(det), det_bytes), keep=(keep, det)))
        for g, mod in enumerate(lora.mods):
            if mod is None:
                continue
            s = float(mod.scale)  # multiplier == 1 inside the training pass
            gdown = net.grad.data_ptr() + 4 * mod.down_off
            c_off = 0
            for t in xs:
                if amode == A_PLAIN:
                    # d lora_down[j][c_off + c] = s * sum_m U[m][g r + j] x[m][c]
                    emit(U.ptr + esz * g * r, U.ld, t.ptr, t.ld, gdown + 4 * c_off, cin_total, 1, t.cols, s, keep=(U, t))
                else:
                    # conv lora_down [r][Cin][3][3]: one gathered product per tap
                    _, ho, wo, hi, wi = conv
                    for tap in range(9):
                        emit(U.ptr + esz * g * r, U.ld, t.ptr, t.ld, gdown + 4 * (c_off * 9 + tap), cin_total * 9, 9, t.cols, s,
                             cv=(amode, ho, wo, hi, wi, tap // 3, tap % 3), keep=(U, t))
                c_off += t.cols
            # d lora_up[n][j] = s * sum_m dy[m][g gn + n] T[m][g r + j]
            emit(T.ptr + esz * g * r, T.ld, dy.ptr + esz * g * gn, dy.ld, net.grad.data_ptr() + 4 * mod.up_off, 1, r, gn, s,
                 keep=(T, dy))

    def gemm_bwd(self, site: GemmSite, xs, y: TRef, T: Optional[TRef], conv, amode, rows, residual):
        out = self.plan.bwd
        dy = self.total_grad(y, out)
        if dy is None:
            return
        if residual is not None and residual.rg:
            residual.gparts.append(dy)
        lora = site.lora
        need = [t for t in xs if t.rg]
        # plain sites whose input needs a gradient: U = dY up^T rides in the dgrad GEMM's K sweep (fused `t_w`), like T in
        # the forward -- 192 skinny N = 32 GEMMs (~9 us each) less per backward
        fuse_u = lora is not None and bool(need) and amode == A_PLAIN and lora.Rp == 32 and not self.eng.f32
        U = self.lora_bwd(site, xs, dy, T, conv, amode, rows, y.name, fuse_u) if lora is not None else None
        if not need:
            return
        kin = sum(t.cols for t in xs)
        if amode == A_PLAIN:
            dx = self.act("g." + y.name + ".dx", rows, kin)
            if fuse_u:
                g = gemm_args(dy.ptr, site.wt, dx.ptr, m=rows, n=kin, k=site.n, lda=dy.ld, ldc=dx.ld, w_ext=lora.dn_p,
                              ext_k=32, ld_wext=lora.Rp, t_w=lora.up_t, t_rows=lora.R16, t_out=U.ptr, ld_tout=U.ld)
                out.append(ops.gemm(g, keep=(site, lora, dy, dx, U), ws=self.ws))
                self.lora_wgrads(site, xs, dy, T, U, conv, amode, rows)
            else:
                g = gemm_args(dy.ptr, site.wt, dx.ptr, m=rows, n=kin, k=site.n, lda=dy.ld, ldc=dx.ld,
                              a_ext=U.ptr if U is not None else None, w_ext=lora.dn_p if lora is not None else None,
                              ext_k=min(lora.Rp, 64) if lora is not None else 0, ld_aext=U.ld if U is not None else 0,
                              ld_wext=lora.Rp if lora is not None else 0)
                out.append(ops.gemm(g, keep=(site, dy, dx, U), ws=self.ws))
                for c in range(64, lora.Rp if lora is not None else 0, 64):   # further slices of U . (scale down)
                    g_c = gemm_args(U.ptr + self.eng.esz * c, lora.dn_p.data_ptr() + self.eng.esz * c, dx.ptr, m=rows, n=kin, k=64, lda=U.ld,
                                    ldw=lora.Rp, ldc=dx.ld, residual=dx.ptr, ldr=dx.ld)
                    out.append(ops.gemm(g_c, keep=(lora, U, dx), ws=self.ws))
        else:
            B, ho, wo, hi, wi = conv
            if amode == A_CONV3_S1:
                dmode, dconv, drows = A_CONV3_S1, (B, hi, wi, ho, wo), B * hi * wi
            elif amode == A_CONV3_S2:
                dmode, dconv, drows = A_CONV3_TR2, (B, hi, wi, ho, wo), B * hi * wi
            else:  # UP2: dgrad w.r.t. the upsampled image, then fold 2x2
                dmode, dconv, drows = A_CONV3_S1, (B, ho, wo, ho, wo), B * ho * wo
            dxa = self.act("g." + y.name + ".dx", drows, kin)
            g = gemm_args(dy.ptr, site.wt, dxa.ptr, m=drows, n=kin, k=9 * site.n, lda=dy.ld, ldc=dxa.ld, a_mode=dmode,
                          conv=dconv)
            out.append(ops.gemm(g, keep=(site, dy, dxa), ws=self.ws))
            if lora is not None:
                # conv LoRA: dX += conv_dgrad(U, scale * lora_down) -- U is a 64-channel image, the packed dn_p is
                # the flipped / in-out-swapped [Cin][3][3][64] operand; accumulated through the residual epilogue
                g2 = gemm_args(U.ptr, lora.dn_p, dxa.ptr, m=drows, n=kin, k=9 * lora.Rp, lda=U.ld, ldc=dxa.ld,
                               a_mode=dmode, conv=dconv, residual=dxa.ptr, ldr=dxa.ld)
                out.append(ops.gemm(g2, keep=(lora, U, dxa), ws=self.ws))
            if amode == A_CONV3_UP2:
                dx = self.act("g." + y.name + ".dxlo", B * hi * wi, kin)
                out.append(ops.upsample2x_bwd(dxa.t, dx.t, B, hi, wi, kin))
            else:
                dx = dxa
        c = 0
        for t in xs:
            if t.rg:
                t.gparts.append(dx.cols_view(c, c + t.cols))
            c += t.cols

    # ---- norms ---------------------------------------------------------------------------------
    def groupnorm(self, norm_name: str, x: Union[TRef, Tuple[TRef, TRef]], hw: int, act: int, eps: float, name: str) -> TRef:
        xs = x if isinstance(x, tuple) else (x,)
        gamma, beta = self.eng.norm_p[norm_name]
        Cc = sum(t.cols for t in xs)
        G = self.cfg.norm_num_groups
        rows = xs[0].rows
        y = self.act(name, rows, Cc, rg=any(t.rg for t in xs))
        stats = self.buf(name + ".stats", (self.B * G * 2 * 257,), torch.float32)
        x1 = xs[1] if len(xs) == 2 else None
        if all(t.cstats is not None for t in xs) and (Cc // G) % self.stat_atom == 0 and xs[0].cols % self.stat_atom == 0:
            # the producers left per-(sample, channel) statistics: one apply pass, no reduction over the tensor
            self.both(ops.Op("leco_groupnorm_apply_stats", (
                xs[0].ptr, xs[0].ld, x1.ptr if x1 else None, x1.ld if x1 else 0, xs[0].cols if x1 else 0, xs[0].cstats,
                x1.cstats if x1 else None, self.stat_atom, gamma.data_ptr(), beta.data_ptr(), self.B, hw, Cc, G, eps, act, stats.data_ptr(),
                y.ptr, y.ld), keep=(xs, y, stats, self.stat_arena)))
        else:
            self.both(ops.Op("leco_groupnorm_fwd", (xs[0].ptr, xs[0].ld, x1.ptr if x1 else None, x1.ld if x1 else 0,
                                                    xs[0].cols if x1 else 0, gamma.data_ptr(), beta.data_ptr(), self.B, hw,
                                                    Cc, G, eps, act, stats.data_ptr(), y.ptr, y.ld), keep=(xs, y, stats)))
        if y.rg:
            def bwd():
                out = self.plan.bwd
                dy = self.total_grad(y, out)
                if dy is None:
                    return
                dx = self.act("g." + name + ".dx", rows, Cc)
                bst = self.buf("g." + name + ".bstats", (self.B * G * 2 * 257,), torch.float32)
                out.append(ops.Op("leco_groupnorm_bwd", (xs[0].ptr, xs[0].ld, x1.ptr if x1 else None,
                                                         x1.ld if x1 else 0, xs[0].cols if x1 else 0, dy.ptr, dy.ld,
                                                         gamma.data_ptr(), beta.data_ptr(), stats.data_ptr(), self.B,
                                                         hw, Cc, G, eps, act, bst.data_ptr(), dx.ptr, dx.ld),
                                  keep=(xs, dy, dx)))
                c = 0
                for t in xs:
                    if t.rg:
                        t.gparts.append(dx.cols_view(c, c + t.cols))
                    c += t.cols
            self.tape.append(bwd)
        return y

    def layernorm(self, norm_name: str, x: TRef, name: str) -> TRef:
        gamma, beta = self.eng.norm_p[norm_name]
        y = self.act(name, x.rows, x.cols, rg=x.rg)
        mean = self.buf(name + ".mean", (x.rows,), torch.float32)
        rstd = self.buf(name + ".rstd", (x.rows,), torch.float32)
        self.both(ops.Op("leco_layernorm_fwd", (x.ptr, x.ld, gamma.data_ptr(), beta.data_ptr(), 1e-5, x.rows, x.cols,
                                                y.ptr, y.ld, mean.data_ptr(), rstd.data_ptr()), keep=(x, y)))
        if y.rg:
            def bwd():
                out = self.plan.bwd
                dy = self.total_grad(y, out)
                if dy is None:
                    return
                # fold the other gradient contributions of x (the residual branch) into this kernel
                dres = self.total_grad(x, out)
                dx = self.act("g." + name + ".dx", x.rows, x.cols)
                out.append(ops.Op("leco_layernorm_bwd", (x.ptr, x.ld, dy.ptr, dy.ld, gamma.data_ptr(), mean.data_ptr(),
                                                         rstd.data_ptr(), dres.ptr if dres else None,
                                                         dres.ld if dres else 0, x.rows, x.cols, dx.ptr, dx.ld),
                                  keep=(x, dy, dres, dx)))
                x.gparts = [dx]
            self.tape.append(bwd)
        return y

    # ---- attention -----------------------------------------------------------------------------
    def attention(self, qsrc: TRef, kvsrc: TRef, heads: int, sq: int, skv: int, name: str) -> TRef:
        """Self-attention: ``qsrc is kvsrc`` = the fused [rows][q|k|v] buffer.  Cross-attention:
        ``qsrc`` = [rows][C] queries, ``kvsrc`` = [B*skv][k|v]."""
        fused = qsrc is kvsrc
        Cc = qsrc.cols // 3 if fused else qsrc.cols
        d = Cc // heads
        B = self.B
        q = qsrc.cols_view(0, Cc)
        k = kvsrc.cols_view(Cc, 2 * Cc) if fused else kvsrc.cols_view(0, Cc)
        v = kvsrc.cols_view(2 * Cc, 3 * Cc) if fused else kvsrc.cols_view(Cc, 2 * Cc)
        o = self.act(name, qsrc.rows, Cc, rg=qsrc.rg or kvsrc.rg)
        lse = self.buf(name + ".lse", (B, heads, sq), torch.float32)
        scale = d ** -0.5
        self.both(ops.Op("leco_attention_fwd", (q.ptr, q.ld, sq * q.ld, k.ptr, k.ld, skv * k.ld, v.ptr, v.ld, skv * v.ld,
                                                o.ptr, o.ld, sq * o.ld, lse.data_ptr(), B, heads, sq, skv, d, scale),
                         keep=(qsrc, kvsrc, o)))
        if o.rg:
            def bwd():
                out = self.plan.bwd
                do = self.total_grad(o, out)
                if do is None:
                    return
                delta = self.buf("g." + name + ".delta", (B, heads, sq), torch.float32)
                if fused:
                    dqkv = self.act("g." + name + ".dqkv", qsrc.rows, 3 * Cc)
                    dq, dk, dv = dqkv.cols_view(0, Cc), dqkv.cols_view(Cc, 2 * Cc), dqkv.cols_view(2 * Cc, 3 * Cc)
                    dqs, dkvs = dqkv, dqkv
                else:
                    dqs = self.act("g." + name + ".dq", qsrc.rows, Cc)
                    dkvs = self.act("g." + name + ".dkv", kvsrc.rows, 2 * Cc)
                    dq, dk, dv = dqs, dkvs.cols_view(0, Cc), dkvs.cols_view(Cc, 2 * Cc)
                out.append(ops.Op("leco_attention_bwd", (
                    q.ptr, q.ld, sq * q.ld, k.ptr, k.ld, skv * k.ld, v.ptr, v.ld, skv * v.ld, o.ptr, o.ld, sq * o.ld,
                    do.ptr, do.ld, sq * do.ld, lse.data_ptr(), delta.data_ptr(), dq.ptr, dq.ld, sq * dq.ld,
                    dk.ptr, dk.ld, skv * dk.ld, dv.ptr, dv.ld, skv * dv.ld, B, heads, sq, skv, d, scale),
                    keep=(qsrc, kvsrc, o, do, dqs, dkvs)))
                qsrc.gparts.append(dqs)
                if not fused:
                    kvsrc.gparts.append(dkvs)
            self.tape.append(bwd)
        return o

    # ---- layers --------------------------------------------------------------------------------
    def resnet(self, rname: str, x: Union[TRef, Tuple[TRef, TRef]], hs: int, ws: int) -> TRef:
        eng, m = self.eng, self.eng.named[rname]
        hw, rows = hs * ws, self.B * hs * ws
        conv = (self.B, hs, ws, hs, ws)
        n1 = self.groupnorm(rname + ".norm1", x, hw, ACT_SILU, self.cfg.norm_eps, rname + ".n1")
        off = eng.temb_off[rname]
        temb = self.temb_all  # fp32 [B][temb_total]
        tsite = eng.temb_lora_sites.get(rname)
        temb_T = None
        if tsite is not None:
            # c3lier: this resnet's time_emb_proj has a LoRA -> its slice of the time-embedding bias is recomputed
            # (after the fused projection) by its own GEMM with the K-extension tile
            cout = m.out_channels
            n_on = len(self.f_on)
            self.gemm_fwd(tsite, self.emb_silu, rname + ".temb", rows=self.B,
                          out_f32=temb[:, off:off + cout], ldc32_override=eng.temb_total)
            temb_T = self._last_T
        site1 = eng.sites[rname + ".conv1"]
        h1 = self.gemm_fwd(site1, n1, rname + ".h1", conv=conv, amode=A_CONV3_S1, rows=rows,
                           rowbias=temb.data_ptr() + 4 * off, rows_per_group=hw, ld_rowbias=eng.temb_total, bias=None,
                           stats_hw=hw)
        if tsite is not None and h1.rg:
            def temb_bwd():
                out = self.plan.bwd
                dh = self.total_grad(h1, out)
                if dh is None:
                    return
                cout = m.out_channels
                dsum = self.buf("g." + rname + ".dtemb32", (self.B, cout), torch.float32)
                dtb = self.act("g." + rname + ".dtemb", self.B, cout)
                out.append(ops.Op("leco_rowgroup_sum", (dh.ptr, dh.ld, dsum.data_ptr(), cout, self.B, hw, cout), keep=(dh, dsum)))
                out.append(ops.cast_f32_bf16(dsum, dtb.t, self.B * cout))
                self.lora_bwd(tsite, (self.emb_silu,), dtb, temb_T, None, A_PLAIN, self.B, rname + ".temb")
            self.tape.append(temb_bwd)
        n2 = self.groupnorm(rname + ".norm2", h1, hw, ACT_SILU, self.cfg.norm_eps, rname + ".n2")
        if m.conv_shortcut is not None:
            sc = self.gemm_fwd(eng.sites[rname + ".conv_shortcut"], x, rname + ".sc", rows=rows)
        else:
            assert not isinstance(x, tuple)
            sc = x
        return self.gemm_fwd(eng.sites[rname + ".conv2"], n2, rname + ".out", conv=conv, amode=A_CONV3_S1, rows=rows,
                             residual=sc, stats_hw=hw)

    def shared_prefix_ok(self, ctx: TRef) -> bool:
        """`share` copies of each sample: can the prompt-independent prefix run once per distinct sample?  Needs the stripe tail
        kernel on the first transformer (it re-expands the batch) and a time embedding that does not depend on the sample
        (not SDXL's text_time add-embedding).  LECO_SHARE_PREFIX=0 switches it off (A/B measurements)."""
        import os
        if self.share <= 1 or self.need_bwd or self.eng.f32 or os.environ.get("LECO_SHARE_PREFIX", "1") == "0":
            return False
        if self.cfg.addition_embed_type is not None:
            return False
        blk = self.eng.unet.down_blocks[0]
        if not blk.has_attn or len(self.eng.named["down_blocks.0.attentions.0"].transformer_blocks) != 1:
            return False
        hw = self.h * self.w
        if (self.Bfull // self.share) * hw % 64:
            return False
        return self.stripe_ok("down_blocks.0.attentions.0.transformer_blocks.0", "down_blocks.0.attentions.0",
                              self.cfg.block_out_channels[0], self.cfg.heads(0), hw, ctx.rows // self.Bfull)

    def stripe_ok(self, bname: str, tname: str, Cc: int, heads: int, hw: int, skv: int) -> bool:
        """The row-stripe fused kernels (csrc/stripe.hip) cover this block: forward-only bf16 plan, a supported shape, and
        LoRA operands (if any) in the 32-column packed form.  LECO_STRIPE=0 keeps the per-op launches (A/B measurements)."""
        import os
        if self.need_bwd or self.eng.f32 or os.environ.get("LECO_STRIPE", "1") == "0":
            return False
        if not ops.xblock_supported(Cc, heads, skv, hw):
            return False
        S = self.eng.sites
        names = [bname + n for n in (".attn1.to_out.0", ".attn2.to_q", ".attn2.to_out.0", ".ff.net.0.proj", ".ff.net.2")]
        names += [tname + ".proj_out", tname + ".proj_in", bname + ".attn1.qkv"]
        for nm in names:
            lo = S[nm].lora
            if lo is not None and (lo.Rp != 32 or (nm.endswith("ff.net.0.proj") and lo.up_pg is None)):
                return False
        return S[bname + ".ff.net.0.proj"].geglu_ok

    def _xlin(self, site: GemmSite, lora_on: bool, geglu: bool = False):
        """`leco_xlin` of a site for the stripe kernels: the frozen weight in MFMA fragment order (re-laid once per site),
        the LoRA operand images as `leco_lora_pack` leaves them."""
        key = "_xw_g" if geglu else "_xw"
        if not hasattr(site, key):
            w, b = site.w_geglu if geglu else (site.w, site.bias)
            setattr(site, key, (hip.pack_fragments(w), b))
        w, b = getattr(site, key)
        lo = site.lora if lora_on else None
        if lo is None:
            return hip.xlin(w, b, packed=True), (site, w, b)
        up = lo.up_pg if geglu else lo.up_p
        return hip.xlin(w, b, lo.dn_s, up, lo.R16, ld_up=lo.Rp, packed=True), (site, w, b, lo)

    def block_tail_fused(self, bname: str, tname: str, a1: TRef, hcur: TRef, kv: TRef, heads: int, hw: int, last: bool,
                         x_res: Optional[TRef]) -> TRef:
        """Everything of the block after its self-attention core (+ proj_out and the Transformer2DModel residual when the
        block is the last one) as ONE launch per list (LoRA on / off): `leco_xblock_tail`."""
        eng, S = self.eng, self.eng.sites
        Bt = self.Bfull            # (the block's inputs may exist once per distinct sample: `shared_prefix`)
        rows, Cc = Bt * hw, hcur.cols
        skv = kv.rows // Bt
        d = Cc // heads
        kp, vt = ops.xattn_buffers(Bt, heads, d, self.dev)
        self.plan.bufs[bname + ".xattn_kp"], self.plan.bufs[bname + ".xattn_vt"] = kp, vt
        prep = ops.xattn_prep(kv.ptr, kv.ld, kp, vt, Bt, heads, skv, d)
        prep.tag = "ctx"
        self.both(prep)
        out = self.act((tname + ".out") if last else (bname + ".h3"), rows, Cc)
        if last:
            out.cstats = self.stat_slice(Cc, hw, batch=Bt) if self.stat_atom % 2 == 0 else None
        g2, b2 = eng.norm_p[bname + ".norm2"]
        g3, b3 = eng.norm_p[bname + ".norm3"]
        for lora_on, lst in ((True, self.f_on), (False, self.f_off)):
            A = hip.XBlockTailArgs()
            keep = [a1, hcur, kv, out, kp, vt, g2, b2, g3, b3, x_res, self.stat_arena]
            A.m, A.c, A.heads, A.skv, A.rows_per_sample = rows, Cc, heads, skv, hw
            if a1.rows != rows:        # batch-shared prefix: the block's inputs exist once per distinct sample
                assert hcur.rows == a1.rows and (x_res is None or x_res.rows == a1.rows) and rows % a1.rows == 0
                A.src_rows = a1.rows
            A.attn, A.ld_attn, A.h_in, A.ld_h = a1.ptr, a1.ld, hcur.ptr, hcur.ld
            for fld, nm, gg in (("to_out1", bname + ".attn1.to_out.0", False), ("to_q2", bname + ".attn2.to_q", False),
                                ("to_out2", bname + ".attn2.to_out.0", False), ("ff1", bname + ".ff.net.0.proj", True),
                                ("ff2", bname + ".ff.net.2", False)) + ((("proj_out", tname + ".proj_out", False),) if last else ()):
                xl, kp_ = self._xlin(S[nm], lora_on, gg)
                setattr(A, fld, xl)
                keep.append(kp_)
            A.ln2_g, A.ln2_b, A.ln3_g, A.ln3_b, A.ln_eps = g2.data_ptr(), b2.data_ptr(), g3.data_ptr(), b3.data_ptr(), 1e-5
            A.kp, A.vt, A.attn_scale = kp.data_ptr(), vt.data_ptr(), d ** -0.5
            if last:
                A.res, A.ld_res = x_res.ptr, x_res.ld
            A.out, A.ld_out = out.ptr, out.ld
            if out.cstats is not None:
                A.col_stats, A.stats_atom = out.cstats, self.stat_atom
            lst.append(ops.xblock_tail(A, keep=keep))
        return out

    def ln_linear(self, norm_name: str, site: GemmSite, x: TRef, ln_name: str, name: str, rows: int, geglu: bool = False) -> TRef:
        """y = Linear(LayerNorm(x)) [GEGLU]: two launches.  (A one-launch form -- LayerNorm folded into the Linear through
        the row statistics of its own operands, rounds 2 and 5 -- measured step-neutral to slightly negative on SD1.5 and
        neutral on SDXL, profiles/r05_bench_lnfold_*.json / r06_switch_ab_sdxl.txt, and was removed in round 6.)"""
        return self.gemm_fwd(site, self.layernorm(norm_name, x, ln_name), name, rows=rows, geglu=geglu)

    def basic_block(self, bname: str, hcur: TRef, ctx: TRef, heads: int, hw: int, tname: str = "", last: bool = False,
                    x_res: Optional[TRef] = None, qkv: Optional[TRef] = None) -> Tuple[TRef, bool]:
        """Returns (output, done): `done` = the output already is the Transformer2DModel's (proj_out + residual applied by the
        fused tail kernel).  ``qkv``: already computed by the fused head kernel (`block_head_fused`)."""
        eng = self.eng
        rows, Cc = hcur.rows, hcur.cols
        S = eng.sites
        fused = bool(tname) and self.stripe_ok(bname, tname, Cc, heads, hw, ctx.rows // self.Bfull)
        if qkv is None:
            qkv = self.ln_linear(bname + ".norm1", S[bname + ".attn1.qkv"], hcur, bname + ".l1", bname + ".qkv", rows)
        a1 = self.attention(qkv, qkv, heads, hw, hw, bname + ".a1")
        if fused:
            n_on, n_off = len(self.f_on), len(self.f_off)
            kv = self.gemm_fwd(S[bname + ".attn2.kv"], ctx, bname + ".kv", rows=ctx.rows)
            for op in self.f_on[n_on:] + self.f_off[n_off:]:
                op.tag = "ctx"
            return self.block_tail_fused(bname, tname, a1, hcur, kv, heads, hw, last, x_res), last
        h1 = self.gemm_fwd(S[bname + ".attn1.to_out.0"], a1, bname + ".h1", rows=rows, residual=hcur)
        q2 = self.ln_linear(bname + ".norm2", S[bname + ".attn2.to_q"], h1, bname + ".l2", bname + ".q2", rows)
        # K/V of cross-attention depend only on the prompt embeddings (and the LoRA weights): tag their ops
        # so that callers replaying the same prompt (the k denoising passes of a step) can run them once
        n_on, n_off = len(self.f_on), len(self.f_off)
        kv = self.gemm_fwd(S[bname + ".attn2.kv"], ctx, bname + ".kv", rows=ctx.rows)
        for op in self.f_on[n_on:] + self.f_off[n_off:]:
            op.tag = "ctx"
        a2 = self.attention(q2, kv, heads, hw, ctx.rows // self.B, bname + ".a2")
        h2 = self.gemm_fwd(S[bname + ".attn2.to_out.0"], a2, bname + ".h2", rows=rows, residual=h1)
        ff1 = S[bname + ".ff.net.0.proj"]
        if not self.need_bwd and not eng.f32 and ff1.geglu_ok and (ff1.lora is None or (ff1.lora.Rp == 32 and ff1.lora.up_pg is not None)):
            gg = self.ln_linear(bname + ".norm3", ff1, h2, bname + ".l3", bname + ".geglu", rows, geglu=True)
            return self.gemm_fwd(S[bname + ".ff.net.2"], gg, bname + ".h3", rows=rows, residual=h2), False
        l3 = self.layernorm(bname + ".norm3", h2, bname + ".l3")
        u = self.gemm_fwd(ff1, l3, bname + ".u", rows=rows)
        gg = self.act(bname + ".geglu", rows, 4 * Cc, rg=u.rg)
        self.both(ops.Op("leco_geglu_fwd", (u.ptr, u.ld, gg.ptr, gg.ld, rows, 4 * Cc), keep=(u, gg)))
        if gg.rg:
            def bwd():
                out = self.plan.bwd
                dg = self.total_grad(gg, out)
                if dg is None:
                    return
                du = self.act("g." + bname + ".du", rows, 8 * Cc)
                out.append(ops.Op("leco_geglu_bwd", (u.ptr, u.ld, dg.ptr, dg.ld, du.ptr, du.ld, rows, 4 * Cc),
                                  keep=(u, dg, du)))
                u.gparts.append(du)
            self.tape.append(bwd)
        return self.gemm_fwd(S[bname + ".ff.net.2"], gg, bname + ".h3", rows=rows, residual=h2), False

    def block_head_fused(self, tname: str, bname: str, x: TRef, hw: int) -> Tuple[TRef, TRef]:
        """GroupNorm (from the statistics the producer of x left, else the per-op GroupNorm first) + proj_in + norm1 + q|k|v of
        the first block as ONE launch per list: `leco_xblock_head`.  Returns (residual stream, qkv)."""
        eng, S = self.eng, self.eng.sites
        rows, Cc = x.rows, x.cols
        G = self.cfg.norm_num_groups
        use_stats = x.cstats is not None and (Cc // G) % self.stat_atom == 0
        src = x if use_stats else self.groupnorm(tname + ".norm", x, hw, ACT_NONE, 1e-6, tname + ".n")
        p = self.act(tname + ".pin", rows, Cc)
        qkv = self.act(bname + ".qkv", rows, 3 * Cc)
        gg, gb = eng.norm_p[tname + ".norm"]
        lg, lb = eng.norm_p[bname + ".norm1"]
        for lora_on, lst in ((True, self.f_on), (False, self.f_off)):
            A = hip.XBlockHeadArgs()
            keep = [x, src, p, qkv, gg, gb, lg, lb, self.stat_arena]
            A.m, A.c, A.rows_per_sample = rows, Cc, hw
            A.x, A.ld_x = src.ptr, src.ld
            if use_stats:
                A.gn_cstats, A.stats_atom, A.groups = x.cstats, self.stat_atom, G
                A.gn_g, A.gn_b, A.gn_eps = gg.data_ptr(), gb.data_ptr(), 1e-6
            for fld, nm in (("proj_in", tname + ".proj_in"), ("qkv", bname + ".attn1.qkv")):
                xl, kp_ = self._xlin(S[nm], lora_on)
                setattr(A, fld, xl)
                keep.append(kp_)
            A.ln1_g, A.ln1_b, A.ln_eps = lg.data_ptr(), lb.data_ptr(), 1e-5
            A.h_out, A.ld_hout, A.qkv_out, A.ld_qkv = p.ptr, p.ld, qkv.ptr, qkv.ld
            lst.append(ops.xblock_head(A, keep=keep))
        return p, qkv

    def transformer(self, tname: str, x: TRef, ctx: TRef, level: int, hs: int, ws: int) -> TRef:
        eng, m = self.eng, self.eng.named[tname]
        hw, rows = hs * ws, self.B * hs * ws
        nb = len(m.transformer_blocks)
        b0 = f"{tname}.transformer_blocks.0"
        qkv0 = None
        import os
        if self.stripe_ok(b0, tname, x.cols, self.cfg.heads(level), hw, ctx.rows // self.Bfull) and not isinstance(x, tuple) \
                and os.environ.get("LECO_STRIPE_HEAD", "1") != "0":
            p, qkv0 = self.block_head_fused(tname, b0, x, hw)
        else:
            n = self.groupnorm(tname + ".norm", x, hw, ACT_NONE, 1e-6, tname + ".n")
            p = self.gemm_fwd(eng.sites[tname + ".proj_in"], n, tname + ".pin", rows=rows)
        for i in range(nb):
            p, done = self.basic_block(f"{tname}.transformer_blocks.{i}", p, ctx, self.cfg.heads(level), hw, tname=tname,
                                       last=i == nb - 1, x_res=x, qkv=qkv0 if i == 0 else None)
            if done:
                return p
        return self.gemm_fwd(eng.sites[tname + ".proj_out"], p, tname + ".out", rows=rows, residual=x, stats_hw=hw)

    # ---- whole network ---------------------------------------------------------------------------
    def build(self) -> Plan:
        eng, cfg, B, h, w, dev = self.eng, self.cfg, self.B, self.h, self.w, self.dev
        P = self.plan
        ch = cfg.block_out_channels
        S = eng.sites
        xl = cfg.addition_embed_type == "text_time"
        # -- dynamic inputs
        P.x_in = self.buf("x_in", (B, cfg.in_channels, h, w))
        P.ctx = self.buf("ctx", (B, 77, cfg.cross_attention_dim))
        P.t_table = self.buf("t_table", (1024,), torch.float32, zero=True)  # timesteps; entry t_idx is used
        P.t_idx = self.buf("t_idx", (1,), torch.int32, zero=True)
        P.pred = self.buf("pred", (B, cfg.out_channels, h, w), torch.float32)
        P.dpred = self.buf("dpred", (B, cfg.out_channels, h, w), torch.float32, zero=True)
        ctx = TRef(P.ctx, B * 77, cfg.cross_attention_dim, name="ctx")
        # -- time embedding
        tsin = self.act("t_sin", B, ch[0])
        self.both(ops.timestep_embedding(P.t_table, P.t_idx, 0, B, ch[0], tsin.t))
        e1 = self.gemm_fwd(S["time_embedding.linear_1"], tsin, "t_e1", rows=B, act=ACT_SILU)
        if not xl:
            emb_silu = self.gemm_fwd(S["time_embedding.linear_2"], e1, "t_emb_silu", rows=B, act=ACT_SILU)
        else:
            emb = self.gemm_fwd(S["time_embedding.linear_2"], e1, "t_emb", rows=B)
            P.text_embeds = self.buf("text_embeds", (B, cfg.projection_class_embeddings_input_dim - 6 * cfg.addition_time_embed_dim))
            P.time_ids = self.buf("time_ids", (B * 6,), torch.float32, zero=True)
            asin = self.act("add_sin", B, 6 * cfg.addition_time_embed_dim)
            self.both(ops.timestep_embedding(P.time_ids, None, 1, B * 6, cfg.addition_time_embed_dim, asin.t))
            te = TRef(P.text_embeds, B, P.text_embeds.shape[1], name="text_embeds")
            a1 = self.gemm_fwd(S["add_embedding.linear_1"], (te, asin), "add_e1", rows=B, act=ACT_SILU)
            emb_silu = self.gemm_fwd(S["add_embedding.linear_2"], a1, "t_emb_silu", rows=B, residual=emb, act=ACT_SILU)
        self.emb_silu = emb_silu
        self.temb_all = self.buf("temb_all", (B, eng.temb_total), torch.float32)
        self.gemm_fwd(S["time_emb_proj_all"], emb_silu, "temb_all_g", rows=B, out_f32=self.temb_all)
        # -- conv_in.  BATCH-SHARED PREFIX (share > 1: the B latents are `share` copies of B / share samples at one timestep):
        # conv_in, the first ResnetBlock2D and the first Transformer2DModel up to and including its self-attention do not see
        # the prompt, so they run ONCE per distinct sample (batch Bp); the stripe tail kernel of that transformer reads them
        # with a row wrap (leco_xblock_tail_args.src_rows) and produces the full batch, conv_in's output -- a skip connection
        # of the last up block -- is replicated by one copy.  Same arithmetic per sample: the result does not change.
        Bp = B // self.share if self.shared_prefix_ok(ctx) else B
        self.B = Bp
        h0 = self.act("conv_in", Bp * h * w, ch[0])
        self.both(ops.conv_in(P.x_in, eng.conv_in_w, eng.conv_in_b, h0.t, Bp, h, w, cfg.in_channels, ch[0]))
        h0.cstats = self.stat_slice(ch[0], h * w)
        if h0.cstats is not None:   # conv_in has no statistics epilogue: one light pass over its output
            self.both(ops.Op("leco_colstats", (h0.ptr, h0.ld, h0.cstats, self.stat_atom, Bp, h * w, ch[0]), keep=(h0, self.stat_arena)))
        cur, hs, ws = h0, h, w
        if Bp != B:
            full = self.act("conv_in.rep", B * h * w, ch[0])
            self.both(ops.repeat(h0.ptr, full.ptr, Bp * h * w * ch[0] * eng.esz, self.share, keep=(h0, full)))
            if h0.cstats is not None:
                nst = 2 * Bp * (ch[0] // self.stat_atom) * 4
                full.cstats = self.stat_slice(ch[0], h * w, batch=B) if nst % 16 == 0 else None
                if full.cstats is not None:
                    self.both(ops.repeat(h0.cstats, full.cstats, nst, self.share, keep=(self.stat_arena,)))
            skips = [full]
        else:
            skips = [h0]
        for i, blk in enumerate(eng.unet.down_blocks):
            bn = f"down_blocks.{i}"
            for j in range(len(blk.resnets)):
                cur = self.resnet(f"{bn}.resnets.{j}", cur, hs, ws)
                if blk.has_attn:
                    cur = self.transformer(f"{bn}.attentions.{j}", cur, ctx, i, hs, ws)
                self.B = B          # (the first transformer's tail has produced the full batch)
                assert cur.rows == B * hs * ws
                skips.append(cur)
            if blk.downsamplers is not None:
                ho, wo = (hs + 1) // 2, (ws + 1) // 2
                cur = self.gemm_fwd(S[f"{bn}.downsamplers.0.conv"], cur, f"{bn}.down", conv=(B, ho, wo, hs, ws),
                                    amode=A_CONV3_S2, rows=B * ho * wo, stats_hw=ho * wo)
                hs, ws = ho, wo
                skips.append(cur)
        lvl = len(ch) - 1
        cur = self.resnet("mid_block.resnets.0", cur, hs, ws)
        cur = self.transformer("mid_block.attentions.0", cur, ctx, lvl, hs, ws)
        cur = self.resnet("mid_block.resnets.1", cur, hs, ws)
        for i, blk in enumerate(eng.unet.up_blocks):
            bn = f"up_blocks.{i}"
            level = len(ch) - 1 - i
            for j in range(len(blk.resnets)):
                skip = skips.pop()
                cur = self.resnet(f"{bn}.resnets.{j}", (cur, skip), hs, ws)
                if blk.has_attn:
                    cur = self.transformer(f"{bn}.attentions.{j}", cur, ctx, level, hs, ws)
            if blk.upsamplers is not None:
                cur = self.gemm_fwd(S[f"{bn}.upsamplers.0.conv"], cur, f"{bn}.up", conv=(B, 2 * hs, 2 * ws, hs, ws),
                                    amode=A_CONV3_UP2, rows=B * 4 * hs * ws, stats_hw=4 * hs * ws)
                hs, ws = 2 * hs, 2 * ws
        assert (hs, ws) == (h, w), "latent size must be divisible by the down-sampling factor"
        nout = self.groupnorm("conv_norm_out", cur, h * w, ACT_SILU, cfg.norm_eps, "norm_out")
        self.both(ops.conv_out(nout.t, eng.conv_out_w, eng.conv_out_b, P.pred, B, h, w, ch[0], cfg.out_channels))
        P.final = nout
        # -- backward: conv_out dgrad seeds the tape
        if nout.rg and self.need_bwd:
            dn = self.act("g.norm_out", B * h * w, ch[0])
            P.bwd.append(ops.conv_out_bwd(P.dpred, eng.conv_out_w, dn.t, B, h, w, ch[0], cfg.out_channels))
            nout.gparts.append(dn)
            for fn in reversed(self.tape):
                fn()
            grouped = ops.lora_wgrad_grouped(self.wgrad_problems, self.dev)
            if grouped is not None:
                grouped.keep = (grouped.keep, self.wgrad_keep)
                P.bwd.append(grouped)
        if self.gn_fused and self.stat_used:
            zero = ops.memset(self.stat_arena[:self.stat_used])
            zero.keep = (self.stat_arena,)
            self.f_on.insert(0, zero)
            self.f_off.insert(0, zero)
        self.tape = []
        return P


# =============================================================================================
# public module
# =============================================================================================
_GRAPH_API_DECLARED = False


def _graph_api():
    global _GRAPH_API_DECLARED
    lib = hip.lib()
    if not _GRAPH_API_DECLARED:
        for nm, at in [("leco_graph_begin_capture", [C.c_void_p]),
                       ("leco_graph_end_capture", [C.c_void_p, C.POINTER(C.c_void_p)]),
                       ("leco_graph_launch", [C.c_void_p, C.c_void_p]), ("leco_graph_destroy", [C.c_void_p])]:
            getattr(lib, nm).argtypes = at
            getattr(lib, nm).restype = C.c_int
        _GRAPH_API_DECLARED = True
    return lib


class _UNetFn(torch.autograd.Function):
    """Ties a plan's forward/backward launch lists into torch autograd so that the reference's
    ``loss.backward()`` (train_lora.py:279) works unchanged: the LoRA slab is the only input that
    receives a gradient."""

    @staticmethod
    def forward(ctx, slab, unet, plan, dtype):
        ctx.unet, ctx.plan = unet, plan
        unet._run(plan, "fwd_on")
        return plan.pred.to(dtype)

    @staticmethod
    def backward(ctx, gout):
        unet, plan = ctx.unet, ctx.plan
        net = unet._engine.network
        plan.dpred.copy_(gout.reshape(plan.dpred.shape))
        if net.grads_cleared():      # optimizer.zero_grad(set_to_none=True) semantics; else accumulate
            net.grad.zero_()
        unet._run(plan, "bwd")
        net.attach_grads()           # per-parameter .grad views into the flat gradient slab
        net.mark_updated()           # an optimizer step follows: re-pack before the next LoRA-on pass
        return None, None, None, None


class UNet2DConditionModel(nn.Module):
    """Drop-in for the object ``model_util.load_models`` returns (model_util.py:104-129): callable as
    ``unet(sample, timestep, encoder_hidden_states=..., [added_cond_kwargs=...]).sample``
    (train_util.py:156-160, 239-244)."""

    def __init__(self, cfg: Optional[UNetConfig] = None):
        super().__init__()
        cfg = cfg or UNetConfig()
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
            attn = t.startswith("CrossAttn")
            self.down_blocks.append((CrossAttnDownBlock2D if attn else DownBlock2D)(
                cfg, i, cin, cout, temb, attn, i == len(ch) - 1))
        self.mid_block = UNetMidBlock2DCrossAttn(cfg, ch[-1], temb)
        rev = list(reversed(ch))
        cout = rev[0]
        for i, t in enumerate(cfg.up_block_types):
            prev, cout = cout, rev[i]
            attn = t.startswith("CrossAttn")
            self.up_blocks.append((CrossAttnUpBlock2D if attn else UpBlock2D)(
                cfg, len(ch) - 1 - i, prev, cout, rev[min(i + 1, len(ch) - 1)], temb, attn, i == len(ch) - 1))
        self.conv_norm_out = nn.GroupNorm(cfg.norm_num_groups, ch[0], eps=cfg.norm_eps)
        self.conv_out = nn.Conv2d(ch[0], cfg.out_channels, 3, 1, 1)
        self._engine: Optional[Engine] = None
        self.use_graphs = False

    # ---- reference call sites that are no-ops here (train_lora.py:68) ------------------------------
    def enable_xformers_memory_efficient_attention(self, *a, **k):
        return None

    @property
    def dtype(self):
        return self.conv_in.weight.dtype

    @property
    def device(self):
        return self.conv_in.weight.device

    def engine(self) -> Engine:
        if self._engine is None:
            self._engine = Engine(self, self.device)
        return self._engine

    def release(self) -> None:
        """Free the launch plans (hipGraph execs + activation buffers) this model has built."""
        if self._engine is not None:
            self._engine.release()

    # ---- execution ----------------------------------------------------------------------------------
    def _run(self, plan: Plan, which: str) -> None:
        oplist = plan.lists[which]
        if not (self.use_graphs and self.device.type == "cuda" and not hip.is_emulated()) or ops._TRACE_OPS:
            ops.run_plan(oplist)
            return
        import os
        eager = os.environ.get("LECO_EAGER_LISTS")      # debugging aid: comma list of "<list name>[:B]" launched eagerly
        if eager and any(e == which.split("@")[0] or e == f"{which.split('@')[0]}:{plan.key[0]}" for e in eager.split(",")):
            ops.run_plan(oplist)
            return
        lib = _graph_api()
        g = plan.graphs.get(which)
        cur = torch.cuda.current_stream()
        if g is None:
            import os
            side = getattr(self, "_capture_stream", None)
            if side is None:
                side = self._capture_stream = torch.cuda.Stream()
            side.wait_stream(cur)
            sp = side.cuda_stream
            hip.check(lib.leco_graph_begin_capture(sp), "graph begin")
            try:
                ops.run_plan(oplist, sp)
            finally:
                gh = C.c_void_p()
                hip.check(lib.leco_graph_end_capture(sp, C.byref(gh)), "graph end")
            g = plan.graphs[which] = gh
        hip.check(lib.leco_graph_launch(g, cur.cuda_stream), "graph launch")

    def prepare(self, sample_shape, lora_on: bool, tag: Optional[str] = None) -> Plan:
        B, _, h, w = sample_shape
        eng = self.engine()
        net = eng.network
        if net is not None and lora_on and (net.needs_repack() or eng._pack_scale != net.multiplier):
            net.sync_shadow()
            eng.refresh_lora(net.multiplier)
        return eng.plan(B, h, w, tag=tag)

    def lora_active(self) -> bool:
        net = self.engine().network
        return net is not None and net.multiplier != 0

    def _adopt_foreign_patches(self) -> None:
        """The reference's LoRAModule (lora.py:97-100) works by re-assigning ``org_module.forward``.  The leaves here only
        HOLD weights -- the launch plans never call ``leaf.forward`` -- so such a network is ADOPTED instead
        (`lora.adopt_forward_patches`: its parameters become slab views, its products run fused in the GEMMs).  The leaf list
        is cached; every forward checks the leaves' instance dicts (cheap) -- also AFTER a network has been attached: a patch
        that is not part of the attached network would be silently ignored by the launch plans, so it raises."""
        leaves = self.__dict__.get("_leaf_cache")
        if leaves is None:
            leaves = self.__dict__["_leaf_cache"] = [(n, m) for n, m in self.named_modules() if isinstance(m, (nn.Linear, nn.Conv2d))]
        patched = [(n, m) for n, m in leaves if "forward" in m.__dict__]
        net = self.engine().network
        if net is None:
            if patched:
                from .lora import adopt_forward_patches
                adopt_forward_patches(self)
            return
        if patched:
            owned = {id(l.fm) for l in net.unet_loras if getattr(l, "fm", None) is not None}
            for n, m in patched:
                fm = getattr(m.__dict__["forward"], "__self__", None)     # a plain function / closure has no __self__
                if fm is None or id(fm) not in owned:
                    raise RuntimeError(f"{n}.forward was re-assigned after a LoRA network had been attached, by something that is "
                                       "not part of that network: the launch plans would silently ignore it.")

    def forward(self, sample, timestep, encoder_hidden_states=None, added_cond_kwargs=None):
        self._adopt_foreign_patches()
        eng = self.engine()
        lora_on = self.lora_active()
        plan = self.prepare(sample.shape, lora_on)
        plan.x_in.copy_(sample)
        plan.set_ctx(encoder_hidden_states)
        t = torch.as_tensor(timestep)
        plan.t_table[:1].copy_(t.reshape(-1)[:1].to(torch.float32))
        plan.t_idx.zero_()
        if self.cfg.addition_embed_type == "text_time":
            plan.text_embeds.copy_(added_cond_kwargs["text_embeds"])
            plan.time_ids.copy_(added_cond_kwargs["time_ids"].reshape(-1).to(torch.float32))
        net = eng.network
        if lora_on and torch.is_grad_enabled() and net is not None and net.slab.requires_grad:
            return UNetOutput(_UNetFn.apply(net.slab, self, plan, sample.dtype))
        self._run(plan, "fwd_on" if lora_on else "fwd_off")
        return UNetOutput(plan.pred.to(sample.dtype))
