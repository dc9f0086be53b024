"""Single-file Stable-Diffusion checkpoints (LDM / CompVis key layout) -> the diffusers key layout this package's
`UNet2DConditionModel` uses.

Replaces `StableDiffusionPipeline.from_single_file(...)` as the reference calls it (`model_util.py:75-101`, selected
by the `.ckpt` / `.safetensors` suffix test at `model_util.py:111-117`).  Only the pieces the LECO path needs are
converted: the UNet (`model.diffusion_model.*`) and the CLIP text encoder (`cond_stage_model.*`); the VAE is skipped
exactly like the reference ("VAE はいらない", `model_util.py:120`).

The LDM UNet is a flat list of blocks; the mapping below is derived from the architecture, not from a table:
    input_blocks.0.0                      conv_in
    input_blocks.{i}.0 / .1               down_blocks.{b}.resnets.{j} / .attentions.{j}
    input_blocks.{i}.0.op                 down_blocks.{b}.downsamplers.0.conv
    middle_block.0 / .1 / .2              mid_block.resnets.0 / attentions.0 / resnets.1
    output_blocks.{i}.0 / .1              up_blocks.{b}.resnets.{j} / .attentions.{j}
    output_blocks.{i}.{1|2}.conv          up_blocks.{b}.upsamplers.0.conv
    time_embed.0 / .2                     time_embedding.linear_1 / linear_2
    label_emb.0.0 / .2                    add_embedding.linear_1 / linear_2          (SDXL)
    out.0 / out.2                         conv_norm_out / conv_out
    ResBlock: in_layers.0, in_layers.2, emb_layers.1, out_layers.0, out_layers.3, skip_connection
          ->  norm1,       conv1,       time_emb_proj, norm2,        conv2,        conv_shortcut
    SpatialTransformer: identical inner names (norm, proj_in, transformer_blocks.*, proj_out).
"""
from __future__ import annotations

import re
from typing import Dict, Optional, Tuple

import torch

from .unet import UNetConfig, sd15_config, sd21_config, sdxl_config

UNET_PREFIX = "model.diffusion_model."
RESNET_MAP = {"in_layers.0": "norm1", "in_layers.2": "conv1", "emb_layers.1": "time_emb_proj",
              "out_layers.0": "norm2", "out_layers.3": "conv2", "skip_connection": "conv_shortcut"}


def unet_module_map(cfg: UNetConfig) -> Dict[str, Tuple[str, str]]:
    """LDM module prefix -> (diffusers module prefix, kind) with kind in {'plain', 'resnet', 'attn'}."""
    m: Dict[str, Tuple[str, str]] = {
        "time_embed.0": ("time_embedding.linear_1", "plain"), "time_embed.2": ("time_embedding.linear_2", "plain"),
        "input_blocks.0.0": ("conv_in", "plain"),
        "middle_block.0": ("mid_block.resnets.0", "resnet"), "middle_block.1": ("mid_block.attentions.0", "attn"),
        "middle_block.2": ("mid_block.resnets.1", "resnet"),
        "out.0": ("conv_norm_out", "plain"), "out.2": ("conv_out", "plain"),
    }
    if cfg.addition_embed_type == "text_time":
        m["label_emb.0.0"] = ("add_embedding.linear_1", "plain")
        m["label_emb.0.2"] = ("add_embedding.linear_2", "plain")
    i = 1
    nb = len(cfg.down_block_types)
    for b, btype in enumerate(cfg.down_block_types):
        for j in range(cfg.layers_per_block):
            m[f"input_blocks.{i}.0"] = (f"down_blocks.{b}.resnets.{j}", "resnet")
            if btype.startswith("CrossAttn"):
                m[f"input_blocks.{i}.1"] = (f"down_blocks.{b}.attentions.{j}", "attn")
            i += 1
        if b != nb - 1:
            m[f"input_blocks.{i}.0.op"] = (f"down_blocks.{b}.downsamplers.0.conv", "plain")
            i += 1
    i = 0
    for b, btype in enumerate(cfg.up_block_types):
        for j in range(cfg.layers_per_block + 1):
            m[f"output_blocks.{i}.0"] = (f"up_blocks.{b}.resnets.{j}", "resnet")
            nxt = 1
            if btype.startswith("CrossAttn"):
                m[f"output_blocks.{i}.1"] = (f"up_blocks.{b}.attentions.{j}", "attn")
                nxt = 2
            if j == cfg.layers_per_block and b != nb - 1:
                m[f"output_blocks.{i}.{nxt}.conv"] = (f"up_blocks.{b}.upsamplers.0.conv", "plain")
            i += 1
    return m


def _translate(key: str, modmap: Dict[str, Tuple[str, str]], reverse: bool) -> Optional[str]:
    """One parameter name across the two layouts (None: not a UNet parameter of this architecture)."""
    table = {v[0]: (k, v[1]) for k, v in modmap.items()} if reverse else modmap
    # longest matching module prefix wins ("input_blocks.3.0.op" before "input_blocks.3.0")
    for pre in sorted(table, key=len, reverse=True):
        if key == pre or key.startswith(pre + "."):
            dst, kind = table[pre]
            rest = key[len(pre):]
            if kind == "resnet":
                sub = {v: k for k, v in RESNET_MAP.items()} if reverse else RESNET_MAP
                for a, b in sub.items():
                    if rest.startswith("." + a + "."):
                        return dst + "." + b + rest[len(a) + 1:]
                return None
            return dst + rest
    return None


def convert_ldm_unet(sd: Dict[str, torch.Tensor], cfg: UNetConfig) -> Dict[str, torch.Tensor]:
    """`model.diffusion_model.*` entries of an LDM checkpoint -> diffusers-named UNet state dict."""
    modmap = unet_module_map(cfg)
    out, unknown = {}, []
    for k, v in sd.items():
        if not k.startswith(UNET_PREFIX):
            continue
        nk = _translate(k[len(UNET_PREFIX):], modmap, reverse=False)
        if nk is None:
            unknown.append(k)
        else:
            out[nk] = v
    if unknown:
        raise KeyError(f"checkpoint has UNet keys that do not fit the detected architecture, e.g. {unknown[:4]}")
    return out


def diffusers_unet_to_ldm(sd: Dict[str, torch.Tensor], cfg: UNetConfig) -> Dict[str, torch.Tensor]:
    """Inverse of `convert_ldm_unet` (used to write single-file checkpoints, and by the round-trip test)."""
    modmap = unet_module_map(cfg)
    out = {}
    for k, v in sd.items():
        nk = _translate(k, modmap, reverse=True)
        if nk is None:
            raise KeyError(f"no LDM name for UNet parameter {k}")
        out[UNET_PREFIX + nk] = v
    return out


def detect_unet_config(sd: Dict[str, torch.Tensor]) -> UNetConfig:
    """Architecture of a single-file checkpoint from its tensor shapes: SDXL has `label_emb`; SD2.x has a
    1024-wide cross-attention context and Linear proj_in; everything else is SD1.x.  Non-standard widths (the
    reduced architectures the tests use) are reconstructed from the shapes."""
    p = UNET_PREFIX
    if p + "input_blocks.0.0.weight" not in sd:
        raise KeyError("not an LDM-layout checkpoint: model.diffusion_model.input_blocks.0.0.weight missing")
    xl = any(k.startswith(p + "label_emb.") for k in sd)
    kv = [v for k, v in sd.items() if k.startswith(p) and k.endswith("attn2.to_k.weight")]
    ctx = int(kv[0].shape[1]) if kv else 768
    pin = [v for k, v in sd.items() if k.startswith(p) and k.endswith(".proj_in.weight")]
    linear = bool(pin) and pin[0].ndim == 2
    c0 = int(sd[p + "input_blocks.0.0.weight"].shape[0])
    if c0 == 320 and xl and ctx == 2048:
        return sdxl_config()
    if c0 == 320 and not xl and ctx == 1024 and linear:
        return sd21_config()
    if c0 == 320 and not xl and ctx == 768 and not linear:
        return sd15_config()
    return _config_from_shapes(sd, xl, ctx, linear)


def _config_from_shapes(sd, xl: bool, ctx: int, linear: bool) -> UNetConfig:
    """Generic reconstruction (block widths, attention placement, transformer depth) for non-standard sizes."""
    p = UNET_PREFIX
    idx = sorted({int(m.group(1)) for k in sd for m in [re.match(re.escape(p) + r"input_blocks\.(\d+)\.", k)] if m})
    widths, attn, depth, levels = [], [], [], []
    cur = None
    for i in idx[1:]:
        if p + f"input_blocks.{i}.0.op.weight" in sd:
            levels.append(cur)
            cur = None
            continue
        w = int(sd[p + f"input_blocks.{i}.0.out_layers.3.weight"].shape[0])
        has = any(k.startswith(p + f"input_blocks.{i}.1.") for k in sd)
        d = len({m.group(1) for k in sd
                 for m in [re.match(re.escape(p) + rf"input_blocks\.{i}\.1\.transformer_blocks\.(\d+)\.", k)] if m})
        cur = (w, has, max(d, 1), (cur[3] + 1) if cur else 1)
    levels.append(cur)
    for w, has, d, n in levels:
        widths.append(w); attn.append(has); depth.append(d)
    lpb = levels[0][3]
    heads = []
    for lvl, w in enumerate(widths):   # head count is not recoverable from shapes: SD1-style 8 heads, else d_head 64
        heads.append(8 if (not linear and not xl) else max(1, w // 64))
    te = int(sd[p + "time_embed.0.weight"].shape[0])
    kw = dict(in_channels=int(sd[p + "input_blocks.0.0.weight"].shape[1]), out_channels=int(sd[p + "out.2.weight"].shape[0]),
              block_out_channels=tuple(widths),
              down_block_types=tuple("CrossAttnDownBlock2D" if a else "DownBlock2D" for a in attn),
              up_block_types=tuple("CrossAttnUpBlock2D" if a else "UpBlock2D" for a in reversed(attn)),
              layers_per_block=lpb, transformer_layers_per_block=tuple(depth), attention_head_dim=tuple(heads),
              cross_attention_dim=ctx, use_linear_projection=linear)
    if xl:
        ain = int(sd[p + "label_emb.0.0.weight"].shape[1])
        kw.update(addition_embed_type="text_time", projection_class_embeddings_input_dim=ain)
    assert te == 4 * widths[0], "time embedding width is 4 x the first block width in every SD UNet"
    return UNetConfig(**kw)


# ---- text encoder -----------------------------------------------------------------------------------------
def convert_ldm_clip(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """SD1.x: `cond_stage_model.transformer.*` already uses the HF CLIPTextModel names."""
    pre = "cond_stage_model.transformer."
    out = {k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}
    out.pop("text_model.embeddings.position_ids", None)   # buffer in old checkpoints, not a parameter
    return out


def convert_open_clip(sd: Dict[str, torch.Tensor], prefix: str = "cond_stage_model.model.") -> Dict[str, torch.Tensor]:
    """SD2.x: OpenCLIP ViT-H text tower (`cond_stage_model.model.*`) -> HF CLIPTextModel names; the fused
    `attn.in_proj_{weight,bias}` is split into q / k / v."""
    out: Dict[str, torch.Tensor] = {}
    simple = {"token_embedding.weight": "text_model.embeddings.token_embedding.weight",
              "positional_embedding": "text_model.embeddings.position_embedding.weight",
              "ln_final.weight": "text_model.final_layer_norm.weight", "ln_final.bias": "text_model.final_layer_norm.bias"}
    layer = {"ln_1": "layer_norm1", "ln_2": "layer_norm2", "mlp.c_fc": "mlp.fc1", "mlp.c_proj": "mlp.fc2",
             "attn.out_proj": "self_attn.out_proj"}
    for k, v in sd.items():
        if not k.startswith(prefix):
            continue
        r = k[len(prefix):]
        if r in simple:
            out[simple[r]] = v
            continue
        m = re.match(r"transformer\.resblocks\.(\d+)\.(.+)\.(weight|bias)$", r)
        if m:
            i, name, wb = m.group(1), m.group(2), m.group(3)
            base = f"text_model.encoder.layers.{i}."
            if name in layer:
                out[base + layer[name] + "." + wb] = v
            continue
        m = re.match(r"transformer\.resblocks\.(\d+)\.attn\.in_proj_(weight|bias)$", r)
        if m:
            i, wb = m.group(1), m.group(2)
            q, kk, vv = v.chunk(3, dim=0)
            base = f"text_model.encoder.layers.{i}.self_attn."
            out[base + "q_proj." + wb], out[base + "k_proj." + wb], out[base + "v_proj." + wb] = q, kk, vv
    return out


def read_checkpoint(path: str) -> Dict[str, torch.Tensor]:
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        return load_file(path)
    obj = torch.load(path, map_location="cpu", weights_only=True)
    return obj.get("state_dict", obj)
