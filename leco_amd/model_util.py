"""Model loading -- interface of the reference's ``model_util.py`` (``load_models`` :104-129,
``load_models_xl`` :205-227, ``create_noise_scheduler`` :230-278) without diffusers:

* a diffusers-format folder (``unet/config.json`` + ``unet/diffusion_pytorch_model.safetensors``,
  ``tokenizer/``, ``text_encoder/``) is read directly: the UNet module tree here uses the diffusers
  state-dict key names, so ``load_state_dict`` applies as is; CLIP comes from ``transformers``;
* ``synthetic:<sd15|sd21|sdxl|tiny>`` builds a seeded random-init UNet of that architecture and a
  deterministic stand-in text encoder (there are no checkpoints and no network on the build /
  benchmark boxes);
* single-file ``.ckpt`` / ``.safetensors`` checkpoints (LDM key layout) go through
  ``leco_amd/ckpt_convert.py`` (architecture detected from tensor shapes, key map derived from the block
  structure; CLIP-L / OpenCLIP-H text towers rebuilt as ``transformers`` models).
"""
from __future__ import annotations

import hashlib
import json
import math
import os
from typing import Optional

import torch

from .scheduler import create_noise_scheduler  # noqa: F401  (re-exported, model_util.py:230)
from .unet import UNet2DConditionModel, UNetConfig, sd15_config, sd21_config, sdxl_config

TOKENIZER_V1_MODEL_NAME = "CompVis/stable-diffusion-v1-4"
TOKENIZER_V2_MODEL_NAME = "stabilityai/stable-diffusion-2-1"


def tiny_config(linear_proj: bool = False) -> UNetConfig:
    return UNetConfig(block_out_channels=(64, 128, 128, 128), layers_per_block=1, attention_head_dim=2,
                      cross_attention_dim=64, use_linear_projection=linear_proj, sample_size=16)


SYNTHETIC = {"sd15": sd15_config, "sd21": sd21_config, "sdxl": sdxl_config, "tiny": tiny_config}


def init_synthetic_(unet: torch.nn.Module, seed: int = 1234) -> torch.nn.Module:
    """Seeded PyTorch-default-style init (U(+-1/sqrt(fan_in)) for conv/linear weights, norm
    gamma=1 / beta=0); keeps activations O(1) through every block."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in unet.named_parameters():
            if p.ndim >= 2:
                bound = 1.0 / math.sqrt(p[0].numel())
                p.copy_((torch.rand(p.shape, generator=g) * 2 - 1) * bound)
            elif "norm" in name:
                p.fill_(1.0 if name.endswith("weight") else 0.0)
            else:
                p.copy_((torch.rand(p.shape, generator=g) * 2 - 1) * 0.02)
    return unet


class _PromptIds(list):
    """Stand-in for the token-id tensor: the prompt strings themselves."""

    def to(self, *a, **k):
        return self


class SyntheticTokenizer:
    model_max_length = 77

    def __call__(self, prompts, **kw):
        class _Out:
            pass
        o = _Out()
        o.input_ids = _PromptIds(prompts)
        return o


class SyntheticTextEncoder(torch.nn.Module):
    """Deterministic stand-in for CLIP: embeds a prompt string as N(0,1) noise seeded by its hash
    (``randn(1,77,C)``).  ``[0]`` of the output is the (1,77,C) embedding."""

    def __init__(self, dim: int, pooled_dim: Optional[int] = None):
        super().__init__()
        self.dim, self.pooled_dim = dim, pooled_dim
        self._p = torch.nn.Parameter(torch.zeros(1))

    @property
    def device(self):
        return self._p.device

    def embed(self, prompt: str) -> torch.Tensor:
        seed = int.from_bytes(hashlib.sha256(prompt.encode()).digest()[:4], "little")
        g = torch.Generator().manual_seed(4321 + seed)
        return torch.randn(1, 77, self.dim, generator=g)

    def forward(self, tokens, **kw):
        emb = torch.cat([self.embed(p) for p in tokens]).to(self._p.device, self._p.dtype)
        return (emb,)


class _XLOut:
    def __init__(self, pooled, hidden):
        self.pooled, self.hidden_states = pooled, [hidden, hidden, hidden]

    def __getitem__(self, i):
        return (self.pooled, self.hidden_states[-1])[i]


class SyntheticTextEncoderXL(SyntheticTextEncoder):
    """Stand-in for the two SDXL text encoders: ``hidden_states[-2]`` is (1,77,dim), ``[0]`` the pooled vector."""

    def forward(self, tokens, output_hidden_states=False, **kw):
        hid = torch.cat([self.embed(p) for p in tokens]).to(self._p.device, self._p.dtype)
        g = torch.Generator().manual_seed(991)
        proj = torch.randn(self.dim, self.pooled_dim or self.dim, generator=g) / self.dim ** 0.5
        pooled = (hid.float().mean(1).cpu() @ proj).to(self._p.device, self._p.dtype)
        return _XLOut(pooled, hid)


def tiny_xl_config() -> UNetConfig:
    return UNetConfig(block_out_channels=(64, 128, 128),
                      down_block_types=("DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D"),
                      up_block_types=("CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "UpBlock2D"), layers_per_block=1,
                      transformer_layers_per_block=(1, 1, 2), attention_head_dim=(2, 2, 2), cross_attention_dim=64,
                      use_linear_projection=True, sample_size=16, addition_embed_type="text_time",
                      addition_time_embed_dim=32, projection_class_embeddings_input_dim=6 * 32 + 64)


def _load_unet_folder(path: str) -> UNet2DConditionModel:
    with open(os.path.join(path, "config.json")) as f:
        cfg = UNetConfig.from_dict(json.load(f))
    unet = UNet2DConditionModel(cfg)
    st = os.path.join(path, "diffusion_pytorch_model.safetensors")
    if os.path.exists(st):
        from safetensors.torch import load_file
        sd = load_file(st)
    else:
        sd = torch.load(os.path.join(path, "diffusion_pytorch_model.bin"), map_location="cpu")
    missing, unexpected = unet.load_state_dict(sd, strict=False)
    if missing:
        raise KeyError(f"UNet checkpoint {path} lacks keys, e.g. {missing[:5]}")
    return unet


def load_diffusers_model(path: str, v2: bool = False, clip_skip: Optional[int] = None,
                         weight_dtype: torch.dtype = torch.float32):
    from transformers import CLIPTextModel, CLIPTokenizer
    tokenizer = CLIPTokenizer.from_pretrained(path, subfolder="tokenizer")
    full = 24 if v2 else 12             # the released checkpoints (model_util.py:43-49, 58-61) ...
    try:                                # ... or whatever the folder's own config says (reduced test models)
        with open(os.path.join(path, "text_encoder", "config.json")) as f:
            full = int(json.load(f).get("num_hidden_layers", full))
    except OSError:
        pass
    default_layers = full - 1 if v2 else full   # v2: penultimate layer (model_util.py:43-49)
    nl = full - (clip_skip - 1) if clip_skip is not None else default_layers
    text_encoder = CLIPTextModel.from_pretrained(path, subfolder="text_encoder", num_hidden_layers=nl,
                                                 torch_dtype=weight_dtype)
    unet = _load_unet_folder(os.path.join(path, "unet")).to(weight_dtype)
    return tokenizer, text_encoder, unet


def load_unet_single_file(path: str) -> UNet2DConditionModel:
    """UNet of a single-file (LDM-layout) checkpoint; the architecture is detected from the tensor shapes."""
    from . import ckpt_convert as cc
    sd = cc.read_checkpoint(path)
    cfg = cc.detect_unet_config(sd)
    unet = UNet2DConditionModel(cfg)
    missing, unexpected = unet.load_state_dict(cc.convert_ldm_unet(sd, cfg), strict=False)
    if missing or unexpected:
        raise KeyError(f"{path}: UNet keys do not match the detected architecture "
                       f"(missing {missing[:3]}, unexpected {unexpected[:3]})")
    return unet


def _find_tokenizer_dir(ckpt_path: str) -> str:
    """CLIP vocabulary files cannot be fetched (no network): `<ckpt dir>/tokenizer/` or $LECO_TOKENIZER_DIR."""
    for d in (os.environ.get("LECO_TOKENIZER_DIR"), os.path.join(os.path.dirname(os.path.abspath(ckpt_path)), "tokenizer")):
        if d and os.path.isfile(os.path.join(d, "vocab.json")):
            return d
    raise FileNotFoundError("single-file checkpoints carry no tokenizer files: put the CLIP tokenizer (vocab.json, "
                            "merges.txt, ...) in a 'tokenizer' folder next to the checkpoint or set LECO_TOKENIZER_DIR")


def load_checkpoint_model(checkpoint_path: str, v2: bool = False, clip_skip: Optional[int] = None,
                          weight_dtype: torch.dtype = torch.float32):
    """model_util.py:75-101 (`StableDiffusionPipeline.from_single_file`): tokenizer, text encoder and UNet of a
    `.ckpt` / `.safetensors` file in the LDM key layout; the VAE is never touched."""
    from transformers import CLIPTextConfig, CLIPTextModel, CLIPTokenizer
    from . import ckpt_convert as cc
    tokenizer_dir = _find_tokenizer_dir(checkpoint_path)                # fail fast: nothing below can fetch it
    sd = cc.read_checkpoint(checkpoint_path)
    cfg = cc.detect_unet_config(sd)
    unet = UNet2DConditionModel(cfg)
    missing, unexpected = unet.load_state_dict(cc.convert_ldm_unet(sd, cfg), strict=False)
    if missing or unexpected:
        raise KeyError(f"{checkpoint_path}: UNet keys do not match the detected architecture "
                       f"(missing {missing[:3]}, unexpected {unexpected[:3]})")
    if any(k.startswith("cond_stage_model.model.") for k in sd):        # SD2.x: OpenCLIP ViT-H text tower
        te_sd = cc.convert_open_clip(sd)
        act, drop_last = "gelu", 1                                      # penultimate layer (model_util.py:43-49)
    elif any(k.startswith("cond_stage_model.transformer.") for k in sd):  # SD1.x: HF CLIP-L names already
        te_sd = cc.convert_ldm_clip(sd)
        act, drop_last = "quick_gelu", 0
    else:
        raise KeyError(f"{checkpoint_path}: no text encoder (cond_stage_model.*) in the checkpoint")
    # geometry from the tensors themselves (CLIP-L: 768 / 12 layers, OpenCLIP-H: 1024 / 24 layers, heads = width / 64)
    emb = te_sd["text_model.embeddings.token_embedding.weight"]
    full = 1 + max(int(k.split(".")[3]) for k in te_sd if k.startswith("text_model.encoder.layers."))
    hidden = int(emb.shape[1])
    tcfg = dict(vocab_size=int(emb.shape[0]), hidden_size=hidden, projection_dim=hidden, hidden_act=act,
                intermediate_size=int(te_sd["text_model.encoder.layers.0.mlp.fc1.weight"].shape[0]),
                num_attention_heads=max(1, hidden // 64),
                max_position_embeddings=int(te_sd["text_model.embeddings.position_embedding.weight"].shape[0]))
    nl = full - (clip_skip - 1) if clip_skip is not None else full - drop_last   # model_util.py:92-96
    # CLIP's BPE vocabulary ends with <|startoftext|>, <|endoftext|> (49406 / 49407); pooling looks for the latter
    text_encoder = CLIPTextModel(CLIPTextConfig(num_hidden_layers=nl, bos_token_id=tcfg["vocab_size"] - 2,
                                                eos_token_id=tcfg["vocab_size"] - 1, **tcfg))
    want = text_encoder.state_dict()
    if not any(k.startswith("text_model.") for k in want):             # transformers >= 5 dropped the prefix
        te_sd = {k[len("text_model."):] if k.startswith("text_model.") else k: v for k, v in te_sd.items()}
    te_sd = {k: v for k, v in te_sd.items() if k in want}               # layers beyond `nl` are dropped
    lacking = [k for k in want if k not in te_sd and not k.endswith("position_ids")]
    if lacking:
        raise KeyError(f"{checkpoint_path}: text encoder lacks {lacking[:3]}")
    text_encoder.load_state_dict(te_sd, strict=False)
    tokenizer = CLIPTokenizer.from_pretrained(tokenizer_dir)
    return tokenizer, text_encoder.to(weight_dtype), unet.to(weight_dtype)


def load_synthetic_model(kind: str, seed: int = 1234):
    cfg = SYNTHETIC[kind]()
    unet = init_synthetic_(UNet2DConditionModel(cfg), seed)
    return SyntheticTokenizer(), SyntheticTextEncoder(cfg.cross_attention_dim), unet


def load_models(pretrained_model_name_or_path: str, scheduler_name: str, v2: bool = False, v_pred: bool = False,
                weight_dtype: torch.dtype = torch.float32):
    p = pretrained_model_name_or_path
    if p.startswith("synthetic:"):
        tokenizer, text_encoder, unet = load_synthetic_model(p.split(":", 1)[1])
    elif p.endswith(".ckpt") or p.endswith(".safetensors"):
        tokenizer, text_encoder, unet = load_checkpoint_model(p, v2=v2, weight_dtype=weight_dtype)
    elif os.path.isdir(p):
        tokenizer, text_encoder, unet = load_diffusers_model(p, v2=v2, weight_dtype=weight_dtype)
    else:
        raise FileNotFoundError(f"{p}: not a local diffusers folder (no network access on this system); "
                                f"use a local path or synthetic:<sd15|sd21|sdxl|tiny>")
    scheduler = create_noise_scheduler(scheduler_name, prediction_type="v_prediction" if v_pred else "epsilon")
    return tokenizer, text_encoder, unet, scheduler


def load_models_xl(pretrained_model_name_or_path: str, scheduler_name: str, weight_dtype: torch.dtype = torch.float32):
    """model_util.py:205-227: returns ([tokenizer, tokenizer_2], [text_encoder, text_encoder_2], unet, scheduler)."""
    p = pretrained_model_name_or_path
    if p.startswith("synthetic:"):
        kind = p.split(":", 1)[1]
        cfg = tiny_xl_config() if kind in ("tiny_xl", "tinyxl") else SYNTHETIC[kind]()
        if cfg.addition_embed_type != "text_time":
            raise ValueError(f"{p} is not an SDXL-style architecture")
        unet = init_synthetic_(UNet2DConditionModel(cfg), 1234)
        pooled = cfg.projection_class_embeddings_input_dim - 6 * cfg.addition_time_embed_dim
        d1 = cfg.cross_attention_dim * 3 // 8
        d2 = cfg.cross_attention_dim - d1           # 768 + 1280 = 2048 for SDXL
        tokenizers = [SyntheticTokenizer(), SyntheticTokenizer()]
        text_encoders = [SyntheticTextEncoderXL(d1, pooled), SyntheticTextEncoderXL(d2, pooled)]
    elif os.path.isdir(p):
        from transformers import CLIPTextModel, CLIPTextModelWithProjection, CLIPTokenizer
        tokenizers = [CLIPTokenizer.from_pretrained(p, subfolder="tokenizer"),
                      CLIPTokenizer.from_pretrained(p, subfolder="tokenizer_2", pad_token_id=0)]
        text_encoders = [CLIPTextModel.from_pretrained(p, subfolder="text_encoder", torch_dtype=weight_dtype),
                         CLIPTextModelWithProjection.from_pretrained(p, subfolder="text_encoder_2",
                                                                     torch_dtype=weight_dtype)]
        unet = _load_unet_folder(os.path.join(p, "unet")).to(weight_dtype)
    else:
        raise FileNotFoundError(f"{p}: not a local diffusers folder; use a local path or synthetic:<sdxl|tiny_xl>")
    return tokenizers, text_encoders, unet, create_noise_scheduler(scheduler_name)
