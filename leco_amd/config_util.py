"""Training-config YAML schema of the reference (`config_util.py:13-72`: sections pretrained_model / network /
train / save / logging / other under a root with `prompts_file`), `parse_precision` (`:75-83`) and
`load_config_from_yaml` (`:86-104`).

The sections are declared once in `_SECTIONS` (field -> (type, default)) and turned into pydantic-v2 models with
`create_model`; the same table drives the loader's "missing optional section -> its defaults" rule (`:92-102`; under
pydantic v2 a bare `Optional[X]` would be *required*, so the optional sections default to None here) and the
`describe()` helper used in error messages.  Unknown keys are ignored and YAML strings such as `lr: 1e-4` are coerced
to float, as with the reference's models."""
from typing import Dict, Literal, Optional, Tuple

import torch
import yaml
from pydantic import BaseModel, create_model

from .lora import TRAINING_METHODS

PRECISION_TYPES = Literal["fp32", "fp16", "bf16", "float32", "float16", "bfloat16"]
NETWORK_TYPES = Literal["lierla", "c3lier"]
SCHEDULER_NAMES = Literal["ddim", "ddpm", "lms", "euler_a"]
_REQUIRED = ...

# section name -> (model class name, optional at the root?, {field: (annotation, default)})
_SECTIONS: Dict[str, Tuple[str, bool, Dict[str, tuple]]] = {
    "pretrained_model": ("PretrainedModelConfig", False, {
        "name_or_path": (str, _REQUIRED), "v2": (bool, False), "v_pred": (bool, False),
        "clip_skip": (Optional[int], None)}),
    "network": ("NetworkConfig", False, {
        "type": (NETWORK_TYPES, "lierla"), "rank": (int, 4), "alpha": (float, 1.0),
        "training_method": (TRAINING_METHODS, "full")}),
    "train": ("TrainConfig", True, {
        "precision": (PRECISION_TYPES, "bfloat16"), "noise_scheduler": (SCHEDULER_NAMES, "ddim"),
        "iterations": (int, 500), "lr": (float, 1e-4), "optimizer": (str, "adamw"), "optimizer_args": (str, ""),
        "lr_scheduler": (str, "constant"), "max_denoising_steps": (int, 50)}),
    "save": ("SaveConfig", True, {
        "name": (str, "untitled"), "path": (str, "./output"), "per_steps": (int, 200),
        "precision": (PRECISION_TYPES, "float32")}),
    "logging": ("LoggingConfig", True, {"use_wandb": (bool, False), "verbose": (bool, False)}),
    "other": ("OtherConfig", True, {"use_xformers": (bool, False)}),
}

_MODELS = {sec: create_model(cls, __base__=BaseModel, __module__=__name__, **fields)
           for sec, (cls, _, fields) in _SECTIONS.items()}
PretrainedModelConfig = _MODELS["pretrained_model"]
NetworkConfig = _MODELS["network"]
TrainConfig = _MODELS["train"]
SaveConfig = _MODELS["save"]
LoggingConfig = _MODELS["logging"]
OtherConfig = _MODELS["other"]

RootConfig = create_model(
    "RootConfig", __base__=BaseModel, __module__=__name__, prompts_file=(str, _REQUIRED),
    **{sec: ((Optional[_MODELS[sec]], None) if optional else (_MODELS[sec], _REQUIRED))
       for sec, (_, optional, _f) in _SECTIONS.items()})

_DTYPES = {torch.float32: ("fp32", "float32"), torch.float16: ("fp16", "float16"), torch.bfloat16: ("bf16", "bfloat16")}


def parse_precision(precision: str) -> torch.dtype:
    for dtype, names in _DTYPES.items():
        if precision in names:
            return dtype
    raise ValueError(f"Invalid precision type: {precision}")


def describe() -> str:
    """One line per section with its fields and defaults (for `--help`-style messages)."""
    return "\n".join(f"{sec}: " + ", ".join(f"{k}={'<required>' if d is _REQUIRED else d!r}" for k, (_, d) in fields.items())
                     for sec, (_, _, fields) in _SECTIONS.items())


def load_config_from_yaml(config_path: str):
    with open(config_path, "r") as fh:
        raw = yaml.load(fh, Loader=yaml.FullLoader)
    root = RootConfig(**raw)
    for sec, (_, optional, _f) in _SECTIONS.items():
        if optional and getattr(root, sec) is None:
            setattr(root, sec, _MODELS[sec]())
    return root
