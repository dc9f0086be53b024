"""Config YAML schema -- same keys, defaults and loader behaviour as the reference's
``config_util.py`` (RootConfig :60-72, parse_precision :75-83, load_config_from_yaml :86-104),
written for pydantic v2 (under v2 the reference's bare ``Optional[X]`` sections become
required; here they default to None and are filled in by the loader, which is what the
reference intends at :92-102).  Unknown keys are ignored and ``lr: 1e-4`` (a YAML string)
is coerced to float, as in the reference."""
from typing import Literal, Optional

import torch
import yaml
from pydantic import BaseModel, model_validator

from .lora import TRAINING_METHODS

PRECISION_TYPES = Literal["fp32", "fp16", "bf16", "float32", "float16", "bfloat16"]
NETWORK_TYPES = Literal["lierla", "c3lier"]


class PretrainedModelConfig(BaseModel):
    name_or_path: str
    v2: bool = False
    v_pred: bool = False
    clip_skip: Optional[int] = None


MAX_LORA_RANK = None       # lierla: any rank, like the reference (lora.py:49-95): stacked q|k|v columns above 64 run as chained 64-wide K-extension steps
MAX_LORA_RANK_C3LIER = None  # conv LoRA: the low-rank image is a whole number of 64-channel chunks; ranks above 64 chain further 64-column slices (round 4)


class NetworkConfig(BaseModel):
    type: NETWORK_TYPES = "lierla"
    rank: int = 4
    alpha: float = 1.0
    training_method: TRAINING_METHODS = "full"

    @model_validator(mode="after")
    def _rank_fits_the_lora_kernels(self):
        # the reference accepts any rank (lora.py clamps a conv module's rank to min(rank, in, out)); so do both paths here
        cap = MAX_LORA_RANK_C3LIER if self.type == "c3lier" else MAX_LORA_RANK
        if self.rank < 1 or (cap is not None and self.rank > cap):
            raise ValueError(f"network.rank={self.rank}: network.type {self.type} supports ranks 1..{cap or 'any'}")
        return self


class TrainConfig(BaseModel):
    precision: PRECISION_TYPES = "bfloat16"
    noise_scheduler: Literal["ddim", "ddpm", "lms", "euler_a"] = "ddim"
    iterations: int = 500
    lr: float = 1e-4
    optimizer: str = "adamw"
    optimizer_args: str = ""
    lr_scheduler: str = "constant"
    max_denoising_steps: int = 50


class SaveConfig(BaseModel):
    name: str = "untitled"
    path: str = "./output"
    per_steps: int = 200
    precision: PRECISION_TYPES = "float32"


class LoggingConfig(BaseModel):
    use_wandb: bool = False
    verbose: bool = False


class OtherConfig(BaseModel):
    use_xformers: bool = False


class RootConfig(BaseModel):
    prompts_file: str
    pretrained_model: PretrainedModelConfig
    network: NetworkConfig
    train: Optional[TrainConfig] = None
    save: Optional[SaveConfig] = None
    logging: Optional[LoggingConfig] = None
    other: Optional[OtherConfig] = None


def parse_precision(precision: str) -> torch.dtype:
    if precision in ("fp32", "float32"):
        return torch.float32
    if precision in ("fp16", "float16"):
        return torch.float16
    if precision in ("bf16", "bfloat16"):
        return torch.bfloat16
    raise ValueError(f"Invalid precision type: {precision}")


def load_config_from_yaml(config_path: str) -> RootConfig:
    with open(config_path, "r") as f:
        config = yaml.load(f, Loader=yaml.FullLoader)
    root = RootConfig(**config)
    if root.train is None:
        root.train = TrainConfig()
    if root.save is None:
        root.save = SaveConfig()
    if root.logging is None:
        root.logging = LoggingConfig()
    if root.other is None:
        root.other = OtherConfig()
    return root
