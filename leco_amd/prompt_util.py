"""Prompt settings schema, embedding cache and the ESD objective -- interface of the
reference's ``prompt_util.py`` (PromptSettings :43-67, PromptEmbedsCache :30-40,
PromptEmbedsXL :17-23, PromptEmbedsPair :70-148, load_prompts_from_yaml :151-160).

``PromptEmbedsPair.loss`` keeps the reference signature (four predicted-noise tensors) and its
exact arithmetic -- ``loss_fn(target, neutral -/+ guidance_scale * (positive - unconditional))``
-- evaluated with torch ops on whatever device the tensors live on (the drop-in path).  The
fused training step (``leco_amd.train``) evaluates the same expression plus its gradient in one
HIP kernel (``leco_esd_loss``) without the reference's four device->host copies."""
from pathlib import Path
from typing import Literal, Optional, Union

import torch
import yaml
from pydantic import BaseModel, model_validator

ACTION_TYPES = Literal["erase", "enhance"]


class PromptEmbedsXL:
    text_embeds: torch.FloatTensor
    pooled_embeds: torch.FloatTensor

    def __init__(self, *args) -> None:
        # the reference's loop passes ONE tuple here and crashes (train_lora_xl.py:131-138,
        # SURVEY.md F-7); accept both the intended unpacked form and the tuple form.
        if len(args) == 1 and isinstance(args[0], (tuple, list)):
            args = tuple(args[0])
        self.text_embeds = args[0]
        self.pooled_embeds = args[1]


PROMPT_EMBEDDING = Union[torch.FloatTensor, PromptEmbedsXL]


class PromptEmbedsCache:
    def __init__(self):
        # per-instance (the reference's dict is class-level and shared by accident)
        self.prompts: dict = {}

    def __setitem__(self, name: str, value: PROMPT_EMBEDDING) -> None:
        self.prompts[name] = value

    def __getitem__(self, name: str) -> Optional[PROMPT_EMBEDDING]:
        return self.prompts.get(name)


class PromptSettings(BaseModel):
    target: str
    positive: Optional[str] = None       # if None, target will be used
    unconditional: str = ""
    neutral: Optional[str] = None        # if None, unconditional will be used
    action: ACTION_TYPES = "erase"
    guidance_scale: float = 1.0
    resolution: int = 512
    dynamic_resolution: bool = False
    batch_size: int = 1
    dynamic_crops: bool = False          # only used when model is XL

    @model_validator(mode="before")
    @classmethod
    def fill_prompts(cls, values):
        values = dict(values)
        if "target" not in values:
            raise ValueError("target must be specified")
        if "positive" not in values:
            values["positive"] = values["target"]
        if "unconditional" not in values:
            values["unconditional"] = ""
        if "neutral" not in values:
            values["neutral"] = values["unconditional"]
        return values


class PromptEmbedsPair:
    def __init__(self, loss_fn, target, positive, unconditional, neutral, settings: PromptSettings) -> None:
        self.loss_fn = loss_fn
        self.target = target
        self.positive = positive
        self.unconditional = unconditional
        self.neutral = neutral
        self.guidance_scale = settings.guidance_scale
        self.resolution = settings.resolution
        self.dynamic_resolution = settings.dynamic_resolution
        self.batch_size = settings.batch_size
        self.dynamic_crops = settings.dynamic_crops
        self.action = settings.action

    def _erase(self, target_latents, positive_latents, unconditional_latents, neutral_latents):
        """action == "erase": push the target prediction AWAY from the positive concept (neutral - g (positive - uncond))."""
        return self.loss_fn(target_latents,
                            neutral_latents - self.guidance_scale * (positive_latents - unconditional_latents))

    def _enhance(self, target_latents, positive_latents, unconditional_latents, neutral_latents):
        """action == "enhance": pull the target prediction TOWARDS the positive concept (neutral + g (positive - uncond))."""
        return self.loss_fn(target_latents,
                            neutral_latents + self.guidance_scale * (positive_latents - unconditional_latents))

    @property
    def sign(self) -> float:
        if self.action == "erase":
            return -1.0
        if self.action == "enhance":
            return 1.0
        raise ValueError("action must be erase or enhance")

    def loss(self, **kwargs):
        if self.action == "erase":
            return self._erase(**kwargs)
        elif self.action == "enhance":
            return self._enhance(**kwargs)
        raise ValueError("action must be erase or enhance")


def load_prompts_from_yaml(path) -> list:
    with open(path, "r") as f:
        prompts = yaml.safe_load(f)
    if prompts is None or len(prompts) == 0:
        raise ValueError("prompts file is empty")
    return [PromptSettings(**prompt) for prompt in prompts]
