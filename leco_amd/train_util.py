"""Step primitives with the reference's names and argument meaning (``train_util.py``):
latents (:20-57), ``concat_embeddings`` (:133-138), ``predict_noise`` (:142-168), ``diffusion``
(:172-193), XL variants (:217-330), optimizer / LR-scheduler factories (:333-401), resolution
bucket (:404-416).  They drive the MI355X UNet engine through the same duck-typed calls the
reference makes, so the reference loop body runs unchanged on top of them (the drop-in path);
``leco_amd.train.FusedStep`` is the graph-captured fast path for the same arithmetic."""
from typing import Optional

import torch

UNET_IN_CHANNELS = 4
VAE_SCALE_FACTOR = 8
UNET_ATTENTION_TIME_EMBED_DIM = 256
TEXT_ENCODER_2_PROJECTION_DIM = 1280
UNET_PROJECTION_CLASS_EMBEDDING_INPUT_DIM = 2816


def get_random_noise(batch_size: int, height: int, width: int, generator: torch.Generator = None) -> torch.Tensor:
    return torch.randn((batch_size, UNET_IN_CHANNELS, height // VAE_SCALE_FACTOR, width // VAE_SCALE_FACTOR),
                       generator=generator, device="cpu")


def apply_noise_offset(latents: torch.Tensor, noise_offset: float):
    return latents + noise_offset * torch.randn((latents.shape[0], latents.shape[1], 1, 1), device=latents.device)


def get_initial_latents(scheduler, n_imgs: int, height: int, width: int, n_prompts: int, generator=None) -> torch.Tensor:
    noise = get_random_noise(n_imgs, height, width, generator=generator).repeat(n_prompts, 1, 1, 1)
    return noise * scheduler.init_noise_sigma


def text_tokenize(tokenizer, prompts):
    return tokenizer(prompts, padding="max_length", max_length=tokenizer.model_max_length, truncation=True,
                     return_tensors="pt").input_ids


def text_encode(text_encoder, tokens):
    return text_encoder(tokens.to(text_encoder.device))[0]


def encode_prompts(tokenizer, text_encoder, prompts):
    return text_encode(text_encoder, text_tokenize(tokenizer, prompts))


def text_encode_xl(text_encoder, tokens, num_images_per_prompt: int = 1):
    prompt_embeds = text_encoder(tokens.to(text_encoder.device), output_hidden_states=True)
    pooled_prompt_embeds = prompt_embeds[0]
    prompt_embeds = prompt_embeds.hidden_states[-2]
    bs_embed, seq_len, _ = prompt_embeds.shape
    prompt_embeds = prompt_embeds.repeat(1, num_images_per_prompt, 1)
    return prompt_embeds.view(bs_embed * num_images_per_prompt, seq_len, -1), pooled_prompt_embeds


def encode_prompts_xl(tokenizers, text_encoders, prompts, num_images_per_prompt: int = 1):
    text_embeds_list, pooled = [], None
    for tokenizer, text_encoder in zip(tokenizers, text_encoders):
        ids = text_tokenize(tokenizer, prompts)
        text_embeds, pooled = text_encode_xl(text_encoder, ids, num_images_per_prompt)
        text_embeds_list.append(text_embeds)
    bs_embed = pooled.shape[0]
    pooled = pooled.repeat(1, num_images_per_prompt).view(bs_embed * num_images_per_prompt, -1)
    return torch.concat(text_embeds_list, dim=-1), pooled


def concat_embeddings(unconditional: torch.Tensor, conditional: torch.Tensor, n_imgs: int):
    return torch.cat([unconditional, conditional]).repeat_interleave(n_imgs, dim=0)


def _guided(unet, scheduler, timestep, latents, text_embeddings, guidance_scale, **unet_kwargs):
    """One classifier-free-guidance UNet evaluation: duplicate the latents, run the UNet on the
    [uncond | cond] embedding batch, combine the halves (train_util.py:151-166 / :231-251)."""
    doubled = scheduler.scale_model_input(torch.cat([latents, latents]), timestep)
    pred = unet(doubled, timestep, encoder_hidden_states=text_embeddings, **unet_kwargs).sample
    uncond, cond = pred.chunk(2)
    return uncond + guidance_scale * (cond - uncond)


def _denoise(step_fn, scheduler, latents, start, stop):
    for t in scheduler.timesteps[start:stop]:
        latents = scheduler.step(step_fn(t, latents), t, latents).prev_sample
    return latents


def predict_noise(unet, scheduler, timestep, latents, text_embeddings, guidance_scale=7.5):
    return _guided(unet, scheduler, timestep, latents, text_embeddings, guidance_scale)


@torch.no_grad()
def diffusion(unet, scheduler, latents, text_embeddings, total_timesteps: int = 1000, start_timesteps=0, **kwargs):
    return _denoise(lambda t, x: predict_noise(unet, scheduler, t, x, text_embeddings, **kwargs), scheduler,
                    latents, start_timesteps, total_timesteps)


def rescale_noise_cfg(noise_cfg, noise_pred_text, guidance_rescale=0.0):
    dims = list(range(1, noise_cfg.ndim))
    ratio = noise_pred_text.std(dim=dims, keepdim=True) / noise_cfg.std(dim=dims, keepdim=True)
    return guidance_rescale * (noise_cfg * ratio) + (1 - guidance_rescale) * noise_cfg


def predict_noise_xl(unet, scheduler, timestep, latents, text_embeddings, add_text_embeddings, add_time_ids,
                     guidance_scale=7.5, guidance_rescale=0.7):
    # NB the reference evaluates rescale_noise_cfg(...) here and then returns the un-rescaled
    # guided prediction (train_util.py:253-257); only the returned value matters.
    return _guided(unet, scheduler, timestep, latents, text_embeddings, guidance_scale,
                   added_cond_kwargs={"text_embeds": add_text_embeddings, "time_ids": add_time_ids})


@torch.no_grad()
def diffusion_xl(unet, scheduler, latents, text_embeddings, add_text_embeddings, add_time_ids,
                 guidance_scale: float = 1.0, total_timesteps: int = 1000, start_timesteps=0):
    return _denoise(lambda t, x: predict_noise_xl(unet, scheduler, t, x, text_embeddings, add_text_embeddings,
                                                  add_time_ids, guidance_scale=guidance_scale), scheduler, latents,
                    start_timesteps, total_timesteps)


def get_add_time_ids(height: int, width: int, dynamic_crops: bool = False, dtype: torch.dtype = torch.float32):
    if dynamic_crops:
        random_scale = torch.rand(1).item() * 2 + 1
        original_size = (int(height * random_scale), int(width * random_scale))
        crops_coords_top_left = (torch.randint(0, original_size[0] - height, (1,)).item(),
                                 torch.randint(0, original_size[1] - width, (1,)).item())
    else:
        original_size = (height, width)
        crops_coords_top_left = (0, 0)
    target_size = (height, width)
    add_time_ids = list(original_size + crops_coords_top_left + target_size)
    passed = UNET_ATTENTION_TIME_EMBED_DIM * len(add_time_ids) + TEXT_ENCODER_2_PROJECTION_DIM
    if passed != UNET_PROJECTION_CLASS_EMBEDDING_INPUT_DIM:
        raise ValueError(f"Model expects an added time embedding vector of length "
                         f"{UNET_PROJECTION_CLASS_EMBEDDING_INPUT_DIM}, but a vector of {passed} was created.")
    return torch.tensor([add_time_ids], dtype=dtype)


_OPTIONAL_OPTIMIZERS = {
    # name -> (package, attribute); imported lazily, exactly the set train_util.py:333-370 accepts
    "dadaptadam": ("dadaptation", "DAdaptAdam"), "dadaptlion": ("dadaptation", "DAdaptLion"),
    "adam8bit": ("bitsandbytes.optim", "Adam8bit"), "lion8bit": ("bitsandbytes.optim", "Lion8bit"),
    "lion": ("lion_pytorch", "Lion"), "prodigy": ("prodigyopt", "Prodigy"),
}


def get_optimizer(name: str):
    key = name.lower()
    if key in ("adam", "adamw"):
        return torch.optim.Adam if key == "adam" else torch.optim.AdamW
    if key in _OPTIONAL_OPTIMIZERS:
        import importlib
        pkg, attr = _OPTIONAL_OPTIMIZERS[key]
        return getattr(importlib.import_module(pkg), attr)
    if key.startswith("dadapt"):
        raise ValueError("DAdapt optimizer must be dadaptadam or dadaptlion")
    if key.endswith("8bit"):
        raise ValueError("8bit optimizer must be adam8bit or lion8bit")
    raise ValueError("Optimizer must be adam, adamw, lion or Prodigy")


def get_lr_scheduler(name: Optional[str], optimizer, max_iterations: Optional[int], lr_min: Optional[float], **kwargs):
    S = torch.optim.lr_scheduler
    table = {
        "cosine": lambda: S.CosineAnnealingLR(optimizer, T_max=max_iterations, eta_min=lr_min, **kwargs),
        "cosine_with_restarts": lambda: S.CosineAnnealingWarmRestarts(optimizer, T_0=max_iterations // 10, T_mult=2,
                                                                      eta_min=lr_min, **kwargs),
        "step": lambda: S.StepLR(optimizer, step_size=max_iterations // 100, gamma=0.999, **kwargs),
        "constant": lambda: S.ConstantLR(optimizer, factor=1, **kwargs),
        "linear": lambda: S.LinearLR(optimizer, factor=0.5, total_iters=max_iterations // 100, **kwargs),
    }
    if name not in table:
        raise ValueError("Scheduler must be cosine, cosine_with_restarts, step, linear or constant")
    return table[name]()


def get_random_resolution_in_bucket(bucket_resolution: int = 512, generator=None):
    """train_util.py:404-416.  ``generator`` (extension): the data-parallel loop draws the bucket from a generator shared by
    all ranks, so that a step has the same shape everywhere."""
    max_resolution, min_resolution, step = bucket_resolution, bucket_resolution // 2, 64
    min_step, max_step = min_resolution // step, max_resolution // step
    height = torch.randint(min_step, max_step, (1,), generator=generator).item() * step
    width = torch.randint(min_step, max_step, (1,), generator=generator).item() * step
    return height, width
