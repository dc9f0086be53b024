#!/usr/bin/env python
"""Sampling smoke script -- the counterpart of the reference's `test/infer_xl.py` (:40-154) on the MI355X path.

Same flow: `model_util.load_models_xl` -> `train_util.encode_prompts_xl` for the prompt and the negative prompt ->
`concat_embeddings` of the text / pooled embeddings and the `add_time_ids` -> `get_initial_latents` (+ the SDXL noise
offset) -> `train_util.diffusion_xl` (DDIM, classifier-free guidance 7) through the HIP UNet, optionally with a trained
LoRA applied (`--lora out/x_last.safetensors`, the file `train_lora_xl.py` writes).

The reference then decodes the latents with diffusers' `AutoencoderKL` and saves a PNG; the VAE is not part of the
training hot path and is not implemented here, so this script stops at the latents and writes them to a safetensors
file (`--out`), which any diffusers VAE decodes as `vae.decode(latents / vae.config.scaling_factor)`.

    python examples/infer_xl.py --model synthetic:tiny_xl --height 128 --width 128 --steps 4
    python examples/infer_xl.py --model /models/sdxl-base --lora output/x_last.safetensors --prompt "a photo of lemonade"
"""
import argparse
import contextlib
import io
import os
import sys

import torch
from safetensors.torch import save_file

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from leco_amd import model_util, train_util  # noqa: E402
from leco_amd.lora import LoRANetwork  # noqa: E402

SDXL_NOISE_OFFSET = 0.0357      # test/infer_xl.py:27


@torch.no_grad()
def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="synthetic:tiny_xl")
    ap.add_argument("--prompt", default="a photo of lemonade")
    ap.add_argument("--negative_prompt", default="")
    ap.add_argument("--height", type=int, default=1024)
    ap.add_argument("--width", type=int, default=768)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--guidance_scale", type=float, default=7.0)
    ap.add_argument("--lora", default=None, help="LoRA weights saved by train_lora_xl.py")
    ap.add_argument("--rank", type=int, default=4)
    ap.add_argument("--alpha", type=float, default=1.0)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--device", default="cuda:0")
    ap.add_argument("--no_graphs", action="store_true", help="eager launches instead of one hipGraph per UNet pass")
    ap.add_argument("--out", default="latents.safetensors")
    args = ap.parse_args(argv)
    dev = torch.device(args.device)
    dtype = torch.bfloat16
    tokenizers, text_encoders, unet, sched = model_util.load_models_xl(args.model, scheduler_name="ddim")
    for te in text_encoders:
        te.to(dev, dtype=dtype)
        te.eval()
    unet.to(dev, dtype=dtype)
    unet.enable_xformers_memory_efficient_attention()
    unet.requires_grad_(False)
    unet.eval()
    unet.use_graphs = dev.type == "cuda" and not args.no_graphs
    network = None
    if args.lora:
        with contextlib.redirect_stdout(io.StringIO()):
            network = LoRANetwork(unet, rank=args.rank, multiplier=1.0, alpha=args.alpha).to(dev)
        network.load_weights(args.lora)
    add_time_ids = train_util.get_add_time_ids(args.height, args.width, dynamic_crops=False).to(dev)
    pos, pos_pooled = train_util.encode_prompts_xl(tokenizers, text_encoders, [args.prompt], num_images_per_prompt=1)
    neg, neg_pooled = train_util.encode_prompts_xl(tokenizers, text_encoders, [args.negative_prompt], num_images_per_prompt=1)
    text_embeds = train_util.concat_embeddings(neg, pos, 1)
    add_text_embeds = train_util.concat_embeddings(neg_pooled, pos_pooled, 1)
    add_time_ids = train_util.concat_embeddings(add_time_ids, add_time_ids, 1)
    sched.set_timesteps(args.steps, device=dev)
    torch.manual_seed(args.seed)
    latents = train_util.get_initial_latents(sched, 1, args.height, args.width, 1)
    latents = train_util.apply_noise_offset(latents * sched.init_noise_sigma, SDXL_NOISE_OFFSET).to(dev, dtype=dtype)
    with (network if network is not None else contextlib.nullcontext()):
        latents = train_util.diffusion_xl(unet, sched, latents, text_embeddings=text_embeds,
                                          add_text_embeddings=add_text_embeds, add_time_ids=add_time_ids,
                                          total_timesteps=args.steps, start_timesteps=0, guidance_scale=args.guidance_scale)
    save_file({"latents": latents.float().cpu().contiguous()}, args.out,
              {"prompt": args.prompt, "steps": str(args.steps), "guidance_scale": str(args.guidance_scale)})
    print(f"Done. latents {tuple(latents.shape)} -> {args.out}")
    return latents


if __name__ == "__main__":
    main()
