#!/usr/bin/env python
"""One command that turns the oracle from "parity unpinned" to pinned -- for whoever has `diffusers` installed.

The reference keeps the UNet / DDIM arithmetic in the un-vendored dependency diffusers==0.20.0 (requirements.txt:1),
which is absent from /root/reference and from the build image (no network), so `oracle/unet_ref.py` and
`oracle/ddim_ref.py` restate the published algorithm and every parity claim in this repository is relative to THEM.
With the real package this script compares them directly, on the same synthetic weights:

    python tools/check_against_diffusers.py [--arch tiny sd15 sd21 sdxl] [--device cuda:0]

  * builds diffusers' `UNet2DConditionModel` from the public unet/config.json values,
    copies its random-init state_dict into the oracle (same parameter names: `load_state_dict(strict=True)` is the key
    -name check), runs both in fp32 on the same latents / timestep / prompt embeddings (+ SDXL added conditions) and
    reports the relative L2 difference of `.sample` (expected: fp32 rounding, <= 1e-5);
  * steps diffusers' `DDIMScheduler` (ctor arguments of model_util.py:239-246) and the oracle's side by side through the
    reference's loop shape (set_timesteps(50), k steps, epsilon and v prediction): timesteps and samples must agree.

Exit code 0: all comparisons within tolerance; 1: a mismatch; 2: diffusers is not importable here (nothing was checked).
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path = [p for p in sys.path if not p.rstrip("/").endswith("stub_diffusers")]      # never the import stub

try:
    import diffusers
    from diffusers import DDIMScheduler, UNet2DConditionModel
    if not hasattr(diffusers, "__version__"):
        raise ImportError("stub")
except Exception as e:      # noqa: BLE001
    print(f"diffusers is not importable here ({e!r}): nothing checked.  Install diffusers==0.20.0 (the reference's pin) "
          f"and re-run.")
    sys.exit(2)

from oracle import unet_ref as R  # noqa: E402
from oracle.ddim_ref import DDIMSchedulerRef  # noqa: E402


def rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


def diffusers_kwargs(c: "R.UNetConfig") -> dict:
    kw = dict(sample_size=c.sample_size, in_channels=c.in_channels, out_channels=c.out_channels,
              down_block_types=tuple(c.down_block_types), up_block_types=tuple(c.up_block_types),
              block_out_channels=tuple(c.block_out_channels), layers_per_block=c.layers_per_block,
              cross_attention_dim=c.cross_attention_dim, attention_head_dim=c.attention_head_dim,
              use_linear_projection=c.use_linear_projection, norm_num_groups=c.norm_num_groups, norm_eps=c.norm_eps,
              transformer_layers_per_block=c.transformer_layers_per_block)
    if c.addition_embed_type == "text_time":
        kw.update(addition_embed_type="text_time", addition_time_embed_dim=c.addition_time_embed_dim,
                  projection_class_embeddings_input_dim=c.projection_class_embeddings_input_dim)
    return kw


def check_unet(arch: str, dev) -> bool:
    cfg = {"tiny": R.tiny_config, "tiny_xl": lambda: R.tiny_config(xl=True), "sd15": R.sd15_config, "sd21": R.sd21_config,
           "sdxl": R.sdxl_config}[arch]()
    torch.manual_seed(1234)
    ref = UNet2DConditionModel(**diffusers_kwargs(cfg)).to(dev).eval()
    ora = R.UNet2DConditionModel(cfg).to(dev).eval()
    ora.load_state_dict(ref.state_dict(), strict=True)          # identical parameter names and shapes, or this raises
    n = sum(p.numel() for p in ref.parameters())
    g = torch.Generator().manual_seed(0)
    s = min(cfg.sample_size, 32)
    x = torch.randn(2, cfg.in_channels, s, s, generator=g).to(dev)
    ctx = torch.randn(2, 77, cfg.cross_attention_dim, generator=g).to(dev)
    kw = {}
    if cfg.addition_embed_type == "text_time":
        pooled = cfg.projection_class_embeddings_input_dim - 6 * cfg.addition_time_embed_dim
        kw["added_cond_kwargs"] = {"text_embeds": torch.randn(2, pooled, generator=g).to(dev),
                                   "time_ids": torch.tensor([[s * 8., s * 8., 0., 0., s * 8., s * 8.]] * 2, device=dev)}
    worst = 0.0
    with torch.no_grad():
        for t in (999, 500, 19):
            a = ref(x, torch.tensor(t, device=dev), encoder_hidden_states=ctx, **kw).sample
            b = ora(x, torch.tensor(t, device=dev), encoder_hidden_states=ctx, **kw).sample
            worst = max(worst, rel(b, a))
    ok = worst <= 1e-5
    print(f"[{'ok' if ok else 'MISMATCH'}] UNet {arch}: {n:,} parameters, state_dict keys identical, "
          f"max relative difference of .sample over t in (999, 500, 19) = {worst:.2e}")
    return ok


def check_ddim() -> bool:
    ok = True
    for pt in ("epsilon", "v_prediction"):
        a = DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", num_train_timesteps=1000,
                          clip_sample=False, prediction_type=pt)
        b = DDIMSchedulerRef(prediction_type=pt)
        a.set_timesteps(50)
        b.set_timesteps(50)
        same_t = [int(t) for t in a.timesteps] == [int(t) for t in b.timesteps]
        g = torch.Generator().manual_seed(1)
        xa = xb = torch.randn(2, 4, 16, 16, generator=g) * float(a.init_noise_sigma)
        worst = 0.0
        for t in a.timesteps[:12]:
            out = torch.randn(2, 4, 16, 16, generator=g)
            xa = a.step(out, t, xa).prev_sample
            xb = b.step(out, t, xb).prev_sample
            worst = max(worst, rel(xb, xa))
        a.set_timesteps(1000)
        b.set_timesteps(1000)
        same_t &= int(a.timesteps[20 * 7]) == int(b.timesteps[20 * 7]) == 999 - 20 * 7       # train_lora.py:195-199
        good = same_t and worst <= 1e-6
        ok &= good
        print(f"[{'ok' if good else 'MISMATCH'}] DDIM {pt}: timesteps identical = {same_t}, max relative difference over 12 "
              f"steps = {worst:.2e}")
    return ok


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--arch", nargs="+", default=["tiny", "tiny_xl", "sd15"])
    ap.add_argument("--device", default="cuda:0" if torch.cuda.is_available() else "cpu")
    a = ap.parse_args()
    print(f"diffusers {diffusers.__version__} (the reference pins 0.20.0), device {a.device}")
    ok = check_ddim()
    for arch in a.arch:
        ok &= check_unet(arch, torch.device(a.device))
    print("oracle pinned against diffusers" if ok else "MISMATCH: the oracle does not reproduce diffusers")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
