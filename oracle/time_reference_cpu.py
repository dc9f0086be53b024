"""ORACLE / test infrastructure -- CPU baseline of kind "reference" (build container only: needs /root/reference).

Times FULL optimizer steps of the reference's loop body (train_lora.py:141-290) driven by the REFERENCE'S OWN
`train_util.py` (diffusion / predict_noise / concat_embeddings / get_initial_latents), `lora.py` (LoRANetwork, the
forward-patching LoRAModule) and `prompt_util.py` (PromptEmbedsPair.loss), imported read-only through
oracle/stub_diffusers, on `torch.device("cpu")` -- BASELINE.json configs[0]: SD1.5 architecture, rank 4 / alpha 1, 512^2,
prompt batch 1, fp32, DDIM-50, AdamW.  The two things the reference takes from diffusers (UNet2DConditionModel.forward,
DDIMScheduler) are the oracle restatements, as everywhere in this repo.  The patches are exactly BASELINE.md section 3's:
device -> cpu, no xformers call, no wandb; `flush()` (train_lora.py:283-290) kept.

    python oracle/time_reference_cpu.py [k ...]        (default k = 1 2)   ->  one JSON line

bench.py's `cpu_baseline` times the same loop through oracle/step_ref.py (kind "port") because /root/reference does not
exist on the GPU box; this script is the cross-check that the port costs what the reference's own code costs."""
import contextlib
import gc
import io
import json
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(HERE, "stub_diffusers"))
sys.path.insert(0, "/root/reference")

import lora as ref_lora  # noqa: E402  (reference)
import prompt_util as ref_pu  # noqa: E402
import train_util as ref_tu  # noqa: E402
from oracle import unet_ref as R  # noqa: E402
from oracle.ddim_ref import DDIMSchedulerRef  # noqa: E402


def main():
    ks = [int(a) for a in sys.argv[1:]] or [1, 2]
    torch.manual_seed(1234)
    unet = R.init_synthetic_(R.UNet2DConditionModel(R.sd15_config()), seed=1234)
    unet.requires_grad_(False)
    unet.eval()
    with contextlib.redirect_stdout(io.StringIO()):
        network = ref_lora.LoRANetwork(unet, rank=4, multiplier=1.0, alpha=1.0)
    g = torch.Generator().manual_seed(99)
    with torch.no_grad():
        for l in network.unet_loras:
            l.lora_up.weight.copy_(torch.randn(l.lora_up.weight.shape, generator=g) * 0.02)
    optimizer = torch.optim.AdamW(network.prepare_optimizer_params(), lr=1e-4)
    lr_scheduler = ref_tu.get_lr_scheduler("constant", optimizer, max_iterations=1000, lr_min=1e-6)
    noise_scheduler = DDIMSchedulerRef()
    eg = torch.Generator().manual_seed(4321)
    emb = {"van gogh": torch.randn(1, 77, 768, generator=eg), "": torch.randn(1, 77, 768, generator=eg)}
    settings = ref_pu.PromptSettings(target="van gogh", positive="van gogh", unconditional="", neutral="", action="erase",
                                     guidance_scale=1.0, resolution=512, batch_size=1)
    pair = ref_pu.PromptEmbedsPair(torch.nn.MSELoss(), emb["van gogh"], emb["van gogh"], emb[""], emb[""], settings)
    with torch.no_grad():      # untimed warm-up pass (thread pool, allocator): the first UNet forward of a process is ~1.6x slow
        unet(torch.zeros(2, 4, 64, 64), torch.tensor(1), encoder_hidden_states=torch.zeros(2, 77, 768))
    times = []
    for i, k in enumerate(ks):
        torch.manual_seed(1000 + i)
        t0 = time.perf_counter()
        # ---- train_lora.py:141-290, device = cpu
        with torch.no_grad():
            noise_scheduler.set_timesteps(50, device="cpu")
            optimizer.zero_grad()
            latents = ref_tu.get_initial_latents(noise_scheduler, pair.batch_size, 512, 512, 1)
            with network:
                denoised = ref_tu.diffusion(unet, noise_scheduler, latents,
                                            ref_tu.concat_embeddings(pair.unconditional, pair.target, pair.batch_size),
                                            start_timesteps=0, total_timesteps=k, guidance_scale=3)
            noise_scheduler.set_timesteps(1000)
            cur = noise_scheduler.timesteps[int(k * 1000 / 50)]
            preds = [ref_tu.predict_noise(unet, noise_scheduler, cur, denoised,
                                          ref_tu.concat_embeddings(pair.unconditional, e, pair.batch_size), guidance_scale=1)
                     for e in (pair.positive, pair.neutral, pair.unconditional)]
        with network:
            target = ref_tu.predict_noise(unet, noise_scheduler, cur, denoised,
                                          ref_tu.concat_embeddings(pair.unconditional, pair.target, pair.batch_size),
                                          guidance_scale=1)
        for p in preds:
            p.requires_grad = False
        loss = pair.loss(target_latents=target, positive_latents=preds[0], neutral_latents=preds[1],
                         unconditional_latents=preds[2])
        loss.backward()
        optimizer.step()
        lr_scheduler.step()
        del preds, target, latents, loss
        gc.collect()          # flush(): empty_cache is a no-op without CUDA
        times.append(time.perf_counter() - t0)
    out = {"kind": "reference", "k": ks, "step_seconds": times, "threads": torch.get_num_threads(), "host_cpus": os.cpu_count()}
    if len(ks) >= 2 and ks[1] != ks[0]:
        b = (times[1] - times[0]) / (ks[1] - ks[0])
        out["a_fixed_s"], out["b_per_pass_s"] = times[0] - b * ks[0], b
    print(json.dumps(out))


if __name__ == "__main__":
    main()
