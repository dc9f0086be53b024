"""Generates tests/golden/*.safetensors by running the REFERENCE'S OWN code -- /root/reference
lora.py, train_util.py, prompt_util.py, imported read-only through oracle/stub_diffusers -- on the
oracle UNet / DDIM restatements (the reference's third-party compute, diffusers 0.20, is absent).
Run in the build container only:   python tests/golden/make_golden.py
The fixtures pin (a) the oracle's restatement of the reference loop / LoRA / loss and (b) the HIP path.
"""
import contextlib
import io
import os
import sys

import torch
from safetensors.torch import save_file

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle", "stub_diffusers"))
sys.path.insert(0, "/root/reference")

import lora as ref_lora  # noqa: E402  (reference)
import prompt_util as ref_pu  # noqa: E402
import train_util as ref_tu  # noqa: E402
from oracle import unet_ref as R  # noqa: E402
from oracle.ddim_ref import DDIMSchedulerRef  # noqa: E402

bf = torch.bfloat16
K, N_STEPS, BS = 3, 10, 1


def build(linear_proj=False):
    torch.manual_seed(0)
    unet = R.init_synthetic_(R.UNet2DConditionModel(R.tiny_config(linear_proj=linear_proj)), seed=1234)
    with torch.no_grad():
        for p in unet.parameters():
            p.copy_(p.to(bf).float())     # weights exactly representable in bf16
    unet.requires_grad_(False)
    torch.manual_seed(42)
    with contextlib.redirect_stdout(io.StringIO()):
        net = ref_lora.LoRANetwork(unet, rank=4, multiplier=1.0, alpha=1.0)
    g = torch.Generator().manual_seed(7)
    with torch.no_grad():
        for l in net.unet_loras:           # non-zero lora_up so the LoRA path is visible
            l.lora_down.weight.copy_((torch.randn(l.lora_down.weight.shape, generator=g) * 0.05).to(bf).float())
            l.lora_up.weight.copy_((torch.randn(l.lora_up.weight.shape, generator=g) * 0.05).to(bf).float())
    return unet, net


def main():
    out = {}
    unet, net = build()
    g = torch.Generator().manual_seed(11)
    emb = {n: (torch.randn(1, 77, 64, generator=g) * 3).to(bf).float()
           for n in ("target", "positive", "neutral", "unconditional")}
    lat = torch.randn(BS, 4, 16, 16, generator=g)
    for n, v in emb.items():
        out["emb." + n] = v
    out["latents"] = lat
    # LoRA names / shapes / init from the reference's own constructor
    torch.manual_seed(42)
    with contextlib.redirect_stdout(io.StringIO()):
        fresh = ref_lora.LoRANetwork(R.UNet2DConditionModel(R.tiny_config()), rank=4, multiplier=1.0, alpha=1.0)
    names = list(fresh.state_dict().keys())
    with open(os.path.join(HERE, "tiny_lora_keys.txt"), "w") as f:
        for k in names:
            f.write(f"{k} {tuple(fresh.state_dict()[k].shape)}\n")
    out["init.first_down"] = fresh.unet_loras[0].lora_down.weight.detach().clone()
    out["init.last_down"] = fresh.unet_loras[-1].lora_down.weight.detach().clone()
    for l in net.unet_loras:
        out["lora." + l.lora_name + ".down"] = l.lora_down.weight.detach().clone()
        out["lora." + l.lora_name + ".up"] = l.lora_up.weight.detach().clone()

    # one iteration of train_lora.py:141-281 with the reference's own primitives (fp32, CPU)
    sched = DDIMSchedulerRef()
    settings = ref_pu.PromptSettings(target="t", positive="p", neutral="n", unconditional="u", guidance_scale=2.0,
                                     batch_size=BS, resolution=128, action="erase")
    pair = ref_pu.PromptEmbedsPair(torch.nn.MSELoss(), emb["target"], emb["positive"], emb["unconditional"],
                                   emb["neutral"], settings)
    opt = torch.optim.AdamW(net.prepare_optimizer_params(), lr=1e-3)
    with torch.no_grad(), contextlib.redirect_stderr(io.StringIO()):
        sched.set_timesteps(N_STEPS)
        opt.zero_grad()
        with net:
            den = ref_tu.diffusion(unet, sched, lat.clone(), ref_tu.concat_embeddings(pair.unconditional, pair.target, BS),
                                   start_timesteps=0, total_timesteps=K, guidance_scale=3)
        sched.set_timesteps(1000)
        cur = sched.timesteps[int(K * 1000 / N_STEPS)]
        pos = ref_tu.predict_noise(unet, sched, cur, den, ref_tu.concat_embeddings(pair.unconditional, pair.positive, BS), guidance_scale=1)
        neu = ref_tu.predict_noise(unet, sched, cur, den, ref_tu.concat_embeddings(pair.unconditional, pair.neutral, BS), guidance_scale=1)
        unc = ref_tu.predict_noise(unet, sched, cur, den, ref_tu.concat_embeddings(pair.unconditional, pair.unconditional, BS), guidance_scale=1)
    with net:
        tgt = ref_tu.predict_noise(unet, sched, cur, den, ref_tu.concat_embeddings(pair.unconditional, pair.target, BS), guidance_scale=1)
    loss = pair.loss(target_latents=tgt, positive_latents=pos, neutral_latents=neu, unconditional_latents=unc)
    loss.backward()
    out["step.t_cur"] = torch.tensor([int(cur)])
    out["step.denoised"] = den
    out["step.pred.positive"], out["step.pred.neutral"], out["step.pred.unconditional"] = pos, neu, unc
    out["step.pred.target"] = tgt.detach()
    out["step.loss"] = loss.detach().reshape(1)
    out["step.grads"] = torch.cat([p.grad.reshape(-1) for l in net.unet_loras for p in (l.lora_down.weight, l.lora_up.weight)])
    opt.step()
    out["step.params_after"] = torch.cat([p.detach().reshape(-1) for l in net.unet_loras
                                          for p in (l.lora_down.weight, l.lora_up.weight)])
    # plain UNet forward (LoRA off) for the whole-UNet parity tests
    x = torch.randn(2, 4, 16, 16, generator=g).to(bf).float()
    ctx = torch.randn(2, 77, 64, generator=g).to(bf).float()
    out["unet.x"], out["unet.ctx"] = x, ctx
    with torch.no_grad():
        out["unet.y_t500"] = unet(x, torch.tensor(500), encoder_hidden_states=ctx).sample
    save_file({k: v.contiguous() for k, v in out.items()}, os.path.join(HERE, "tiny_step.safetensors"))
    print("loss", loss.item(), "t_cur", int(cur), "wrote", len(out), "tensors")


if __name__ == "__main__":
    main()
