"""ORACLE (test infrastructure only) -- restatement of ONE iteration of the reference training
loop body (train_lora.py:141-281) and of the step primitives it calls (train_util.py:133-193,
prompt_util.py:107-148) on top of the oracle UNet / DDIM / LoRA, in plain PyTorch on any device
and dtype.  The golden fixtures in tests/golden/ were produced by the reference's OWN
train_util.py / prompt_util.py / lora.py driving the same oracle UNet (tests/golden/make_golden.py),
and tests/test_oracle_golden.py checks this file reproduces them bit for bit in fp32.
Only tests/, smoke() and bench.py's cpu_baseline may import this."""
import torch
import torch.nn.functional as F


def concat_embeddings(unconditional, conditional, n_imgs):          # train_util.py:133-138
    return torch.cat([unconditional, conditional]).repeat_interleave(n_imgs, dim=0)


def predict_noise(unet, scheduler, timestep, latents, text_embeddings, guidance_scale, added=None):
    """train_util.py:142-168; with ``added`` = {"text_embeds","time_ids"} it is predict_noise_xl (:217-257,
    whose rescale_noise_cfg result is discarded by the reference)."""
    x = scheduler.scale_model_input(torch.cat([latents] * 2), timestep)
    kw = {} if added is None else {"added_cond_kwargs": added}
    pred = unet(x, timestep, encoder_hidden_states=text_embeddings, **kw).sample
    u, c = pred.chunk(2)
    return u + guidance_scale * (c - u)


@torch.no_grad()
def diffusion(unet, scheduler, latents, text_embeddings, total_timesteps, guidance_scale, added=None):
    """train_util.py:172-193 (XL: :260-291)."""
    for t in scheduler.timesteps[0:total_timesteps]:
        eps = predict_noise(unet, scheduler, t, latents, text_embeddings, guidance_scale, added)
        latents = scheduler.step(eps, t, latents).prev_sample
    return latents


def esd_loss(target, positive, neutral, unconditional, guidance_scale, action):    # prompt_util.py:107-148
    sign = -1.0 if action == "erase" else 1.0
    return F.mse_loss(target, neutral + sign * guidance_scale * (positive - unconditional))


def leco_step(unet, network, scheduler, emb, latents, k, max_denoising_steps, guidance_scale=1.0, action="erase",
              batch_size=1, pooled=None, add_time_ids=None):
    """emb: dict target/positive/neutral/unconditional -> (1,77,C).  SDXL (train_lora_xl.py:160-354): ``pooled`` is
    the same dict of (1,P) pooled embeddings and ``add_time_ids`` the (1,6) size/crop vector.  Returns a dict with the
    denoised latents, the four predictions, the loss (graph attached) -- the caller does backward/optimizer."""
    def added(name):
        if pooled is None:
            return None
        return {"text_embeds": concat_embeddings(pooled["unconditional"], pooled[name], batch_size),
                "time_ids": concat_embeddings(add_time_ids, add_time_ids, batch_size)}
    with torch.no_grad():
        scheduler.set_timesteps(max_denoising_steps)
        with network:
            denoised = diffusion(unet, scheduler, latents,
                                 concat_embeddings(emb["unconditional"], emb["target"], batch_size), k, 3, added("target"))
        scheduler.set_timesteps(1000)
        t_cur = scheduler.timesteps[int(k * 1000 / max_denoising_steps)]
        preds = {}
        for name in ("positive", "neutral", "unconditional"):
            preds[name] = predict_noise(unet, scheduler, t_cur, denoised,
                                        concat_embeddings(emb["unconditional"], emb[name], batch_size), 1, added(name)).float()
    with network:
        preds["target"] = predict_noise(unet, scheduler, t_cur, denoised,
                                        concat_embeddings(emb["unconditional"], emb["target"], batch_size), 1,
                                        added("target")).float()
    loss = esd_loss(preds["target"], preds["positive"], preds["neutral"], preds["unconditional"], guidance_scale, action)
    return dict(denoised=denoised, preds=preds, loss=loss, t_cur=int(t_cur))
