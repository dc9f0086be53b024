"""Pins the ORACLE: its restatements of the reference loop body / LoRA / loss / DDIM must reproduce,
bit for bit in fp32 on CPU, the golden vectors that the reference's own lora.py, train_util.py and
prompt_util.py produced (tests/golden/make_golden.py).  Also the known answers that exist without
diffusers: public parameter counts, LoRA module census, DDIM timestep tables."""
import contextlib
import io
import os

import pytest
import torch
from safetensors.torch import load_file

from conftest import rel_err
from oracle import lora_ref, step_ref
from oracle import unet_ref as R
from oracle.ddim_ref import DDIMSchedulerRef

GOLD = load_file(os.path.join(os.path.dirname(__file__), "golden", "tiny_step.safetensors"))
bf = torch.bfloat16


def _unet():
    u = R.init_synthetic_(R.UNet2DConditionModel(R.tiny_config()), seed=1234)
    with torch.no_grad():
        for p in u.parameters():
            p.copy_(p.to(bf).float())
    return u.requires_grad_(False)


@pytest.mark.parametrize("name,cfg,params,loras,lora_params,rank", [
    ("sd15", R.sd15_config, 859520964, 192, 1695744, 4),
    ("sd21", R.sd21_config, 865910724, 192, 1728512, 4),
    ("sdxl", R.sdxl_config, 2567463684, 722, 42557440, 16)])
def test_public_parameter_counts_and_lora_census(name, cfg, params, loras, lora_params, rank):
    with torch.device("meta"):
        m = R.UNet2DConditionModel(cfg())
        assert sum(p.numel() for p in m.parameters()) == params
        with contextlib.redirect_stdout(io.StringIO()):
            net = lora_ref.LoRANetworkRef(m, rank=rank)
    assert len(net.unet_loras) == loras
    assert sum(p.numel() for p in net.parameters()) == lora_params


def test_ddim_tables():
    s = DDIMSchedulerRef()
    s.set_timesteps(50)
    assert s.timesteps.tolist() == list(range(980, -1, -20))
    s.set_timesteps(1000)
    assert s.timesteps.tolist() == list(range(999, -1, -1))
    # closed form of the scaled-linear schedule
    b = torch.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000, dtype=torch.float64) ** 2
    ac = torch.cumprod(1 - b, 0)
    assert torch.allclose(s.alphas_cumprod.double(), ac, rtol=1e-5)
    assert abs(float(ac[-1]) - 0.0047) < 2e-4       # SD's terminal alpha_bar


def test_oracle_lora_init_stream_equals_reference_golden():
    torch.manual_seed(42)
    with contextlib.redirect_stdout(io.StringIO()):
        net = lora_ref.LoRANetworkRef(R.UNet2DConditionModel(R.tiny_config()), rank=4)
    assert torch.equal(net.unet_loras[0].lora_down.weight, GOLD["init.first_down"])
    assert torch.equal(net.unet_loras[-1].lora_down.weight, GOLD["init.last_down"])
    want = [l.split(" ")[0] for l in open(os.path.join(os.path.dirname(__file__), "golden", "tiny_lora_keys.txt"))]
    assert list(net.state_dict().keys()) == want


def test_oracle_step_reproduces_reference_golden_bitwise():
    u = _unet()
    with contextlib.redirect_stdout(io.StringIO()):
        net = lora_ref.LoRANetworkRef(u, rank=4, multiplier=1.0, alpha=1.0)
    with torch.no_grad():
        for l in net.unet_loras:
            l.lora_down.weight.copy_(GOLD["lora." + l.lora_name + ".down"])
            l.lora_up.weight.copy_(GOLD["lora." + l.lora_name + ".up"])
    emb = {n: GOLD["emb." + n] for n in ("target", "positive", "neutral", "unconditional")}
    opt = torch.optim.AdamW([{"params": [p for l in net.unet_loras for p in l.parameters()]}], lr=1e-3)
    out = step_ref.leco_step(u, net, DDIMSchedulerRef(), emb, GOLD["latents"].clone(), 3, 10, guidance_scale=2.0)
    assert out["t_cur"] == int(GOLD["step.t_cur"])
    assert torch.equal(out["denoised"], GOLD["step.denoised"])
    for n in ("positive", "neutral", "unconditional", "target"):
        assert torch.equal(out["preds"][n], GOLD["step.pred." + n]), n
    assert torch.equal(out["loss"].detach().reshape(1), GOLD["step.loss"])
    out["loss"].backward()
    g = torch.cat([p.grad.reshape(-1) for l in net.unet_loras for p in (l.lora_down.weight, l.lora_up.weight)])
    assert torch.equal(g, GOLD["step.grads"])
    opt.step()
    pa = torch.cat([p.detach().reshape(-1) for l in net.unet_loras for p in (l.lora_down.weight, l.lora_up.weight)])
    assert torch.equal(pa, GOLD["step.params_after"])


def test_oracle_unet_fp64_self_consistency():
    u = _unet()
    x, ctx = GOLD["unet.x"], GOLD["unet.ctx"]
    with torch.no_grad():
        y32 = u(x, torch.tensor(500), encoder_hidden_states=ctx).sample
        y64 = u.double()(x.double(), torch.tensor(500), encoder_hidden_states=ctx.double()).sample
    assert torch.equal(y32, GOLD["unet.y_t500"])
    assert ((y32.double() - y64).norm() / y64.norm()).item() < 1e-5


@pytest.mark.parametrize("pred", ["epsilon", "v_prediction"])
def test_non_ddim_schedulers_rows_match_the_stepwise_restatement(pred):
    """leco_amd.scheduler's pre-multiplied coefficient rows (what the fused loop consumes) against the step-by-step
    restatement in oracle/sched_ref.py, for every step of a 50-step schedule, plus the public sigma_max."""
    from leco_amd import scheduler as S
    from oracle import sched_ref as R
    g = torch.Generator().manual_seed(5)
    n = 50
    sch = R.SigmaSchedule(n)
    assert abs(sch.init_noise_sigma - 14.6146) < 1e-3
    for name in ("euler_a", "lms", "ddpm"):
        s = S.create_noise_scheduler(name, prediction_type=pred)
        s.set_timesteps(n)
        assert abs(float(s.init_noise_sigma) - (14.6146 if name != "ddpm" else 1.0)) < 1e-3
        x = torch.randn(2, 4, 8, 8, generator=g, dtype=torch.float64) * float(s.init_noise_sigma)
        xr = x.clone()
        hist, hist_r = [], []
        rows = s.rows().double()
        for i, t in enumerate(s.timesteps):
            out = torch.randn(2, 4, 8, 8, generator=g, dtype=torch.float64)
            noise = torch.randn(2, 4, 8, 8, generator=g, dtype=torch.float64)
            # (a) the scheduler object's own step (drop-in path)
            if name == "lms":
                a = s.step(out, t, x).prev_sample
            else:
                a = s.step(out, t, x, noise=noise).prev_sample
            # (b) the coefficient row (fused path), with the derivative history it implies
            r = rows[i]
            h = hist + [torch.zeros_like(x)] * 3
            b = r[0] * x + r[1] * out + r[2] * noise + r[3] * h[0] + r[4] * h[1] + r[5] * h[2]
            hist = [r[7] * x + r[8] * out] + hist[:2]
            # (c) the step-by-step restatement
            if name == "euler_a":
                c = R.euler_a_step(sch, i, xr, out, noise, pred)
            elif name == "lms":
                c = R.lms_step(sch, i, xr, out, hist_r, pred)
            else:
                c = R.ddpm_step(int(t), n, xr, out, noise, pred)
            assert rel_err(a, c) < 1e-5 and rel_err(b, c) < 1e-5, (name, i)
            if name != "ddpm" and i + 1 < n:
                assert abs(float(r[6]) - sch.scale(i + 1)) < 1e-6
            x, xr = a, c
