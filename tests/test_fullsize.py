"""Full-size parity on gfx950 (`-m gpu`) of the launch plans a training step ACTUALLY runs, at the BASELINE.json
configurations -- the round-1 NaN lived in plans no test reached (forward-only B = 2 bs with the fused GEGLU
epilogue and fused LoRA down-projection, batched frozen B = 6 bs, the backward at 64^2 latents).

One `FusedStep.step` with hipGraphs on (exactly what bench.py / train() execute) against ONE iteration of the
reference loop restated on the fp32 oracle (oracle/step_ref.leco_step = train_lora.py:141-281; UNet / DDIM / LoRA
oracles), both on the GPU box:

    denoised latents  <- k passes of the forward-only LoRA-ON plan + CFG/DDIM kernel   (train_util.py:172-193)
    positive / neutral / unconditional predictions <- the batched LoRA-OFF plan         (train_lora.py:202-237)
    target prediction <- the training plan, LoRA ON                                      (train_lora.py:244-256)
    loss, LoRA gradients (backward plan), AdamW-updated parameters                       (train_lora.py:265-281)

Tolerance: relative L2 vs the fp32 oracle, calibrated by the error the SAME oracle graph makes when
run in plain torch bf16 on the same device: rel_hip <= 1.25 * rel_torch_bf16 (floors for quantities whose bf16
error happens to be tiny).  EVERY case measures that calibration in the run (cal = None: the torch-bf16 error is not read
from a table), and every BASELINE configuration is also run at the benchmark's loop depth k = 25.  Results are printed (`-s`)
and copied into profiles/ by tools/gpu_round_run.sh."""
import contextlib
import io
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import pytest
import torch

from conftest import rel_err
from leco_amd import model_util, prompt_util
from leco_amd.lora import DEFAULT_TARGET_REPLACE, UNET_TARGET_REPLACE_MODULE_CONV, LoRANetwork
from leco_amd.scheduler import create_noise_scheduler
from leco_amd.train import FusedStep
from leco_amd.unet import UNet2DConditionModel
from oracle import lora_ref, step_ref
from oracle import unet_ref as R
from oracle.ddim_ref import DDIMSchedulerRef

bf = torch.bfloat16
pytestmark = pytest.mark.gpu
NAMES = ("target", "positive", "neutral", "unconditional")
ARCH = {"sd15": (R.sd15_config, 768), "sd21": (R.sd21_config, 1024), "sdxl": (R.sdxl_config, 2048)}


def _device():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from conftest import _bind_hip
    _bind_hip()
    return torch.device("cuda:0")


def _models(arch, dev, seed, m):
    """Oracle + HIP model with the same bf16-representable synthetic weights.  Built and initialised ON the device
    (SURVEY 8d rule: U(+-1/sqrt(fan_in)) matrices, norm gamma 1 / beta 0, small biases): the CPU constructors' default
    init alone costs ~20 s per SD1.5-sized model."""
    with torch.device(dev):
        ref = R.UNet2DConditionModel(ARCH[arch][0]())
    gen = torch.Generator(device=dev).manual_seed(seed)
    with torch.no_grad():
        for name, p in ref.named_parameters():
            if p.ndim >= 2:
                p.copy_(((torch.rand(p.shape, generator=gen, device=dev) * 2 - 1) / p[0].numel() ** 0.5).to(bf).float())
            elif "norm" in name:
                p.fill_(1.0 if name.endswith("weight") else 0.0)
            else:
                p.copy_(((torch.rand(p.shape, generator=gen, device=dev) * 2 - 1) * 0.02).to(bf).float())
    ref.requires_grad_(False)
    m.load_state_dict(ref.state_dict())
    m.to(dev, bf)
    m.requires_grad_(False)
    return ref


def _loras(ref, m, rank, c3lier, gen):
    targets = list(DEFAULT_TARGET_REPLACE) + (list(UNET_TARGET_REPLACE_MODULE_CONV) if c3lier else [])
    with contextlib.redirect_stdout(io.StringIO()):
        rnet = lora_ref.LoRANetworkRef(ref, rank=rank, targets=targets)
        net = LoRANetwork(m, rank=rank, multiplier=1.0, alpha=1.0, target_replace_modules=targets)
    assert [l.lora_name for l in rnet.unet_loras] == [l.lora_name for l in net.unet_loras]
    dev = next(ref.parameters()).device
    rnet.to(dev)
    with torch.no_grad():
        for rl, l in zip(rnet.unet_loras, net.unet_loras):
            fan_in = rl.lora_down.weight[0].numel()
            d = ((torch.rand(rl.lora_down.weight.shape, generator=gen) * 2 - 1) / fan_in ** 0.5).to(bf).float()
            u = (torch.randn(rl.lora_up.weight.shape, generator=gen) * 0.02).to(bf).float()
            rl.lora_down.weight.copy_(d)
            rl.lora_up.weight.copy_(u)
            l.lora_down.weight.copy_(d.reshape(l.lora_down.weight.shape))
            l.lora_up.weight.copy_(u.reshape(l.lora_up.weight.shape))
    net.mark_updated()
    return rnet, net


def _ref_grads(rnet):
    return torch.cat([p.grad.reshape(-1).float() for l in rnet.unet_loras for p in (l.lora_down.weight, l.lora_up.weight)])


def _oracle_step(ref, rnet, emb, lat, k, bs, gscale, action, dtype, pooled=None, ids=None, v_pred=False):
    """One reference iteration on the oracle graph in `dtype`; returns tensors in fp32 + the LoRA gradients."""
    ref.to(dtype)
    rnet.to(dtype)
    for p in rnet.parameters():
        p.grad = None
    sched = DDIMSchedulerRef(prediction_type="v_prediction" if v_pred else "epsilon")
    e = {n: v.to(dtype) for n, v in emb.items()}
    pl = None if pooled is None else {n: v.to(dtype) for n, v in pooled.items()}
    out = step_ref.leco_step(ref, rnet, sched, e, lat.to(dtype), k, 50, guidance_scale=gscale, action=action, batch_size=bs,
                             pooled=pl, add_time_ids=None if ids is None else ids.to(dtype))
    if os.environ.get("LECO_FS_NOBWD"):      # experiment switch (segfault hunt): skip the oracle's autograd pass
        for p in rnet.parameters():
            p.grad = torch.zeros_like(p)
    else:
        out["loss"].float().backward()
    res = dict(denoised=out["denoised"].float(), loss=float(out["loss"].detach()), grads=_ref_grads(rnet), t_cur=out["t_cur"])
    res.update({n: out["preds"][n].detach().float() for n in NAMES})
    ref.to(torch.float32)
    rnet.to(torch.float32)
    return res


def _check_step(arch, res, bs, rank, k, **kw):
    dev = _device()
    with torch.device(dev):
        m = UNet2DConditionModel(model_util.SYNTHETIC[arch]())
    try:
        return _check_step_on(dev, m, arch, res, bs, rank, k, **kw)
    finally:        # several multi-GB models run in one process: give graphs and buffers back before the next one
        # (LECO_FS_KEEP: experiment switches of the capture-crash hunt, DESIGN.md section 6 -- "release": keep the graph
        # execs, "cache": keep torch's cached blocks)
        keep = os.environ.get("LECO_FS_KEEP", "")
        if "release" not in keep:
            m.release()
        else:
            globals().setdefault("_alive", []).append(m)
        del m
        import gc
        gc.collect()
        torch.cuda.synchronize()
        if "cache" not in keep:
            torch.cuda.empty_cache()


def _check_step_on(dev, m, arch, res, bs, rank, k, c3lier=False, v_pred=False, gscale=1.0, action="erase", seed=1234,
                   lr=1e-4, cal=None, dedup=False):
    ref = _models(arch, dev, seed, m)
    g = torch.Generator().manual_seed(seed + 1)
    rnet, net = _loras(ref, m, rank, c3lier, g)
    cdim = ARCH[arch][1]
    xl = arch == "sdxl"
    emb = {n: torch.randn(1, 77, cdim, generator=g).to(bf).float().to(dev) for n in NAMES}
    pooled = {n: torch.randn(1, 1280, generator=g).to(bf).float().to(dev) for n in NAMES} if xl else None
    ids = torch.tensor([[float(res), float(res), 0.0, 0.0, float(res), float(res)]], device=dev) if xl else None
    lat = torch.randn(bs, 4, res // 8, res // 8, generator=g).to(dev)
    gold = _oracle_step(ref, rnet, emb, lat, k, bs, gscale, action, torch.float32, pooled, ids, v_pred)
    if cal is None or os.environ.get("LECO_FULLSIZE_CALIBRATE"):
        # the torch-bf16 error of the same oracle graph on this device (the calibration the tolerances come from)
        cal_out = _oracle_step(ref, rnet, emb, lat, k, bs, gscale, action, bf, pooled, ids, v_pred)
        cal = {n: rel_err(cal_out[n], gold[n]) for n in ("denoised", "grads") + NAMES}
        cal["loss"] = abs(cal_out["loss"] - gold["loss"]) / gold["loss"]
    else:
        cal = dict(cal, **{n: cal["pred"] for n in NAMES})
    # ---- the HIP path, as train() / bench.py run it
    m.use_graphs = True
    if xl:
        mk = lambda n: prompt_util.PromptEmbedsXL(emb[n], pooled[n])
    else:
        mk = lambda n: emb[n]
    settings = prompt_util.PromptSettings(target="t", positive="p", neutral="n", unconditional="u", guidance_scale=gscale,
                                          batch_size=bs, resolution=res, action=action)
    pair = prompt_util.PromptEmbedsPair(torch.nn.MSELoss(), mk("target"), mk("positive"), mk("unconditional"), mk("neutral"),
                                        settings)
    sched = create_noise_scheduler("ddim", prediction_type="v_prediction" if v_pred else "epsilon")
    fs = FusedStep(m, net, sched, 50, lr=lr, dedup=dedup)
    before = net.slab.detach()[:net.numel].clone()
    loss = fs.step(pair, k, lat.clone(), add_time_ids=ids)
    torch.cuda.synchronize()
    st = fs._state[(bs, res // 8, res // 8)]
    if dedup:      # the guidance-1 passes ran on the conditional samples only: the predictions ARE the cond halves
        last = st["last"]
        assert last["dedup"] and last["plan"].pred.shape[0] == bs and last["fplan"].pred.shape[0] == 3 * bs     # four distinct prompts here
        got = dict(denoised=st["x"], target=last["plan"].pred, grads=net.grad[:net.numel])
        got.update({n: last["preds"][n] for n in ("positive", "neutral", "unconditional")})
    else:
        got = dict(denoised=st["x"], target=st["plan"].pred[bs:], grads=net.grad[:net.numel])
        got.update({n: st["preds"][n][bs:] for n in ("positive", "neutral", "unconditional")})
    err = {n: rel_err(got[n], gold[n]) for n in got}
    err["loss"] = abs(loss.item() - gold["loss"]) / gold["loss"]
    print(f"\n{arch} {res}^2 bs={bs} rank={rank}{' c3lier' if c3lier else ''}{' v-pred' if v_pred else ''} k={k} "
          f"t_cur={gold['t_cur']} loss={loss.item():.4e} (oracle {gold['loss']:.4e})")
    for n in ("denoised",) + NAMES + ("loss", "grads"):
        print(f"    {n:14s} rel_hip={err[n]:.3e}   rel_torch_bf16={cal[n]:.3e}")
    assert all(torch.isfinite(v.float()).all() for v in got.values()) and torch.isfinite(loss).all()
    for n in ("denoised",) + NAMES:
        assert err[n] <= max(1.25 * cal[n], 2e-3), (n, err[n], cal[n])
    assert err["loss"] <= max(1.25 * cal["loss"], 3e-2), (err["loss"], cal["loss"])
    assert err["grads"] <= max(1.25 * cal["grads"], 3e-2), (err["grads"], cal["grads"])
    # AdamW moved every parameter that has a gradient, by at most lr (first step: |update| <= lr (1 + wd |p|))
    after = net.slab.detach()[:net.numel]
    delta = (after - before).abs()
    assert torch.isfinite(after).all() and delta.max().item() <= lr * 1.05 and (delta > 0).float().mean().item() > 0.9
    # a second step on the updated parameters stays finite (re-packed LoRA operands, graph replay)
    loss2 = fs.step(pair, 1, lat.clone(), add_time_ids=ids)
    assert torch.isfinite(loss2).all() and torch.isfinite(net.grad).all()
    return err, cal


# `cal = None`: the torch-bf16 error of the SAME oracle graph is measured in the run (it doubles the oracle time of a case).
CASES = {
    # BASELINE config 2 (the headline benchmark shape): SD1.5, 512^2, prompt batch 2, rank-4 lierla
    # (cal = None: the torch-bf16 error the tolerances are calibrated by is re-measured in the run, not read from a table)
    "sd15_512_bs2_rank4": dict(arch="sd15", res=512, bs=2, rank=4, k=2, cal=None),
    # the benchmark's loop depth: k = 25 (the mean of the reference's randint(1, 50)) and k = 49 (its maximum; the frozen /
    # target passes then run at t = 999 - 20 * 49 = 19): DDIM error compounding through k replays of the forward-only
    # graph, the device-side t_idx advance and the CFG / DDIM kernel (train_util.py:172-193, train_lora.py:148-199)
    "sd15_512_bs2_rank4_k25": dict(arch="sd15", res=512, bs=2, rank=4, k=25, seed=2025, cal=None),
    "sd15_512_bs2_rank4_k49": dict(arch="sd15", res=512, bs=2, rank=4, k=49, seed=2049, cal=None),
    # the de-duplicated pass structure (FusedStep.dedup, train()'s default): guidance-1 passes on the conditional samples only --
    # same oracle (the reference loop), same tolerances, at k = 2 and at the benchmark's loop depth
    "sd15_512_bs2_rank4_dedup": dict(arch="sd15", res=512, bs=2, rank=4, k=2, cal=None, dedup=True),
    "sd15_512_bs2_rank4_k25_dedup": dict(arch="sd15", res=512, bs=2, rank=4, k=25, seed=2025, cal=None, dedup=True),
    # same shapes, the other branch of the objective (action = enhance, guidance_scale 3), k = 3
    "sd15_512_bs2_rank4_enhance_g3": dict(arch="sd15", res=512, bs=2, rank=4, k=3, gscale=3.0, action="enhance", seed=4321, cal=None),
    # BASELINE config 3: SD2.1 (linear projections, head dim 64), v-prediction, 768^2, prompt batch 2 -- at k = 2 and at the
    # benchmark's loop depth k = 25 (the v-prediction DDIM update compounding through 25 replays)
    "sd21_768_bs2_rank4_vpred": dict(arch="sd21", res=768, bs=2, rank=4, k=2, v_pred=True, seed=77, cal=None),
    "sd21_768_bs2_rank4_vpred_k25": dict(arch="sd21", res=768, bs=2, rank=4, k=25, v_pred=True, seed=2077, cal=None),
    # BASELINE config 4: SD1.5, rank-8 c3lier (conv + time_emb_proj LoRA, 278 modules), 512^2, prompt batch 4
    "sd15_512_bs4_rank8_c3lier": dict(arch="sd15", res=512, bs=4, rank=8, k=2, c3lier=True, seed=99, cal=None),
    "sd15_512_bs4_rank8_c3lier_k25": dict(arch="sd15", res=512, bs=4, rank=8, k=25, c3lier=True, seed=2099, cal=None),
    # BASELINE config 5: SDXL (depth 2 / 10 transformers, text_time add-embedding), rank 16, 1024^2, prompt batch 1
    "sdxl_1024_bs1_rank16": dict(arch="sdxl", res=1024, bs=1, rank=16, k=2, seed=5, cal=None),
    "sdxl_1024_bs1_rank16_k25": dict(arch="sdxl", res=1024, bs=1, rank=16, k=25, seed=2005, cal=None),
}


def _check_dynamic_resolution(buckets=((448, 320), (256, 384), (448, 320)), bs=2, rank=4, k=2, seed=606):
    """`dynamic_resolution` (train_lora.py:160-170, train_util.py:404-416; the configuration of the reference's one
    published number): consecutive steps of ONE FusedStep at non-square buckets in ONE process -- latents 56x40 (levels
    28x20, 14x10, 7x5: ragged M tiles, shapes the tuner table has never seen, patch-conv geometries that fall back), then
    32x48, then 56x40 again with only ONE bucket allowed resident, so every change of bucket evicts the previous plans
    and graphs and the third step re-builds and re-captures what the first one had.  lr = 0: the LoRA parameters stay
    put, so each step is compared with the oracle on the same weights."""
    dev = _device()
    with torch.device(dev):
        m = UNet2DConditionModel(model_util.SYNTHETIC["sd15"]())
    ref = _models("sd15", dev, seed, m)
    g = torch.Generator().manual_seed(seed + 1)
    rnet, net = _loras(ref, m, rank, False, g)
    emb = {n: torch.randn(1, 77, 768, generator=g).to(bf).float().to(dev) for n in NAMES}
    m.use_graphs = True
    settings = prompt_util.PromptSettings(target="t", positive="p", neutral="n", unconditional="u", guidance_scale=1.0,
                                          batch_size=bs, resolution=512, dynamic_resolution=True, action="erase")
    pair = prompt_util.PromptEmbedsPair(torch.nn.MSELoss(), emb["target"], emb["positive"], emb["unconditional"],
                                        emb["neutral"], settings)
    fs = FusedStep(m, net, create_noise_scheduler("ddim"), 50, lr=0.0, weight_decay=0.0)
    fs.MAX_BUCKETS = 1
    before = net.slab.detach().clone()
    for i, (hh, ww) in enumerate(buckets):
        lat = torch.randn(bs, 4, hh // 8, ww // 8, generator=g).to(dev)
        gold = _oracle_step_hw(ref, rnet, emb, lat, k, bs, torch.float32)
        cal_out = _oracle_step_hw(ref, rnet, emb, lat, k, bs, bf)
        cal = {n: rel_err(cal_out[n], gold[n]) for n in ("denoised", "grads") + NAMES}
        loss = fs.step(pair, k, lat.clone())
        torch.cuda.synchronize()
        assert len(fs._state) == 1, "one resident bucket: the previous plans must have been evicted"
        st = fs._state[(bs, hh // 8, ww // 8)]
        got = dict(denoised=st["x"], target=st["plan"].pred[bs:], grads=net.grad[:net.numel])
        got.update({n: st["preds"][n][bs:] for n in ("positive", "neutral", "unconditional")})
        err = {n: rel_err(got[n], gold[n]) for n in got}
        lerr = abs(loss.item() - gold["loss"]) / gold["loss"]
        print(f"\nsd15 dynamic_resolution step {i}: {hh}x{ww} (latents {hh // 8}x{ww // 8}) bs={bs} k={k} loss={loss.item():.4e} "
              f"(oracle {gold['loss']:.4e}, rel {lerr:.2e})")
        for n in ("denoised",) + NAMES + ("grads",):
            print(f"    {n:14s} rel_hip={err[n]:.3e}   rel_torch_bf16={cal[n]:.3e}")
        assert all(torch.isfinite(v.float()).all() for v in got.values()) and torch.isfinite(loss).all()
        for n in ("denoised",) + NAMES:
            assert err[n] <= max(1.25 * cal[n], 2e-3), (i, n, err[n], cal[n])
        assert err["grads"] <= max(1.25 * cal["grads"], 3e-2), (i, err["grads"], cal["grads"])
    assert torch.equal(net.slab.detach(), before)


def _check_two_arithmetics(steps=20, bs=2, rank=4, res=512, seed=4242):
    """Round-5 verdict, "two arithmetics feed one subtraction": the frozen predictions come from the forward-only plan (stripe
    kernels at level 0, fused GEGLU, batch-shared prefix), the target prediction from the per-op training plan; the two differ
    by bf16 rounding (~7e-3), and in the reference they are the same function (LoRA-on with lora_up = 0 is bit-identical to
    LoRA-off).  Is that difference a bias or noise?  Train `steps` optimizer steps twice from the same LoRA, noise and k
    sequence: (a) as shipped, (b) with the VALUE of the target prediction taken from the forward-only plan as well
    (FusedStep.target_from_forward_only; the backward still differentiates the training plan).  The two LoRAs must stay as
    close to each other as two runs of (a) that differ only by the fp32-atomic order of the gradient sums (run (a) twice)."""
    dev = _device()
    with torch.device(dev):
        m = UNet2DConditionModel(model_util.SYNTHETIC["sd15"]())
    g0 = torch.Generator().manual_seed(seed)
    ref = _models("sd15", dev, seed, m)
    del ref
    g = torch.Generator().manual_seed(seed + 1)
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        net = LoRANetwork(m, rank=rank, multiplier=1.0, alpha=1.0).to(dev)
    with torch.no_grad():
        for l in net.unet_loras:
            l.lora_up.weight.copy_((torch.randn(l.lora_up.weight.shape, generator=g) * 0.02).to(dev))
    net.mark_updated()
    emb = {n: torch.randn(1, 77, 768, generator=g).to(bf).float().to(dev) for n in NAMES}
    m.use_graphs = True
    settings = prompt_util.PromptSettings(target="t", positive="p", neutral="n", unconditional="u", guidance_scale=1.0,
                                          batch_size=bs, resolution=res, action="erase")
    pair = prompt_util.PromptEmbedsPair(torch.nn.MSELoss(), emb["target"], emb["positive"], emb["unconditional"], emb["neutral"],
                                        settings)
    slab0 = net.slab.detach().clone()
    ks = [int(torch.randint(1, 8, (1,), generator=g0)) for _ in range(steps)]
    lats = [torch.randn(bs, 4, res // 8, res // 8, generator=g0) for _ in range(steps)]

    def run(from_fwd_only):
        with torch.no_grad():
            net.slab.copy_(slab0)
            net.exp_avg.zero_(); net.exp_avg_sq.zero_()
        net.sync_shadow(); net.mark_updated()
        fs = FusedStep(m, net, create_noise_scheduler("ddim"), 50, lr=1e-4)
        fs.target_from_forward_only = from_fwd_only
        losses = [fs.step(pair, ks[i], lats[i].clone()).item() for i in range(steps)]
        torch.cuda.synchronize()
        return (net.slab.detach()[:net.numel] - slab0[:net.numel]).clone(), losses
    da, la = run(False)
    da2, _ = run(False)
    db, lb = run(True)

    def cos(x, y):
        return (x @ y / (x.norm() * y.norm())).item()
    noise, diff = rel_err(da2, da), rel_err(db, da)
    print(f"\ntwo arithmetics, {steps} steps: LoRA update |d|={da.norm().item():.4e}; (a) vs (a) again: rel {noise:.3e} cos {cos(da, da2):.5f}; "
          f"(b) target value from the forward-only plan vs (a): rel {diff:.3e} cos {cos(da, db):.5f}; last loss {la[-1]:.4e} / {lb[-1]:.4e}")
    assert all(torch.isfinite(torch.tensor(x)).all() for x in (la, lb))
    # AdamW's first steps move every parameter by ~lr sign(g): elements whose gradient is near zero flip under ANY perturbation,
    # so the updates are compared as directions, and against the run-to-run noise of the shipped path itself
    assert cos(da, db) > 0.9 and diff <= max(3.0 * noise, 0.35), (noise, diff)


def _oracle_step_hw(ref, rnet, emb, lat, k, bs, dtype):
    return _oracle_step(ref, rnet, emb, lat, k, bs, 1.0, "erase", dtype)


def test_full_size_dynamic_resolution_buckets():
    """Own interpreter, like the cases below."""
    import subprocess
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "dynamic_resolution"], capture_output=True, text=True,
                       timeout=1500, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    print("\n" + "\n".join(l for l in r.stdout.splitlines() if "rel_hip" in l or "loss=" in l))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "PASS dynamic_resolution" in r.stdout


def test_full_size_two_arithmetics_train_to_the_same_lora():
    """Own interpreter, like the cases below."""
    import subprocess
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "two_arithmetics"], capture_output=True, text=True,
                       timeout=1500, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    print("\n" + "\n".join(l for l in r.stdout.splitlines() if "two arithmetics" in l))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "PASS two_arithmetics" in r.stdout


@pytest.mark.parametrize("case", list(CASES))
def test_full_size_step_vs_oracle(case):
    """Each configuration runs in its own interpreter (`python tests/test_fullsize.py <case>`): one process then holds
    one fp32 oracle + one HIP model, and a crash in one configuration cannot take the rest of the suite down.
    (Open issue, DESIGN.md section 8: capturing graphs for a SECOND multi-GB model in a process that has run the fp32
    oracle's autograd backward on the GPU segfaults inside hipStreamBeginCapture on ROCm 7.2.)"""
    import os
    import subprocess
    import sys
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    r = subprocess.run([sys.executable, os.path.abspath(__file__), case], capture_output=True, text=True, timeout=1500,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    out = "\n".join(l for l in r.stdout.splitlines() if "rel_hip" in l or "loss=" in l)
    print("\n" + out)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "PASS " + case in r.stdout


if __name__ == "__main__":
    import sys
    for name in sys.argv[1:]:
        try:
            if name == "dynamic_resolution":
                _check_dynamic_resolution()
            elif name == "two_arithmetics":
                _check_two_arithmetics()
            else:
                _check_step(**CASES[name])
        except AssertionError:
            if not os.environ.get("LECO_FS_NOBWD"):
                raise
        print("PASS " + name, flush=True)
