"""Data-parallel path on CPU: 2 processes, gloo, the host emulator as kernel backend.  Each rank runs a
full fused step on its own noise with the SAME k; the only collective is the all-reduce of the flat LoRA
gradient slab.  Afterwards the parameters must be identical on both ranks and equal to a single-process
run that averages the two ranks' gradients."""
import contextlib
import io
import math
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _setup():
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    os.environ.setdefault("LECO_EMU_THREADS", "4")
    import build_emu
    from leco_amd import hip
    hip._use_library(build_emu.build())


def _make(world, pg=None):
    from leco_amd import model_util, prompt_util
    from leco_amd.lora import LoRANetwork
    from leco_amd.scheduler import create_noise_scheduler
    from leco_amd.train import FusedStep
    from leco_amd.unet import UNet2DConditionModel
    m = model_util.init_synthetic_(UNet2DConditionModel(model_util.tiny_config()), 1234).to(torch.bfloat16)
    m.requires_grad_(False)
    torch.manual_seed(1234)
    with contextlib.redirect_stdout(io.StringIO()):
        net = LoRANetwork(m, rank=4)
    g = torch.Generator().manual_seed(3)
    with torch.no_grad():
        for l in net.unet_loras:
            l.lora_up.weight.copy_(torch.randn(l.lora_up.weight.shape, generator=g) * 0.05)
    net.mark_updated()
    eg = torch.Generator().manual_seed(5)
    emb = [torch.randn(1, 77, 64, generator=eg) * 3 for _ in range(4)]
    s = prompt_util.PromptSettings(target="t", positive="p", neutral="n", unconditional="u", guidance_scale=2.0,
                                   batch_size=1, resolution=128)
    pair = prompt_util.PromptEmbedsPair(torch.nn.MSELoss(), emb[0], emb[1], emb[2], emb[3], s)
    fs = FusedStep(m, net, create_noise_scheduler("ddim"), 10, lr=1e-3, world_size=world, process_group=pg)
    return fs, net, pair


def _lat(rank):
    return torch.randn(1, 4, 8, 8, generator=torch.Generator().manual_seed(100 + rank))


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    _setup()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    fs, net, pair = _make(world)
    fs.step(pair, 1, _lat(rank))
    # numpy, not torch: a tensor travels through the queue as a file descriptor that the parent fetches from THIS
    # process -- which may have exited by then (FileNotFoundError in resource_sharer); an ndarray is pickled by value
    q.put((rank, net.slab.detach()[:net.numel].numpy().copy(), net.grad[:net.numel].numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("deterministic", [False, True])
def test_two_rank_step_equals_gradient_average(deterministic, monkeypatch):
    """deterministic: LECO_DETERMINISTIC=1 + LECO_GN_FUSED=0 (no fp32 atomics anywhere in a step) -- the all-reduced slab is
    then BIT-EQUAL to the single-process sum of the two ranks' gradients."""
    if deterministic:
        monkeypatch.setenv("LECO_DETERMINISTIC", "1")
        monkeypatch.setenv("LECO_GN_FUSED", "0")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict()
    for _ in range(2):
        r, slab, grad = q.get(timeout=600)
        slab, grad = torch.from_numpy(slab), torch.from_numpy(grad)
        res[r] = (slab, grad)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert torch.equal(res[0][0], res[1][0]), "ranks diverged"
    assert torch.equal(res[0][1], res[1][1]), "all-reduced gradient slabs differ"
    # single-process reference: sum of the two ranks' gradients, AdamW with grad_scale 1/2
    _setup()
    from leco_amd import ops
    grads = []
    for r in range(2):
        fs, net, pair = _make(1)
        fs.step(pair, 1, _lat(r))
        grads.append(net.grad.clone())
    fs, net, pair = _make(1)
    net.grad.copy_(grads[0] + grads[1])
    # two separate runs of a step differ at the bf16-noise level (GroupNorm statistics are reduced with
    # fp32 atomics, so rounding flips are not reproducible run to run): compare in relative L2
    def rel(a, b):
        return ((a - b).norm() / b.norm()).item()
    assert rel(net.grad[:net.numel], res[0][1]) < 0.2
    if deterministic:
        assert torch.equal(net.grad[:net.numel], res[0][1]), "deterministic mode: all-reduced slab != single-process sum"
    net.hyper.copy_(torch.tensor([1e-3, 1 - 0.9, 1 - 0.999, 0.5]))
    ops.adamw(net.slab.detach(), net.grad, net.exp_avg, net.exp_avg_sq, net.shadow, net.hyper, 0.9, 0.999, 1e-8, 1e-2,
              net.slab.numel()).run()
    # AdamW applied to the REDUCED gradient of the 2-rank run must reproduce the ranks' parameters exactly
    fs2, net2, _ = _make(1)
    net2.grad[:net2.numel].copy_(res[0][1])
    net2.hyper.copy_(torch.tensor([1e-3, 1 - 0.9, 1 - 0.999, 0.5]))
    ops.adamw(net2.slab.detach(), net2.grad, net2.exp_avg, net2.exp_avg_sq, net2.shadow, net2.hyper, 0.9, 0.999, 1e-8,
              1e-2, net2.slab.numel()).run()
    assert torch.equal(net2.slab.detach()[:net2.numel], res[0][0])
    assert rel(net.slab.detach()[:net.numel], res[0][0]) < 2e-2


# ---------------------------------------------------------------------------------------------------------
# train() itself under data parallelism (every rank used to draw the SAME prompt pair and noise)
def _train_worker(rank, world, port, q, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    _setup()
    from leco_amd import config_util, prompt_util, train as T
    seen = []
    orig = T.FusedStep.step

    def spy(self, pair, timesteps_to, latents, **kw):
        seen.append((int(timesteps_to), float(latents.double().sum()), pair.target.flatten()[0].item()))
        return orig(self, pair, timesteps_to, latents, **kw)
    T.FusedStep.step = spy
    cfg = config_util.RootConfig(
        prompts_file="unused", pretrained_model=dict(name_or_path="synthetic:tiny"),
        network=dict(type="lierla", rank=4, alpha=1.0),
        train=dict(precision="bfloat16", noise_scheduler="ddim", iterations=4, lr=1e-3, optimizer="AdamW",
                   lr_scheduler="constant", max_denoising_steps=4),
        save=dict(name="dp", path=out_dir, per_steps=100), logging={}, other={})
    prompts = [prompt_util.PromptSettings(target=t, positive=t, unconditional="", neutral="", action="erase",
                                          guidance_scale=1.0, resolution=64, batch_size=1) for t in ("van gogh", "monet")]
    with contextlib.redirect_stdout(io.StringIO()):
        net, _ = T.train(cfg, prompts, device=torch.device("cpu"), use_graphs=False, progress=False, save_state=True,
                         stop_after=1)
    q.put((rank, seen, net.slab.detach()[:net.numel].numpy().copy()))      # by value (see _worker)
    dist.barrier()
    dist.destroy_process_group()


def test_train_under_dp_draws_per_rank_data_and_a_shared_k(tmp_path):
    """rank r draws its own prompt pair and noise, `k` comes from a shared-seed generator, the LoRA
    parameters stay identical; the resumable state holds one RNG stream per rank."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_train_worker, args=(r, 2, port, q, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        r, seen, slab = q.get(timeout=900)
        slab = torch.from_numpy(slab)
        res[r] = (seen, slab)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    s0, s1 = res[0][0], res[1][0]
    assert len(s0) == len(s1) == 2
    assert [k for k, _, _ in s0] == [k for k, _, _ in s1], "ranks must run the same number of denoising passes"
    assert all(a[1] != b[1] for a, b in zip(s0, s1)), "ranks drew identical initial latents"
    assert torch.equal(res[0][1], res[1][1]), "LoRA parameters diverged across ranks"
    blob = torch.load(tmp_path / "dp_state.pt", map_location="cpu", weights_only=True)
    assert len(blob["rng"]) == 2 and not torch.equal(blob["rng"][0]["cpu"], blob["rng"][1]["cpu"])


def test_bench_multi_gpu_launch_path_dry_run():
    """`python bench.py --gpus 2` end to end WITHOUT GPUs: the self-relaunch under torch.distributed.run (one process per
    rank, 127.0.0.1 rendezvous), per-rank build lock and library load, barrier-bracketed timing, max over ranks, rank 0's
    single JSON line -- with the host emulator as kernel backend and gloo (LECO_BENCH_EMU=1).  Asserts what the driver's
    scaling run relies on: whole-job value = world * steps / time, global batch = world * bs, the same k on every rank,
    exactly one data-path collective (the LoRA gradient all-reduce) per step."""
    import json
    import subprocess
    env = dict(os.environ, LECO_BENCH_EMU="1", LECO_EMU_THREADS="4", OMP_NUM_THREADS="2")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--arch", "tiny", "--res", "64", "--bs", "1", "--k", "1", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=1500, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]          # rank 0 only, one line
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["warmup"] == 1 and out["scaling"] == "weak"
    # the second timed loop (de-duplicated pass structure) ran on both ranks too: same collectives, its own line
    assert out["dedup"]["distinct_frozen_prompts"] == 2 and math.isfinite(out["dedup"]["loss"]) and out["dedup"]["value"] > 0
    assert out["config"]["global_batch"] == 2 and out["config"]["parallelism"] == "dp2"
    assert out["config"]["collectives_per_step"] == 1.0 and out["config"]["k_identical_across_ranks"] is True
    assert abs(out["value"] - 2 * 2 / (out["ms_per_step"] * 2 / 1e3)) < 1e-6 * out["value"]
    assert "EMULATOR" in out["data"] and all(l == l for l in out["config"]["losses"])       # finite losses (NaN != NaN)


def _shape_worker(rank, world, port, q, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    _setup()
    from leco_amd import config_util, prompt_util, train as T
    seen = []
    orig = T.FusedStep.step

    def spy(self, pair, timesteps_to, latents, **kw):
        seen.append((tuple(latents.shape), pair.target.flatten()[0].item()))
        return orig(self, pair, timesteps_to, latents, **kw)
    T.FusedStep.step = spy
    cfg = config_util.RootConfig(
        prompts_file="unused", pretrained_model=dict(name_or_path="synthetic:tiny"),
        network=dict(type="lierla", rank=4, alpha=1.0),
        train=dict(precision="bfloat16", noise_scheduler="ddim", iterations=4, lr=1e-3, optimizer="AdamW",
                   lr_scheduler="constant", max_denoising_steps=2),
        save=dict(name="dpshape", path=out_dir, per_steps=100), logging={}, other={})
    mk = lambda t, res, bs, dyn=False: prompt_util.PromptSettings(target=t, positive=t, unconditional="", neutral="", action="erase",
                                                       guidance_scale=1.0, resolution=res, batch_size=bs, dynamic_resolution=dyn)
    # two shape classes (they differ in the prompt batch: the 64-px latents keep the emulated passes cheap) with two prompts
    # each, plus a dynamic-resolution prompt whose bucket is drawn from the generator all ranks share (128: the draw happens,
    # the only bucket is 64 x 64 -- larger buckets cost minutes on the emulator; tests/test_fullsize.py runs real ones on the GPU)
    prompts = [mk("van gogh", 64, 1), mk("monet", 64, 1), mk("picasso", 64, 2), mk("dali", 64, 2), mk("klimt", 128, 1, True)]
    with contextlib.redirect_stdout(io.StringIO()):
        T.train(cfg, prompts, device=torch.device("cpu"), use_graphs=False, progress=False)
    q.put((rank, seen))
    T.shutdown_distributed()            # what the train CLIs / bench.py end with: barrier + destroy, on every rank
    assert not dist.is_initialized()
    T.shutdown_distributed()            # and it is a no-op without a process group


def test_train_under_dp_runs_the_same_shape_class_on_every_rank(tmp_path):
    """VERDICT r3 / SURVEY 5.8: with mixed-resolution prompts every rank must enter the same (batch, h, w) bucket in a step
    (the launch plans, hence the step time, depend on it); WHICH prompt of the class a rank trains on is its own draw."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_shape_worker, args=(r, 2, port, q, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        r, seen = q.get(timeout=1500)
        res[r] = seen
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert len(res[0]) == len(res[1]) == 4
    assert [s for s, _ in res[0]] == [s for s, _ in res[1]], "ranks ran different (batch, h, w) shapes in the same step"
    assert len({s for s, _ in res[0]}) >= 2, "the schedule should visit more than one shape class"


_DET_PROBE = """
import hashlib, os, sys
import torch
sys.path.insert(0, %r)
import test_dist as D
D._setup()
fs, net, pair = D._make(1)
fs.step(pair, 2, torch.randn(1, 4, 32, 32, generator=torch.Generator().manual_seed(100)))
h = lambda t: hashlib.sha1(t.detach()[:net.numel].contiguous().numpy().tobytes()).hexdigest()
print("DET", h(net.grad), h(net.slab))
"""


def test_deterministic_mode_is_independent_of_workgroup_and_wave_order():
    """LECO_DETERMINISTIC=1 (+ LECO_GN_FUSED=0) promises a bitwise reproducible step: no result may depend on the order
    in which workgroups or waves happen to run.  One optimizer step of the tiny model on a 32 x 32 latent (large enough
    that the DEFAULT mode's fp32 atomics make the slabs differ between exactly these two runs) in separate interpreters --
    ascending workgroups / lockstep waves on 8 emulator threads vs a scrambled workgroup order, random wave schedule and
    3 threads -- must leave bit-identical gradient and parameter slabs."""
    import subprocess
    base = dict(os.environ, LECO_DETERMINISTIC="1", LECO_GN_FUSED="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        base.pop(k, None)

    def run(**env):
        r = subprocess.run([sys.executable, "-c", _DET_PROBE % os.path.join(ROOT, "tests")], cwd=ROOT, env=dict(base, **env),
                           capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-3000:]
        return [l for l in r.stdout.splitlines() if l.startswith("DET")][0]

    a = run(LECO_EMU_THREADS="8")
    b = run(LECO_EMU_THREADS="3", LECO_EMU_BLOCKS="random", LECO_EMU_SCHED="random")
    assert a == b, (a, b)
