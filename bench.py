#!/usr/bin/env python
"""bench.py -- LECO training steps/sec on MI355X (BASELINE.json metric).

Workload (config.workload): BASELINE.json configs[1] -- SDv1.5 UNet (random-init weights of the
exact architecture, synthetic prompt embeddings), LoRA rank 4 / alpha 1 "lierla" (192 modules),
512x512 (latent 64x64), prompt batch_size 2 (UNet batch 4 with the classifier-free-guidance
duplication), bf16 MFMA / fp32 accumulate, DDIM, max_denoising_steps 50.  One "step" is one full
reference-faithful optimizer step (train_lora.py:141-290): k guided denoising passes with LoRA on,
three LoRA-off passes, one LoRA-on pass, ESD loss, backward, (all-reduce,) AdamW.  k follows the
seeded sequence `torch.Generator().manual_seed(0); randint(1, 50)` (mean 25), the same on all ranks.

N > 1: one process per GPU (torchrun env), weak scaling -- every rank runs its own step on its own
noise; the only collective is the all-reduce of the flat LoRA gradient slab.  `value` is the
whole-job rate: world_size * K / max-over-ranks wall time.
"""
import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

F_FWD_SD15_512 = 0.8033e12   # FLOPs of one sample-pass
ATTN_SHARE = 0.157
PEAK_BF16 = 2.5e15           # dense bf16 MFMA, MI355X_MICROARCH.md


def step_flops(bs: int, k: int) -> float:
    """Reference-faithful algorithmic work of one step: W_ref(k) = 2 bs F_fwd (k + 4 + 1 + a)."""
    return 2 * bs * F_FWD_SD15_512 * (k + 5 + ATTN_SHARE)


def cpu_baseline(k_mean: float, bs: int, ks=(1, 2), budget_s: float = 150.0):
    """The reference loop on the host CPU (BASELINE.json configs[0]: SD1.5, rank 4, 512^2, prompt batch 1, fp32,
    DDIM, AdamW), as a port: `oracle/step_ref.leco_step` restates one iteration of train_lora.py:141-281 on the
    oracle UNet / DDIM / LoRA (the reference's own files need `diffusers` and do not exist on the GPU box), followed
    by `loss.backward()`, `optimizer.step()`, `lr_scheduler.step()` and the reference's per-step `flush()`
    (train_lora.py:279-290).  FULL optimizer steps are timed, one per entry of `ks` (k denoising passes each); a step
    costs a + b k, so two steps with different k give both coefficients, which are then evaluated at the k mean of
    the GPU run and scaled by the prompt batch (2 x the samples -> 2 x the time; a CPU has no idle lanes to fill)."""
    import gc
    from oracle import lora_ref, step_ref
    from oracle import unet_ref as R
    from oracle.ddim_ref import DDIMSchedulerRef
    torch.manual_seed(1234)
    unet = R.init_synthetic_(R.UNet2DConditionModel(R.sd15_config()), seed=1234)
    unet.requires_grad_(False)
    unet.eval()
    net = lora_ref.LoRANetworkRef(unet, rank=4, multiplier=1.0, alpha=1.0)
    g = torch.Generator().manual_seed(99)
    with torch.no_grad():
        for l in net.unet_loras:
            l.lora_up.weight.copy_(torch.randn(l.lora_up.weight.shape, generator=g) * 0.02)
    params = [p for l in net.unet_loras for p in (l.lora_down.weight, l.lora_up.weight)]
    opt = torch.optim.AdamW(params, lr=1e-4)
    lrs = torch.optim.lr_scheduler.ConstantLR(opt, factor=1, total_iters=1000)
    sched = DDIMSchedulerRef()
    eg = torch.Generator().manual_seed(4321)
    emb = {n: torch.randn(1, 77, 768, generator=eg) for n in ("target", "neutral")}
    emb["positive"], emb["unconditional"] = emb["target"], emb["neutral"]       # 'van gogh' erase: 2 distinct prompts
    times, losses = [], []
    t_all = time.perf_counter()
    for i, k in enumerate(ks):
        if times and time.perf_counter() - t_all + times[-1] * 1.3 > budget_s:
            break
        lat = torch.randn(1, 4, 64, 64, generator=torch.Generator().manual_seed(1000 + i))
        t0 = time.perf_counter()
        out = step_ref.leco_step(unet, net, sched, emb, lat, k, 50, guidance_scale=1.0, action="erase", batch_size=1)
        opt.zero_grad()
        out["loss"].backward()
        opt.step()
        lrs.step()
        del out
        gc.collect()
        times.append(time.perf_counter() - t0)
        losses.append(None)
    if len(times) >= 2 and ks[1] != ks[0]:
        b = (times[1] - times[0]) / (ks[1] - ks[0])
        a = times[0] - b * ks[0]
        how = f"a + b k with a = {a:.1f} s, b = {b:.1f} s from the two steps"
    else:   # one step only: W_ref(k) = 2 bs F_fwd (k + 5 + a_attn)
        b = times[0] / (ks[0] + 5 + ATTN_SHARE)
        a = b * (5 + ATTN_SHARE)
        how = f"W_ref(k): {b:.1f} s per forward-equivalent from the one step"
    t_step = bs * (a + b * k_mean)
    return {"value": 1.0 / t_step, "unit": "steps/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{len(times)} full fp32 optimizer steps of the ported reference loop at prompt batch 1 with k = "
                      f"{list(ks[:len(times)])}: {', '.join(f'{t:.1f} s' for t in times)}; evaluated at k = {k_mean:.1f} "
                      f"and prompt batch {bs} ({how})",
            "steps_timed": len(times), "k": list(ks[:len(times)]), "step_seconds": times,
            "host_cpus": os.cpu_count(), "threads": torch.get_num_threads()}


def dominant_kernel_roofline(dev):
    """The kernel with the largest share of the step (profiles/: the 3x3-conv implicit GEMM) timed live with HIP
    events on one of its heaviest launches: ResnetBlock conv 640->640 at 32x32, UNet batch 4 (M=4096, N=640,
    K=5760; algorithmic work 2*M*N*K)."""
    from leco_amd import hip, ops
    B, H, C = 4, 32, 640
    M, N, K = B * H * H, C, 9 * C
    x = torch.randn(M, C, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    ws = torch.empty(32 * 1024 * 1024, device=dev)
    g = hip.gemm_args(x, w, out, m=M, n=N, k=K, a_mode=hip.A_CONV3_S1, conv=(B, H, H, H, H), lda=C)
    stream = ops.default_stream()
    for _ in range(10):
        hip.gemm(g, stream, 0, 0, ws)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    iters = 50
    e0.record()
    for _ in range(iters):
        hip.gemm(g, stream, 0, 0, ws)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    tf = 2.0 * M * N * K / us / 1e6
    return {"name": "gemm_kernel<256,128,conv3x3> (+ split-K finish)", "shape": f"M={M} N={N} K={K}",
            "us_per_launch": us, "achieved": tf, "peak": PEAK_BF16 / 1e12, "unit": "TFLOP/s", "frac": tf / (PEAK_BF16 / 1e12),
            "algorithmic_bytes": 2 * (M * C + N * K + M * N), "traffic": pmc_fetch_bytes_per_launch("gemm_kernel<256, 128, true")}


def pmc_fetch_bytes_per_launch(kernel_prefix: str):
    """HBM-side read bytes per launch of one kernel from the committed counter pass (profiles/r01_pmc_fetch_step.csv:
    rocprofv3 --pmc FETCH_SIZE in its own run, summed per kernel by tools/pmc_summary.py).  FETCH_SIZE is in KB and
    reports half of the bytes of wide coalesced reads on gfx950 (MI355X_MICROARCH.md, HBM section): x 1024 x 2.
    Averaged over every launch of that kernel in the profiled step (all conv shapes), not only the timed shape.
    None when the summary is absent."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_fetch_step.csv")
    try:
        for line in open(path):
            if line.startswith(kernel_prefix):
                _, calls, kb = line.rsplit(",", 2)
                return float(kb) * 1024.0 * 2.0 / float(calls)
    except (OSError, ValueError):
        pass
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-graphs", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--arch", default="sd15")
    ap.add_argument("--bs", type=int, default=2)
    ap.add_argument("--res", type=int, default=512)
    ap.add_argument("--rank", type=int, default=4)
    ap.add_argument("--k", type=int, default=0, help="profiling only: fixed number of denoising passes per step "
                                                     "(0 = the seeded reference distribution; the headline number uses 0)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: re-launch this script as N ranks (one process per GPU) under torchrun
        import socket
        import subprocess
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))

    from leco_amd import model_util, prompt_util, train_util
    from leco_amd.lora import LoRANetwork
    from leco_amd.train import FusedStep, init_distributed

    rank, world, local = init_distributed()
    assert world == args.gpus, f"WORLD_SIZE={world} but --gpus {args.gpus}"
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    dev = torch.device(f"cuda:{local}")
    torch.cuda.set_device(dev)

    import io
    import contextlib
    tokenizer, text_encoder, unet, sched = model_util.load_models(f"synthetic:{args.arch}", "ddim")
    unet.to(dev, dtype=torch.bfloat16)
    unet.requires_grad_(False)
    unet.eval()
    unet.use_graphs = not args.no_graphs
    torch.manual_seed(1234)
    with contextlib.redirect_stdout(io.StringIO()):
        net = LoRANetwork(unet, rank=args.rank, multiplier=1.0, alpha=1.0).to(dev)
    # lora_up starts at zero in the reference; give it a small value so the LoRA path does real work
    g = torch.Generator().manual_seed(99)
    with torch.no_grad():
        for l in net.unet_loras:
            l.lora_up.weight.copy_((torch.randn(l.lora_up.weight.shape, generator=g) * 0.02).to(dev))
    net.mark_updated()
    settings = prompt_util.PromptSettings(target="van gogh", positive="van gogh", unconditional="", neutral="",
                                          action="erase", guidance_scale=1.0, resolution=args.res, batch_size=args.bs)
    emb = {p: text_encoder([p])[0] for p in ("van gogh", "")}
    pair = prompt_util.PromptEmbedsPair(torch.nn.MSELoss(), emb["van gogh"], emb["van gogh"], emb[""], emb[""], settings)
    fused = FusedStep(unet, net, sched, 50, lr=1e-4, world_size=world)

    kgen = torch.Generator().manual_seed(0)
    ks = [torch.randint(1, 50, (1,), generator=kgen).item() for _ in range(args.warmup + args.steps)]
    if args.k > 0:
        ks = [args.k] * len(ks)
    noise_gen = torch.Generator().manual_seed(1000 + rank)

    def one(i):
        lat = train_util.get_initial_latents(sched, args.bs, args.res, args.res, 1, generator=noise_gen)
        return fused.step(pair, ks[i], lat)

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier(device_ids=[local])
        torch.cuda.synchronize()

    for i in range(args.warmup):
        one(i)
    barrier()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    losses = []
    for i in range(args.warmup, args.warmup + args.steps):
        losses.append(one(i).clone())       # device-side copy of the loss scalar: no host sync in the timed loop
    e1.record()
    barrier()
    dt = time.perf_counter() - t0
    dt_ev = e0.elapsed_time(e1) * 1e-3
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([dt], device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    losses = [float(l.item()) for l in losses]
    finite = all(math.isfinite(l) for l in losses)
    if rank != 0:
        if not finite:
            sys.exit(f"rank {rank}: non-finite loss in the timed steps: {losses}")
        return
    timed_ks = ks[args.warmup:]
    flops = sum(step_flops(args.bs, k) for k in timed_ks)
    achieved = flops / dt_ev / 1e12
    out = {
        "metric": "LECO train-steps/sec, SDv1.5 rank-4 512px bs=2", "value": world * args.steps / dt, "unit": "steps/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": f"SDv1.5 UNet (random init) LECO erase step, LoRA rank {args.rank} lierla, "
                               f"{args.res}x{args.res}, prompt batch {args.bs} (UNet batch {2 * args.bs}), DDIM 50, "
                               f"reference-faithful pass structure (k+3+1 fwd, 1 bwd)",
                   "global_batch": args.bs * world, "k_sequence_seed": 0 if args.k <= 0 else f"fixed k={args.k} (profiling run)", "k_mean": sum(timed_ks) / len(timed_ks),
                   "hip_graphs": bool(unet.use_graphs), "parallelism": f"dp{world}", "loss": losses[-1], "losses": losses},
        "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_BF16 / 1e12, "unit": "TFLOP/s",
                     "frac": achieved / (PEAK_BF16 / 1e12), "traffic": None,
                     "note": "algorithmic FLOPs W_ref(k)=2*bs*F_fwd*(k+5+a) summed over the timed steps / HIP-event "
                             "time on the compute stream (rank 0); per-kernel breakdown in profiles/"},
    }
    try:
        out["roofline"]["dominant_kernel"] = dominant_kernel_roofline(dev)
    except Exception as e:  # never hide the step number
        out["roofline"]["dominant_kernel"] = {"error": repr(e)}
    if world == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(sum(timed_ks) / len(timed_ks), args.bs)
        except Exception as e:  # the baseline leg must never hide the GPU number
            out["cpu_baseline"] = {"value": None, "error": repr(e)}
    print(json.dumps(out))
    if not finite:   # a timing of a broken computation is not a result
        sys.exit(f"non-finite loss in the timed steps: {losses}")


if __name__ == "__main__":
    main()
