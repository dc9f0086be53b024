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

# FLOPs of one sample-pass F_fwd and attention share a of the backward, at the resolution named
F_FWD = {("sd15", 512): (0.8033e12, 0.157), ("sd21", 768): (2.149e12, 0.292), ("sdxl", 1024): (6.761e12, 0.116)}
F_FWD_SD15_512, ATTN_SHARE = F_FWD[("sd15", 512)]
PEAK_BF16 = 2.5e15           # dense bf16 MFMA, MI355X_MICROARCH.md


def step_flops(bs: int, k: int, arch: str = "sd15", res: int = 512) -> float:
    """Reference-faithful algorithmic work of one step: W_ref(k) = 2 bs F_fwd (k + 4 + 1 + a)."""
    f, a = F_FWD.get((arch, res), (F_FWD_SD15_512 * (res / 512.0) ** 2, ATTN_SHARE))
    return 2 * bs * f * (k + 5 + a)


def step_flops_dedup(bs: int, k: int, U: int, arch: str = "sd15", res: int = 512) -> float:
    """SURVEY 8(d) W_min: the de-duplicated step executes bs F_fwd (2 k + U + 1 + (1 + a)) -- the guidance-1 passes on the
    conditional samples only, U distinct frozen prompts, a backward of batch bs."""
    f_fwd, a = F_FWD.get((arch, res), (F_FWD_SD15_512 * (res / 512.0) ** 2, ATTN_SHARE))
    return bs * f_fwd * (2 * k + U + 1 + (1.0 + a))


def cpu_baseline(k_mean: float, bs: int, ks=(1, 2), budget_s: float = 900.0):
    """The reference loop on the host CPU (BASELINE.json configs[0]: SD1.5, rank 4, 512^2, prompt batch 1, fp32,
    DDIM, AdamW), as a port: `oracle/step_ref.leco_step` restates one iteration of train_lora.py:141-281 on the
    oracle UNet / DDIM / LoRA (the reference's own files need `diffusers` and do not exist on the GPU box), followed
    by `loss.backward()`, `optimizer.step()`, `lr_scheduler.step()` and the reference's per-step `flush()`
    (train_lora.py:279-290).  FULL optimizer steps are timed AT THE GPU RUN'S PROMPT BATCH (`bs` latents per step: nothing
    about the batch is assumed), one per entry of `ks` (k denoising passes each) after an untimed warm-up forward; the
    steady-state step is scaled with W_ref(k) to the k mean of the GPU run.  The reference's OWN files driving the same
    oracle UNet cost the same (oracle/time_reference_cpu.py, build container: 64 s / 50 s for k = 1 / 2 at prompt batch 1)."""
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
    with torch.no_grad():      # untimed warm-up pass (thread pool, allocator): the first forward of a process is ~1.6x slow
        unet(torch.zeros(2 * bs, 4, 64, 64), torch.tensor(1), encoder_hidden_states=torch.zeros(2 * bs, 77, 768))
    times, losses = [], []
    t_all = time.perf_counter()
    for i, k in enumerate(ks):
        # TWO full steps always (the first still pays allocator / autograd warm-up: 135.6 s vs 122.2 s at prompt
        # batch 2 on the GPU box, profiles/r04_bench.json -- about 4.5 minutes of CPU work there); further ones only inside the
        # budget
        if len(times) >= 2 and time.perf_counter() - t_all + times[-1] * 1.3 > budget_s:
            break
        lat = torch.randn(bs, 4, 64, 64, generator=torch.Generator().manual_seed(1000 + i))
        t0 = time.perf_counter()
        out = step_ref.leco_step(unet, net, sched, emb, lat, k, 50, guidance_scale=1.0, action="erase", batch_size=bs)
        opt.zero_grad()
        out["loss"].backward()
        opt.step()
        lrs.step()
        del out
        gc.collect()
        times.append(time.perf_counter() - t0)
        losses.append(None)
    # the LAST timed step is the steady-state one when there are two (the first still pays allocator / autograd warm-up: measured
    # 64 s vs 50 s on the build container, 135.6 s vs 122.2 s on the GPU box at prompt batch 2 -- a single-step sample
    # under-states the CPU by ~10 %, said in `sample`); its cost per forward-equivalent, W_ref(k) = 2 bs F_fwd (k + 5 + a_attn), is evaluated at
    # the GPU run's k mean
    b = times[-1] / (ks[len(times) - 1] + 5 + ATTN_SHARE)
    a = b * (5 + ATTN_SHARE)
    how = f"W_ref(k): {b:.1f} s per forward-equivalent from the last step"
    t_step = a + b * k_mean
    return {"value": 1.0 / t_step, "unit": "steps/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{len(times)} full fp32 optimizer steps of the ported reference loop at prompt batch {bs} (the GPU run's: "
                      f"measured, not scaled from batch 1) with k = {list(ks[:len(times)])}: "
                      f"{', '.join(f'{t:.1f} s' for t in times)}; the LAST one, evaluated at k = {k_mean:.1f} ({how})",
            "prompt_batch": bs,
            "steps_timed": len(times), "k": list(ks[:len(times)]), "step_seconds": times,
            "host_cpus": os.cpu_count(), "threads": torch.get_num_threads()}


# ---------------------------------------------------------------------------------------------------------------
# dominant kernel: found LIVE (every distinct launch of a step timed with HIP events on the compute stream, weighted by how
# often the step issues it), cross-checked against the top row of the committed rocprofv3 kernel trace
def _latest(suffix):
    """profiles/rNN_<suffix> of the newest round that has one."""
    import glob
    hits = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_" + suffix)))
    return hits[-1] if hits else os.path.join(ROOT, "profiles", "r00_" + suffix)


PROFILE_STATS = _latest("step_kernel_stats.txt")
PMC_FETCH = _latest("pmc_dominant_fetch.csv")
PMC_MFMA = _latest("pmc_dominant_mfma.csv")


def kernel_sources_hash():
    """sha1 over the kernel sources (leco_amd/csrc): counter summaries carry it in their header (tools/pmc_summary.py),
    and numbers from a summary whose hash differs from the tree's are NOT printed -- they describe other kernels."""
    import hashlib
    h = hashlib.sha1()
    csrc = os.path.join(ROOT, "leco_amd", "csrc")
    for dp, _, files in sorted(os.walk(csrc)):
        if os.path.basename(dp) == "_obj":
            continue
        for f in sorted(files):
            if f.endswith((".hip", ".h", ".cpp")):
                with open(os.path.join(dp, f), "rb") as fh:
                    h.update(f.encode() + b"\0" + fh.read())
    return h.hexdigest()[:12]


def _summary_hash(path):
    try:
        with open(path) as fh:
            for line in fh:
                if not line.startswith("#"):
                    break
                if "csrc_sha1=" in line:
                    return line.split("csrc_sha1=")[1].split()[0]
    except OSError:
        pass
    return None


def _launch_identity(op):
    """-> (kernel names as rocprofv3 prints them, shape key, algorithmic flops) of one plan launch."""
    from leco_amd import hip
    a = op.args
    if op.name == "leco_gemm_ex":
        g = op.keep[0]
        parts = hip.gemm_describe(g, a[1], a[2], a[3], a[4]).split(" ; ")
        names = [p.split(" grid=")[0] for p in parts]
        ext = g.ext_k if (g.a_ext or g.t_w) else 0
        key = (op.name, g.m, g.n, g.k, g.a_mode, ext, bool(g.t_w), bool(g.residual), g.act, g.batch, g.h_in, g.h_out, a[1], a[2])
        kin = g.k // 9 if g.a_mode else g.k
        rows_in = g.m if g.a_mode == 0 else g.batch * g.h_in * g.w_in
        # algorithmic operand bytes: every input / weight / output element once (bf16), + the residual read
        by = 2.0 * (rows_in * kin + g.n * g.k + g.m * g.n * (0.5 if g.act == 2 else 1.0) + (g.m * g.n if g.residual else 0))
        return names, key, 2.0 * g.m * g.n * (g.k + ext), by
    if op.name == "leco_attention_fwd":
        B, H, sq, skv, d = a[13], a[14], a[15], a[16], a[17]
        # the dispatcher of csrc/attention.hip::launch_fwd, restated: LDS-DMA staged kernel for unmasked d = 40 / 64 / 80
        # problems with >= 512 workgroups, else the register-staged one
        wgs2, wgs1 = -(-sq // 128) * H * B, -(-sq // 64) * H * B
        dma = int(os.environ.get("LECO_ATTN_DMA", "1") or 1)
        name = f"attn_fwd_kernel<{d}, {2 if wgs2 >= 1024 else 1}, {'true' if skv % 64 else 'false'}>"
        if d in (40, 64, 80) and dma and skv % 64 == 0:
            nbuf = 2 if d == 64 else 3
            if sq % 128 == 0 and (wgs2 >= 512 or dma == 2):
                name = f"attn_fwd_dma_kernel<{d}, 2, {nbuf}>"
            elif sq % 64 == 0 and (wgs1 >= 512 or dma == 2):
                name = f"attn_fwd_dma_kernel<{d}, 1, {nbuf}>"
        return ([name], (op.name, B, H, sq, skv, d), 4.0 * B * H * sq * skv * d, 2.0 * B * H * d * (2 * sq + 2 * skv))
    if op.name == "leco_xblock_tail":
        A = op.keep[0]
        po = 1 if A.proj_out.w else 0
        lora = 1 if A.to_out1.dn else 0
        return ([f"xblock_tail_kernel<{A.c}, {A.c // A.heads}>"], (op.name, A.m, A.c, A.heads, po, lora),
                A.m * (2.0 * A.c * A.c * (15 + po) + 4.0 * A.skv * A.c),
                2.0 * (A.m * A.c * (3 + po) + A.c * A.c * (15 + po)))
    if op.name == "leco_xblock_head":
        A = op.keep[0]
        return ([f"xblock_head_kernel<{A.c}>"], (op.name, A.m, A.c, 1 if A.gn_cstats else 0, 1 if A.proj_in.dn else 0),
                A.m * 2.0 * A.c * A.c * 4, 2.0 * (A.m * A.c * 5 + 4 * A.c * A.c))
    if op.name == "leco_xgemm":
        A = op.keep[0]
        ext = 32 if A.lin.dn else 0
        bm = 32 if A.k == 1280 else 64
        return ([f"xgemm_kernel<{bm}, {A.k}, {os.environ.get('LECO_XGEMM_VAR', '1')}>"], (op.name, A.m, A.n, A.k, ext, 1 if A.residual else 0),
                2.0 * A.m * A.n * (A.k + ext), 2.0 * (A.m * A.k + A.n * A.k + A.m * A.n * (2 if A.residual else 1)))
    if op.name in ("leco_groupnorm_fwd", "leco_groupnorm_apply_stats"):
        B, hw, c = (a[7], a[8], a[9]) if op.name == "leco_groupnorm_fwd" else (a[10], a[11], a[12])
        return [op.name], (op.name, B, hw, c), 0.0, 4.0 * B * hw * c
    if op.name == "leco_layernorm_fwd":
        return [op.name], (op.name, a[5], a[6]), 0.0, 4.0 * a[5] * a[6]
    if op.name == "leco_attention_bwd":
        B, H, sq, skv, d = a[26], a[27], a[28], a[29], a[30]
        return ["attention_bwd(3 kernels)"], (op.name, B, H, sq, skv, d), 10.0 * B * H * sq * skv * d, 2.0 * B * H * d * (4 * sq + 4 * skv)
    return [op.name], (op.name,) + tuple(x for x in a if isinstance(x, int) and 0 <= x < (1 << 20)), 0.0, 0.0


def step_launches(st, k_mean):
    """[(op, launches per step)] of one reference-faithful step with k_mean denoising passes."""
    out = []
    skip = ("leco_advance", "leco_cfg_ddim_step", "leco_cfg_sched_step")      # step state; negligible time
    for plan, which, w in ((st["dplan"], "ctx_on", 1.0), (st["dplan"], "denoise", float(k_mean)), (st["fplan"], "fwd_off", 1.0),
                           (st["plan"], "fwd_on", 1.0), (st["plan"], "bwd", 1.0)):
        out += [(op, w, which) for op in plan.lists[which] if op.name not in skip]
    return out


def _profile_top_row():
    try:
        for line in open(PROFILE_STATS):
            f = line.split(None, 6)
            if len(f) == 7 and f[0].replace(".", "").isdigit():
                return {"name": f[6].strip(), "share_pct": float(f[0]), "avg_us": float(f[3])}
    except OSError:
        pass
    return None


def _pmc_row(path, name):
    try:
        lines = [l for l in open(path).read().splitlines() if not l.startswith("#")]
        hdr = lines[0].split(",")
        for line in lines[1:]:
            if line.startswith(name + ","):
                vals = line[len(name) + 1:].split(",")
                return dict(zip(hdr[1:], [float(v) for v in vals]))
    except (OSError, IndexError, ValueError):
        pass
    return None


def _time_launch_us(op, reps=8):
    """One plan launch in isolation: HIP events on the compute stream around back-to-back repeats (L2-warm)."""
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(2):
        op.run()
    e0.record()
    for _ in range(reps):
        op.run()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def _graph_launch_us(op, copies=16, reps=20):
    """One plan launch replayed from a hipGraph holding `copies` of it (no eager launch floor: what the step's graphs pay;
    the ~1.5 us node-to-node floor of a dependent chain is part of the figure)."""
    import ctypes as C
    from leco_amd import hip, ops
    from leco_amd.unet import _graph_api
    lib = _graph_api()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    hip.check(lib.leco_graph_begin_capture(side.cuda_stream), "begin")
    try:
        ops.run_plan([op] * copies, side.cuda_stream)
    finally:
        g = C.c_void_p()
        hip.check(lib.leco_graph_end_capture(side.cuda_stream, C.byref(g)), "end")
    cur = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        lib.leco_graph_launch(g, cur)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        lib.leco_graph_launch(g, cur)
    e1.record()
    e1.synchronize()
    lib.leco_graph_destroy(g)
    return e0.elapsed_time(e1) / reps / copies * 1e3


HBM_PEAK = 8.0e12            # HBM3E, MI355X_MICROARCH.md


def hbm_bound_launches(groups, top_n=6):
    """The GroupNorm / LayerNorm launches of a step (the HBM-bound kernels of the path, SURVEY 8d: read + write 2 B per
    element each): the `top_n` shapes by time per step, each timed from a hipGraph, GB/s against the 8 TB/s roof."""
    cand = [(key, op, w, by) for key, (op, names, fl, w, by) in groups.items()
            if key[0] in ("leco_groupnorm_fwd", "leco_groupnorm_apply_stats", "leco_layernorm_fwd") and by > 0]
    rows = []
    for key, op, w, by in cand:
        us = _graph_launch_us(op)
        rows.append({"op": key[0][5:], "shape": list(key[1:]), "launches_per_step": w, "us_per_launch": us,
                     "algorithmic_bytes": by, "achieved_GBps": by / us / 1e3, "frac": by / us / 1e3 / (HBM_PEAK / 1e9)})
    rows.sort(key=lambda r: -r["launches_per_step"] * r["us_per_launch"])
    tot = sum(r["launches_per_step"] * r["us_per_launch"] for r in rows)
    return {"peak_GBps": HBM_PEAK / 1e9, "us_per_step_all_norm_launches": tot, "top": rows[:top_n],
            "note": "us from a hipGraph of 16 copies of the launch (includes the ~1.5 us node-to-node floor); bytes = one bf16 read + "
                    "one bf16 write of the tensor; these launches are latency-, not bandwidth-bound at UNet batch 4 (DESIGN 8.000)"}


def dominant_kernel_roofline(st, k_mean, only_replay=False, dump_shapes=None, headline_workload=True):
    """Times every distinct launch of a step in isolation (HIP events on the compute stream, L2-warm repeats), attributes
    each to its kernel instantiation, picks the instantiation with the largest share of the step and reports its
    launch-weighted average duration and algorithmic FLOPs per launch.  MFMA-busy and HBM traffic per launch come from the
    committed counter passes over EXACTLY these launches (`rocprofv3 --pmc ... -- python bench.py --dominant-only`)."""
    launches = step_launches(st, k_mean)
    groups = {}                                   # shape key -> [op, names, flops, launches per step]
    per_list = {}                                 # list name -> {shape key: launches per run of that list}
    for op, w, which in launches:
        names, key, fl, by = _launch_identity(op)
        g = groups.setdefault(key, [op, names, fl, 0.0, by])
        g[3] += w
        d = per_list.setdefault(which, {})
        d[key] = d.get(key, 0) + 1
    top = _profile_top_row()
    if only_replay:
        # counter pass: replay the dominant instantiation's launches with their per-step multiplicities, nothing else
        want = top["name"] if top else None
        n = 0
        for op, names, fl, w, _by in groups.values():
            if want is not None and names == [want]:
                for _ in range(max(1, int(round(w)))):
                    op.run()
                    n += 1
        torch.cuda.synchronize()
        return {"replayed": n, "kernel": want}
    per_name = {}
    key_us = {}
    for key, (op, names, fl, w, by) in groups.items():
        us = key_us[key] = _time_launch_us(op)
        name = names[0] if len(names) == 1 else " + ".join(names)
        a = per_name.setdefault(name, [0.0, 0.0, 0.0, {}, 0.0])
        a[0] += w            # launches per step
        a[1] += w * us       # us per step
        a[2] += w * fl       # flops per step
        a[3][key] = (w, us, fl)
        a[4] += w * by       # algorithmic operand bytes per step
    total_us = sum(a[1] for a in per_name.values())
    # the committed trace and counter passes are of the HEADLINE workload's command: another configuration (--arch / --res / --bs /
    # --rank / --c3lier / --v-pred) ranks its kernels live and carries no trace or counter figures
    cur = kernel_sources_hash()
    ref = cur if headline_workload else None          # what a quoted trace / counter summary must have been taken on
    ranked = sorted(per_name.items(), key=lambda kv: -kv[1][1])
    live_name = ranked[0][0]
    # The dominant kernel is the top row of the committed kernel trace of this very command WHEN that trace was taken on
    # the kernel sources that are running (hash in its header) -- the trace, the counter passes (`--dominant-only` replays
    # that row's launches) and this line then describe the same kernel.  Two instantiations within a percent of each other
    # can swap places between the trace (launches in step order) and the isolated live timing; without a current trace the
    # live ranking decides.
    name = top["name"] if (top and top["name"] in per_name and _summary_hash(PROFILE_STATS) == ref) else live_name

    trace_current = bool(top and top["name"] in per_name and _summary_hash(PROFILE_STATS) == ref)

    def describe_kernel(nm):
        n, us, fl, shapes, by = per_name[nm]
        achieved = fl / us / 1e6 if us else 0.0       # TFLOP/s, from the isolated (L2-warm, back-to-back) live timing
        heavy = max(shapes.items(), key=lambda kv: kv[1][0] * kv[1][1])
        return {"name": nm, "launches_per_step": n, "us_per_launch": us / n, "flops_per_launch": fl / n,
                "algorithmic_bytes_per_launch": by / n,
                "share_of_step_kernel_time": us / total_us, "achieved": achieved, "peak": PEAK_BF16 / 1e12, "unit": "TFLOP/s",
                "frac": achieved / (PEAK_BF16 / 1e12), "distinct_shapes": len(shapes),
                "heaviest_shape": {"key": [str(x) for x in heavy[0][1:9]], "launches_per_step": heavy[1][0], "us": heavy[1][1],
                                   "tflops": heavy[1][2] / heavy[1][1] / 1e6 if heavy[1][1] else 0.0}}
    out = describe_kernel(name)
    out.update({"profile_top_row": top if headline_workload else None, "live_top": live_name,
                "agrees_with_profile": bool(top and top["name"] == name) if headline_workload else None})
    # The headline fraction is measured LIVE in this run: algorithmic FLOPs per launch / the kernel's launch-weighted average
    # duration, HIP events on the compute stream around its launches (every distinct shape, weighted by how often a step
    # issues it).  The committed kernel trace of the same command (launches in step order) is the cross-check: when it was
    # taken on the running kernel sources its average duration rides along as `us_per_launch_trace` / `frac_trace` and the
    # two must agree (round-4 verdict: the trace figure alone describes the box the trace was taken on, not this run).
    out["timing_source"] = "live: HIP events around this run's own launches of the kernel (isolated, back-to-back repeats per shape)"
    if trace_current and top["avg_us"] > 0:
        out["us_per_launch_trace"] = top["avg_us"]
        out["achieved_trace"] = out["flops_per_launch"] / top["avg_us"] / 1e6
        out["frac_trace"] = out["achieved_trace"] / (PEAK_BF16 / 1e12)
        out["trace_source"] = f"avg duration of the row in {os.path.relpath(PROFILE_STATS, ROOT)} (rocprofv3 --kernel-trace of this command on kernel sources {cur})"
    # isolated GPU time of the step's launch lists: a step with k denoising passes keeps the GPU busy for at least
    # fixed + k x per_denoise_pass microseconds; (that sum) / (measured step time) is the step's busy fraction
    list_us = {which: sum(n * key_us[key] for key, n in d.items()) for which, d in per_list.items()}
    out["isolated_us"] = {"per_denoise_pass": list_us.get("denoise", 0.0),
                          "fixed": sum(v for k_, v in list_us.items() if k_ != "denoise"), "by_list": list_us,
                          "launches": {which: sum(d.values()) for which, d in per_list.items()}}
    try:
        out["hbm_bound"] = hbm_bound_launches(groups)
    except Exception as e:       # never hide the MFMA figures behind the secondary list
        out["hbm_bound"] = {"error": repr(e)}
    if dump_shapes:
        # every shape the dominant kernel runs in a step: the numerator of the fraction can be recomputed from this table
        n, us, fl, shapes, by = per_name[name]
        with open(dump_shapes, "w") as fh:
            fh.write(f"# dominant kernel {name} on kernel sources {cur}: launches per step (k mean {k_mean:.2f}), algorithmic GFLOP and "
                     f"isolated us per launch, by shape key {'(op, m, n, k, a_mode, ext_k, fusedT, residual, act, batch, h_in, h_out, tile, split)' if 'gemm' in name or 'conv' in name else ''}\n")
            for key, (w, u, f) in sorted(shapes.items(), key=lambda kv: -kv[1][0] * kv[1][2]):
                fh.write(f"{w:9.2f} launches  {f / 1e9:10.3f} GFLOP  {u:8.1f} us  {' '.join(str(x) for x in key)}\n")
            fh.write(f"# total: {n:.2f} launches per step, {fl / 1e9:.1f} GFLOP per step -> {fl / n / 1e9:.3f} GFLOP per launch (launch-weighted)\n")
    # the next instantiations by share of the step (live timing), each with its own fraction of the MFMA peak and -- from the
    # same counter files, which also hold the launches of the plan-building pass -- its MFMA-busy share
    out["next_kernels"] = []
    for nm, _ in ranked[:4]:
        if nm == name:
            continue
        d = describe_kernel(nm)
        row = _pmc_row(PMC_MFMA, nm) if _summary_hash(PMC_MFMA) == ref else None
        if row and row.get("GRBM_GUI_ACTIVE"):
            d["mfma_busy"] = row.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (row["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
        d.pop("heaviest_shape")
        out["next_kernels"].append(d)
    # counter passes are separate rocprofv3 runs (tools/gpu_round_run.sh); their summaries are only quoted when they were
    # taken on THESE kernel sources (hash in the header) and every number names the file it comes from
    out["kernel_sources"] = cur
    fetch = _pmc_row(PMC_FETCH, name) if _summary_hash(PMC_FETCH) == ref else None
    mfma = _pmc_row(PMC_MFMA, name) if _summary_hash(PMC_MFMA) == ref else None
    if fetch and fetch.get("calls"):
        # FETCH_SIZE is in KB and counts a wide coalesced read at half its bytes on gfx950 (MI355X_MICROARCH.md, HBM)
        out["traffic"] = fetch.get("FETCH_SIZE", 0.0) * 1024.0 * 2.0 / fetch["calls"]
        out["traffic_note"] = (f"HBM-side read bytes per launch (FETCH_SIZE KB x 1024 x 2) from {os.path.relpath(PMC_FETCH, ROOT)}: "
                               f"rocprofv3 --pmc FETCH_SIZE over `bench.py --dominant-only` on kernel sources {cur}")
    else:
        out["traffic"] = None
        out["traffic_note"] = (f"no counter pass on kernel sources {cur} ({os.path.relpath(PMC_FETCH, ROOT)} has "
                               f"{_summary_hash(PMC_FETCH)}): not quoted" if headline_workload else
                               "the committed counter passes are of the headline workload's command; none was taken for this configuration")
    if mfma and mfma.get("GRBM_GUI_ACTIVE"):
        out["mfma_busy"] = mfma.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (mfma["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
        out["mfma_busy_note"] = f"SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs) from {os.path.relpath(PMC_MFMA, ROOT)}"
    return out


def box_calibration(dev):
    """Three one-second measurements that characterise THIS box independently of the kernels under test (round 5: leases of
    one pool ran the identical step 16.5 % apart with nothing in clocks / power / throttle flags to show for it): a copy that
    streams HBM, a copy that stays in the memory-side cache, the vendor library's large bf16 GEMM.  Reported, never used."""
    out = {}
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def timed(fn, reps):
        for _ in range(2):
            fn()
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) * 1e-3 / reps
    try:
        a = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
        b = torch.empty_like(a)
        out["hbm_copy_gb_s"] = 2 * a.numel() / timed(lambda: b.copy_(a), 10) / 1e9          # read + write
        c, d = a[:32 << 20], b[:32 << 20]
        out["cache_resident_copy_gb_s"] = 2 * c.numel() / timed(lambda: d.copy_(c), 50) / 1e9
        del a, b, c, d
        x = torch.randn(8192, 8192, device=dev).to(torch.bfloat16)
        y = torch.randn(8192, 8192, device=dev).to(torch.bfloat16)
        out["vendor_gemm_8192_bf16_tflops"] = 2 * 8192 ** 3 / timed(lambda: torch.matmul(x, y), 10) / 1e12
        del x, y
        pr = torch.cuda.get_device_properties(dev)
        out["device"] = {"name": pr.name, "compute_units": pr.multi_processor_count, "total_memory_gb": round(pr.total_memory / 2 ** 30, 1),
                         "clock_rate_mhz": getattr(pr, "clock_rate", 0) / 1e3, "memory_clock_rate_mhz": getattr(pr, "memory_clock_rate", 0) / 1e3,
                         "l2_cache_mb": getattr(pr, "L2_cache_size", 0) / 2 ** 20, "gcn_arch": getattr(pr, "gcnArchName", "")}
        torch.cuda.empty_cache()
    except Exception as e:      # never in the way of the number
        out["error"] = repr(e)
    return out


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
    ap.add_argument("--c3lier", action="store_true", help="network.type c3lier (conv + time_emb_proj LoRA; BASELINE config 4)")
    ap.add_argument("--v-pred", action="store_true", help="v-prediction scheduler (BASELINE config 3)")
    ap.add_argument("--dominant-only", action="store_true",
                    help="counter passes: build the plans, replay only the dominant kernel's launches of one step "
                         "(rocprofv3 --pmc ... -- python bench.py --dominant-only), print nothing else")
    ap.add_argument("--no-dominant", action="store_true",
                    help="skip the per-launch isolation timing behind the roofline object (kernel traces of the benchmark "
                         "command are taken with it, so the trace holds the step's own launches only)")
    ap.add_argument("--dump-shapes", default=None, metavar="FILE",
                    help="write the per-shape (launches, GFLOP, us) table of the dominant kernel to FILE (profiles/rNN_dominant_shapes.txt)")
    ap.add_argument("--no-telemetry", action="store_true", help="no clock / power sampler thread beside the timed loop")
    ap.add_argument("--telemetry-hz", type=float, default=10.0)
    ap.add_argument("--no-dedup", action="store_true",
                    help="skip the second timed loop (the de-duplicated pass structure, reported beside the headline as `dedup`)")
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
    from leco_amd.train import FusedStep, init_distributed, shutdown_distributed

    # LECO_BENCH_EMU=1 (tests/test_dist.py only): the same launch / rendezvous / timing / JSON plumbing with the host
    # emulator of the kernel sources on CPU tensors and gloo -- a dry run of the multi-GPU path for boxes without GPUs.
    # It measures nothing (the emulator is test infrastructure) and says so in the output.
    emu = os.environ.get("LECO_BENCH_EMU") == "1"
    if emu:
        sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
        import build_emu
        from leco_amd import hip
        hip._use_library(build_emu.build())
    rank, world, local = init_distributed("gloo" if emu else None)
    assert world == args.gpus, f"WORLD_SIZE={world} but --gpus {args.gpus}"
    assert emu or torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    dev = torch.device("cpu") if emu else torch.device(f"cuda:{local}")
    if not emu:
        torch.cuda.set_device(dev)
    n_allreduce = [0]
    if world > 1:      # count the data-path collectives of the timed steps (must be exactly one per step)
        import torch.distributed as dist
        _all_reduce = dist.all_reduce

        def counted(*a, **k):
            n_allreduce[0] += 1
            return _all_reduce(*a, **k)
        dist.all_reduce = counted

    import io
    import contextlib
    xl = args.arch == "sdxl"
    if xl:
        tokenizers, text_encoders, unet, sched = model_util.load_models_xl("synthetic:sdxl", "ddim")
    else:
        tokenizer, text_encoder, unet, sched = model_util.load_models(f"synthetic:{args.arch}", "ddim", v_pred=args.v_pred)
    unet.to(dev, dtype=torch.bfloat16)
    unet.requires_grad_(False)
    unet.eval()
    unet.use_graphs = not args.no_graphs and not emu
    torch.manual_seed(1234)
    from leco_amd.lora import DEFAULT_TARGET_REPLACE, UNET_TARGET_REPLACE_MODULE_CONV
    targets = list(DEFAULT_TARGET_REPLACE) + (list(UNET_TARGET_REPLACE_MODULE_CONV) if args.c3lier else [])
    with contextlib.redirect_stdout(io.StringIO()):
        net = LoRANetwork(unet, rank=args.rank, multiplier=1.0, alpha=1.0, target_replace_modules=targets).to(dev)
    # lora_up starts at zero in the reference; give it a small value so the LoRA path does real work
    g = torch.Generator().manual_seed(99)
    with torch.no_grad():
        for l in net.unet_loras:
            l.lora_up.weight.copy_((torch.randn(l.lora_up.weight.shape, generator=g) * 0.02).to(dev))
    net.mark_updated()
    settings = prompt_util.PromptSettings(target="van gogh", positive="van gogh", unconditional="", neutral="",
                                          action="erase", guidance_scale=1.0, resolution=args.res, batch_size=args.bs)
    if xl:
        emb = {p: prompt_util.PromptEmbedsXL(*train_util.encode_prompts_xl(tokenizers, text_encoders, [p], num_images_per_prompt=1))
               for p in ("van gogh", "")}
        add_time_ids = train_util.get_add_time_ids(args.res, args.res)
    else:
        emb = {p: text_encoder([p])[0] for p in ("van gogh", "")}
        add_time_ids = None
    pair = prompt_util.PromptEmbedsPair(torch.nn.MSELoss(), emb["van gogh"], emb["van gogh"], emb[""], emb[""], settings)
    fused = FusedStep(unet, net, sched, 50, lr=1e-4, world_size=world)

    kgen = torch.Generator().manual_seed(0)
    ks = [torch.randint(1, 50, (1,), generator=kgen).item() for _ in range(args.warmup + args.steps)]
    if args.k > 0:
        ks = [args.k] * len(ks)
    noise_gen = torch.Generator().manual_seed(1000 + rank)

    def one(i):
        lat = train_util.get_initial_latents(sched, args.bs, args.res, args.res, 1, generator=noise_gen)
        return fused.step(pair, ks[i], lat, add_time_ids=add_time_ids)

    if args.dominant_only:
        unet.use_graphs = False
        fused.step(pair, 1, train_util.get_initial_latents(sched, args.bs, args.res, args.res, 1, generator=noise_gen),
                   add_time_ids=add_time_ids)
        torch.cuda.synchronize()
        st = fused._state[(args.bs, args.res // 8, args.res // 8)]
        k_mean = sum(ks[args.warmup:]) / max(1, len(ks[args.warmup:]))
        print(json.dumps(dominant_kernel_roofline(st, k_mean, only_replay=True)))
        return

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier(**({} if emu else {"device_ids": [local]}))
        if not emu:
            torch.cuda.synchronize()

    # clocks / power / throttle state before, during (sampler thread) and after the timed loop (tools/gpu_telemetry.py)
    tele = None
    if not emu and not args.no_telemetry and rank == 0:
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            from gpu_telemetry import Telemetry
            tele = Telemetry(local, getattr(torch.cuda.get_device_properties(dev), "pci_bus_id", None))
        except Exception as e:
            tele = None
            print(f"telemetry unavailable: {e!r}", file=sys.stderr)
    tele_idle = tele.snapshot() if tele else None
    box = box_calibration(dev) if (not emu and rank == 0 and not args.no_telemetry) else None
    for i in range(args.warmup):
        one(i)
    barrier()
    tele_before = tele.snapshot() if tele else None
    if tele:
        tele.start(args.telemetry_hz)
    if not emu:
        e0 = torch.cuda.Event(enable_timing=True)
        step_ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    t0 = time.perf_counter()
    if not emu:
        e0.record()
    n_allreduce[0] = 0
    losses, host_done = [], []
    for i in range(args.warmup, args.warmup + args.steps):
        losses.append(one(i).clone())       # device-side copy of the loss scalar: no host sync in the timed loop
        if not emu:
            step_ev[i - args.warmup].record()       # one HIP event per step, on the compute stream
        host_done.append(time.perf_counter() - t0)  # when the HOST had finished enqueueing this step
    collectives = n_allreduce[0]
    barrier()
    dt = time.perf_counter() - t0
    t1_host = t0 + dt
    if tele:
        tele_samples = tele.stop()
        tele_after = tele.snapshot()
    # ---- the same k sequence once more on the DE-DUPLICATED pass structure (FusedStep.dedup, what train() runs unless
    # --strict_reference): reported beside the headline, never as the headline -- the headline executes the reference's own
    # k + 3 + 1 CFG-doubled passes.  Two untimed steps first (plan build + graph capture).
    dedup_out = None
    if not args.no_dedup and not args.dominant_only:
        fused.dedup = True
        for i in range(min(2, args.warmup + args.steps)):
            one(i)
        barrier()
        td0 = time.perf_counter()
        dlosses = []
        for i in range(args.warmup, args.warmup + args.steps):
            dlosses.append(one(i).clone())
        barrier()
        dtd = time.perf_counter() - td0
        fused.dedup = False
        if world > 1:
            import torch.distributed as dist
            t = torch.tensor([dtd], device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dtd = t.item()
        U = fused._dedup_info(pair, args.bs)["U"]
        dfl = sum(step_flops_dedup(args.bs, k, U, args.arch, args.res) for k in ks[args.warmup:])
        dedup_out = {"value": world * args.steps / dtd, "unit": "steps/s", "ms_per_step": dtd / args.steps * 1e3,
                     "distinct_frozen_prompts": U, "loss": float(dlosses[-1].item()), "losses": [float(l.item()) for l in dlosses],
                     "achieved": dfl / dtd / 1e12, "peak": PEAK_BF16 / 1e12, "frac": dfl / dtd / PEAK_BF16,
                     "note": "same seeded k sequence on FusedStep(dedup=True): the guidance-1 passes run on the conditional samples "
                             "only (train_util.py:151,163-166: u + 1 (c - u) = c) and identical prompts once (train_lora.py:202-237); "
                             "achieved = W_min(k) = bs F_fwd (2 k + U + 2 + a) over wall time (barrier to barrier), whole job"}
    e1 = step_ev[-1] if not emu else None
    dt_ev = e0.elapsed_time(e1) * 1e-3 if not emu else dt
    gpu_done = [e0.elapsed_time(ev) * 1e-3 for ev in step_ev] if not emu else list(host_done)
    step_ms = [(b - a) * 1e3 for a, b in zip([0.0] + gpu_done[:-1], gpu_done)]
    ks_same = True
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([dt], device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
        all_ks = [None] * world
        dist.all_gather_object(all_ks, ks)
        ks_same = all(x == all_ks[0] for x in all_ks)
        # that was the last collective: take the process group down HERE, at the same point on every rank, instead of
        # leaving it to interpreter exit at different times (ranks != 0 return now, rank 0 goes on alone for a minute
        # with the per-launch timing; an implicit teardown of a communicator whose peers are gone is where torch's
        # c10d back ends abort -- seen once as "terminate called without an active exception" on rank 1 of the gloo
        # dry run, which makes torchrun kill rank 0)
        shutdown_distributed()
    losses = [float(l.item()) for l in losses]
    finite = all(math.isfinite(l) for l in losses)
    if rank != 0:
        if not finite:
            sys.exit(f"rank {rank}: non-finite loss in the timed steps: {losses}")
        return
    timed_ks = ks[args.warmup:]
    default_cfg = (args.arch, args.res, args.bs, args.rank, args.c3lier) == ("sd15", 512, 2, 4, False)
    flops = sum(step_flops(args.bs, k, args.arch, args.res) for k in timed_ks)
    achieved = flops / dt_ev / 1e12
    out = {
        "metric": "LECO train-steps/sec, SDv1.5 rank-4 512px bs=2" if default_cfg else
                  f"LECO train-steps/sec, {args.arch} rank-{args.rank}{' c3lier' if args.c3lier else ''} {args.res}px bs={args.bs}",
        "value": world * args.steps / dt, "unit": "steps/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": f"{'SDv1.5' if args.arch == 'sd15' else args.arch} UNet (random init) LECO erase step, LoRA rank {args.rank} "
                               f"{'c3lier' if args.c3lier else 'lierla'}{', v-prediction' if args.v_pred else ''}, "
                               f"{args.res}x{args.res}, prompt batch {args.bs} (UNet batch {2 * args.bs}), DDIM 50, "
                               f"reference-faithful pass structure (k+3+1 fwd, 1 bwd)",
                   "global_batch": args.bs * world, "k_sequence_seed": 0 if args.k <= 0 else f"fixed k={args.k} (profiling run)", "k_mean": sum(timed_ks) / len(timed_ks),
                   "hip_graphs": bool(unet.use_graphs), "parallelism": f"dp{world}", "loss": losses[-1], "losses": losses,
                   "collectives_per_step": collectives / args.steps, "k_identical_across_ranks": ks_same},
    }
    if dedup_out is not None:
        out["dedup"] = dedup_out
    if emu:
        out["data"] = "synthetic; HOST EMULATOR DRY RUN (LECO_BENCH_EMU=1): plumbing check, not a measurement"
        out["dtype"] = "bf16 (emulated)"
    whole = {"achieved": achieved, "peak": PEAK_BF16 / 1e12, "unit": "TFLOP/s", "frac": achieved / (PEAK_BF16 / 1e12),
             "note": "algorithmic FLOPs W_ref(k) = 2 bs F_fwd (k + 5 + a) summed over the timed steps / HIP-event time on the "
                     "compute stream (rank 0)"}
    # ---- what the timed region looked like from the inside (round-4 verdict item 1): one HIP event per step, when the host
    # had finished enqueueing each step, the per-step time normalised by the step's own work
    a_share = F_FWD.get((args.arch, args.res), (0.0, ATTN_SHARE))[1]
    fe = [k + 5 + a_share for k in timed_ks]                     # forward-equivalents of each timed step, W_ref(k) / (2 bs F_fwd)
    per_fe = [m / f for m, f in zip(step_ms, fe)]
    srt = sorted(step_ms)
    half = max(1, len(per_fe) // 2)
    out["timing"] = {
        "step_ms": [round(x, 3) for x in step_ms], "k": timed_ks,
        "step_ms_min_median_max": [srt[0], srt[len(srt) // 2], srt[-1]],
        "sum_step_ms": sum(step_ms), "hip_event_ms": dt_ev * 1e3, "wall_ms": dt * 1e3,
        "ms_per_forward_equivalent": [round(x, 4) for x in per_fe],
        # a drift of the normalised step time along the loop = clocks (DVFS / power cap / temperature) moving under sustained load
        "ms_per_forward_equivalent_first_half_vs_second_half": [sum(per_fe[:half]) / half, sum(per_fe[half:]) / max(1, len(per_fe) - half)],
        # how far the host's enqueueing ran AHEAD of the GPU's completion of the same step: ~0 means the GPU waited for the host
        "host_lead_ms": [round((g - h) * 1e3, 2) for g, h in zip(gpu_done, host_done)],
        "host_enqueue_ms_per_step": host_done[-1] / args.steps * 1e3,
        "note": "step_ms: differences of one HIP event per step on the compute stream (their sum is the headline's timed region); "
                "host_lead_ms[i] = GPU completion time of step i - time the host returned from enqueueing it: positive = the GPU was "
                "the bottleneck, near zero = the launch path was",
    }
    if box is not None:
        out["box_calibration"] = box
    if tele:
        out["telemetry"] = {"idle_before_warmup": tele_idle, "before_timed": tele_before, "after_timed": tele_after,
                            "during_timed": Telemetry.summarize(tele_samples, t0, t1_host),
                            "accumulated_over_timed": Telemetry.delta(tele_before, tele_after), "sampler_hz": args.telemetry_hz}
    try:
        if args.no_dominant or emu:
            raise RuntimeError("--no-dominant")
        st = fused._state[(args.bs, args.res // 8, args.res // 8)]
        if tele:
            tele.start(args.telemetry_hz)
        try:
            dom = dominant_kernel_roofline(st, sum(timed_ks) / len(timed_ks), dump_shapes=args.dump_shapes,
                                           headline_workload=(args.arch == "sd15" and args.bs == 2 and args.res == 512 and args.rank == 4
                                                              and not args.c3lier and not args.v_pred))
        finally:
            if tele:   # clocks while the launches were timed in isolation (bursts): the reference point for the loop's clocks
                out["telemetry"]["during_isolated_launch_timing"] = Telemetry.summarize(tele.stop())
        iso = dom.get("isolated_us")
        if iso:
            # busy fraction per step: isolated GPU time of the launches the step issues (fixed lists + k denoising passes) over
            # the step's measured time.  Uniformly low with low clocks in `telemetry` = DVFS; dips on single steps = launch path
            busy = [(iso["fixed"] + k * iso["per_denoise_pass"]) * 1e-3 / m for k, m in zip(timed_ks, step_ms)]
            out["timing"]["gpu_busy_frac"] = [round(b, 4) for b in busy]
            out["timing"]["gpu_busy_frac_mean"] = sum(busy) / len(busy)
            out["timing"]["isolated_launch_sum_ms_per_step"] = sum(iso["fixed"] + k * iso["per_denoise_pass"] for k in timed_ks) * 1e-3 / len(timed_ks)
        # the roofline object is the DOMINANT KERNEL's (algorithmic FLOPs per launch / its average launch duration);
        # the whole-step figure rides along
        out["roofline"] = {"bound": "mfma", "achieved": dom["achieved"], "peak": dom["peak"], "unit": "TFLOP/s",
                           "frac": dom["frac"], "traffic": dom.get("traffic"), "kernel": dom, "whole_step": whole,
                           "hbm_bound": dom.pop("hbm_bound", None)}
    except Exception as e:  # never hide the step number
        out["roofline"] = {"bound": "mfma", "achieved": whole["achieved"], "peak": whole["peak"], "unit": "TFLOP/s",
                           "frac": whole["frac"], "traffic": None, "whole_step": whole, "kernel": {"error": repr(e)}}
    if world == 1 and not args.no_cpu_baseline and not emu:
        try:
            out["cpu_baseline"] = cpu_baseline(sum(timed_ks) / len(timed_ks), args.bs)
        except Exception as e:  # the baseline leg must never hide the GPU number
            out["cpu_baseline"] = {"value": None, "error": repr(e)}
    print(json.dumps(out))
    if not finite:   # a timing of a broken computation is not a result
        sys.exit(f"non-finite loss in the timed steps: {losses}")


if __name__ == "__main__":
    main()
