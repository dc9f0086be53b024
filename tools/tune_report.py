#!/usr/bin/env python
"""Measure every GEMM launch shape of the step's plans (leco_amd/tune.py, LECO_GEMM_TUNE=force) and write the table.

    python tools/tune_report.py [--arch sd15 --bs 2 --res 512 --rank 4 [--c3lier]] [--out gpurun_out/gemm_tune_gfx950.json]

Prints, per shape: launches per denoising pass / per step-fixed part, the C heuristic's time, the best (tile, split_k)
and its time; then the projected saving per pass."""
import argparse
import contextlib
import io
import os
import sys
from collections import Counter

os.environ["LECO_GEMM_TUNE"] = "force"
import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from leco_amd import model_util, tune  # noqa: E402
from leco_amd.lora import DEFAULT_TARGET_REPLACE, UNET_TARGET_REPLACE_MODULE_CONV, LoRANetwork  # noqa: E402
from leco_amd.train import FusedStep  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--arch", default="sd15")
    ap.add_argument("--bs", type=int, default=2)
    ap.add_argument("--res", type=int, default=512)
    ap.add_argument("--rank", type=int, default=4)
    ap.add_argument("--c3lier", action="store_true")
    ap.add_argument("--out", default="gpurun_out/gemm_tune_gfx950.json")
    ap.add_argument("--dedup", action="store_true",
                    help="also build (= tune) the plans of the de-duplicated pass structure: training plan at UNet batch bs, frozen "
                         "plans at U bs for U = 2 and 3 distinct prompts")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    x = torch.randn(4096, 4096, device=dev)
    for _ in range(30):
        (x @ x).sum().item()      # clock ramp before anything is timed
    if args.arch == "sdxl":
        _, _, unet, sched = model_util.load_models_xl("synthetic:sdxl", "ddim")
    else:
        _, _, unet, sched = model_util.load_models(f"synthetic:{args.arch}", "ddim")
    unet.to(dev, dtype=torch.bfloat16)
    unet.requires_grad_(False)
    targets = list(DEFAULT_TARGET_REPLACE) + (list(UNET_TARGET_REPLACE_MODULE_CONV) if args.c3lier else [])
    with contextlib.redirect_stdout(io.StringIO()):
        net = LoRANetwork(unet, rank=args.rank, multiplier=1.0, alpha=1.0, target_replace_modules=targets).to(dev)
    fused = FusedStep(unet, net, sched, 50, lr=1e-4)
    st = fused._bucket(args.bs, args.res // 8, args.res // 8)       # builds the three plans -> tunes every shape
    extra = []
    if args.dedup:
        for U in (2, 3):
            pd, fd = fused._deduped(st, U)
            extra += [(fd, "fwd_off")] + ([(pd, "fwd_on"), (pd, "bwd")] if U == 2 else [])
    torch.cuda.synchronize()
    per_pass = Counter()
    for op in st["dplan"].lists["denoise"]:
        if op.name == "leco_gemm_ex":
            per_pass[tune.shape_key(op.keep[0])] += 1
    fixed = Counter()
    for plan, which in [(st["fplan"], "fwd_off"), (st["plan"], "fwd_on"), (st["plan"], "bwd"), (st["dplan"], "ctx_on")] + extra:
        for op in plan.lists[which]:
            if op.name == "leco_gemm_ex":
                fixed[tune.shape_key(op.keep[0])] += 1
    rows = tune.report()
    save_pass = save_fixed = tot_pass = tot_fixed = 0.0
    print(f"{'pass':>4} {'fix':>4} {'heur us':>8} {'best us':>8} {'best':>8}  shape")
    for key, t0, best, t1 in sorted(rows, key=lambda r: -(per_pass[r[0]] * 24 + fixed[r[0]]) * ((r[1] or 0) - (r[3] or 0))):
        if t0 is None or t1 is None:
            continue
        print(f"{per_pass[key]:4d} {fixed[key]:4d} {t0:8.1f} {t1:8.1f} {str(best):>8}  {key}")
        save_pass += per_pass[key] * (t0 - t1)
        save_fixed += fixed[key] * (t0 - t1)
        tot_pass += per_pass[key] * t0
        tot_fixed += fixed[key] * t0
    print(f"# GEMM time per denoising pass: heuristic {tot_pass/1e3:.3f} ms -> tuned {(tot_pass-save_pass)/1e3:.3f} ms; "
          f"fixed part of a step (3 frozen + target fwd + bwd): {tot_fixed/1e3:.3f} -> {(tot_fixed-save_fixed)/1e3:.3f} ms")
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    n = tune.save_table(args.out, merge=True)      # keep the committed entries of the other configurations
    print(f"# wrote {n} entries to {args.out}")


if __name__ == "__main__":
    main()
