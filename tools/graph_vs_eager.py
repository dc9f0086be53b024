"""Debugging aid: run one launch list of a plan eagerly and from its captured hipGraph on identical inputs, then compare EVERY
buffer of the plan in creation order -- the first buffer that differs names the launch whose behaviour depends on how it is
issued (round 6: the de-duplicated step's training forward gave NaN from its graph and the right loss eagerly).

    python tools/graph_vs_eager.py [--bs 2] [--which fwd_on] [--plan train|frozen|denoise] [--dedup]"""
import argparse
import contextlib
import io
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from leco_amd import model_util, ops, prompt_util, train_util  # noqa: E402
from leco_amd.lora import LoRANetwork  # noqa: E402
from leco_amd.train import FusedStep  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--arch", default="sd15")
    ap.add_argument("--res", type=int, default=512)
    ap.add_argument("--bs", type=int, default=2)
    ap.add_argument("--rank", type=int, default=4)
    ap.add_argument("--which", default="fwd_on")
    ap.add_argument("--plan", default="train")
    ap.add_argument("--dedup", action="store_true")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--bench-like", action="store_true",
                    help="graphs on from the first step, one reference-faithful step first, then flip to --dedup (bench.py's order)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    tok, te, unet, sched = model_util.load_models(f"synthetic:{args.arch}", "ddim")
    unet.to(dev, dtype=torch.bfloat16)
    unet.requires_grad_(False)
    unet.use_graphs = False
    torch.manual_seed(1234)
    with contextlib.redirect_stdout(io.StringIO()):
        net = LoRANetwork(unet, rank=args.rank, multiplier=1.0, alpha=1.0).to(dev)
    g = torch.Generator().manual_seed(99)
    with torch.no_grad():
        for l in net.unet_loras:
            l.lora_up.weight.copy_((torch.randn(l.lora_up.weight.shape, generator=g) * 0.02).to(dev))
    net.mark_updated()
    settings = prompt_util.PromptSettings(target="van gogh", positive="van gogh", unconditional="", neutral="",
                                          action="erase", guidance_scale=1.0, resolution=args.res, batch_size=args.bs)
    emb = {p: te([p])[0] for p in ("van gogh", "")}
    pair = prompt_util.PromptEmbedsPair(torch.nn.MSELoss(), emb["van gogh"], emb["van gogh"], emb[""], emb[""], settings)
    lat = train_util.get_initial_latents(sched, args.bs, args.res, args.res, 1, generator=torch.Generator().manual_seed(1))
    if args.bench_like:
        unet.use_graphs = True
        fused = FusedStep(unet, net, sched, 50, lr=1e-4, dedup=False)
        print(f"graphs on, faithful step loss {fused.step(pair, 3, lat).item():.6e}")
        fused.dedup = args.dedup
        for i in range(2):
            print(f"graphs on, {'dedup' if args.dedup else 'faithful'} step {i} loss {fused.step(pair, 3, lat).item():.6e}")
        unet.use_graphs = False
    else:
        fused = FusedStep(unet, net, sched, 50, lr=1e-4, dedup=args.dedup)
    loss = fused.step(pair, 2, lat)          # eager: builds the plans, leaves real inputs in every plan
    torch.cuda.synchronize()
    print(f"eager step loss {loss.item():.6e}")
    st = fused._state[(args.bs, args.res // 8, args.res // 8)]
    plan = {"train": st["last"]["plan"], "frozen": st["last"]["fplan"], "denoise": st["dplan"]}[args.plan]
    which = st["dn"] if args.which == "denoise" else args.which
    net.multiplier = 0 if which == "fwd_off" else 1.0
    print(f"plan key {plan.key}: list {which} has {len(plan.lists[which])} launches, {len(plan.bufs)} buffers")

    def snapshot():
        torch.cuda.synchronize()
        return {k: v.clone() for k, v in plan.bufs.items()}
    # inputs of the list must not be its outputs: x_in / ctx / t_table are only read by fwd lists
    ops.run_plan(plan.lists[which])
    ref = snapshot()
    ops.run_plan(plan.lists[which])
    again = snapshot()
    unstable = [k for k in ref if not torch.equal(ref[k].view(torch.uint8), again[k].view(torch.uint8))]
    print(f"eager vs eager: {len(unstable)} buffers differ bitwise (atomics / in-place state): {unstable[:12]}")
    unet.use_graphs = True
    for r in range(args.reps):
        unet._run(plan, which)
        got = snapshot()
        bad = []
        for k in ref:
            a, b = ref[k].float(), got[k].float()
            if not torch.isfinite(b).all() or (k not in unstable and not torch.equal(ref[k].view(torch.uint8), got[k].view(torch.uint8))):
                d = (a - b).norm().item() / max(a.norm().item(), 1e-30)
                bad.append((k, d, bool(torch.isfinite(b).all().item())))
        print(f"graph replay {r}: {len(bad)} of {len(ref)} buffers differ from the eager run; first: "
              + "; ".join(f"{k} rel {d:.3g} finite={f}" for k, d, f in bad[:8]))


if __name__ == "__main__":
    main()
