#!/usr/bin/env python
"""Per-launch finiteness probe of the full-size step (the eager-mode check SURVEY 5.2 asks for).

Builds the bench workload (SD1.5, bs=2, rank 4, 512^2), then walks the launch lists of one FusedStep.step
ONE OP AT A TIME and, after each launch, tests every tensor the op keeps alive (inputs and outputs) for
non-finite values.  The first launch whose tensors turn non-finite is printed with its GEMM shape.

    python tools/nan_probe.py [--k 2] [--steps 2] [--arch sd15] [--bs 2] [--res 512]
"""
import argparse
import contextlib
import io
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from leco_amd import hip, model_util, ops, prompt_util, train_util  # noqa: E402
from leco_amd.lora import LoRANetwork  # noqa: E402
from leco_amd.train import FusedStep  # noqa: E402
from leco_amd.unet import TRef  # noqa: E402


def tensors_of(keep, out=None):
    out = [] if out is None else out
    if isinstance(keep, (tuple, list)):
        for k in keep:
            tensors_of(k, out)
    elif isinstance(keep, TRef):
        out.append((keep.name or "tref", keep.t))
    elif isinstance(keep, torch.Tensor):
        out.append(("tensor%s" % (tuple(keep.shape),), keep))
    return out


def describe(op):
    if op.name == "leco_gemm_ex":
        a = op.keep[0]
        fields = {f: getattr(a, f) for f, _ in a._fields_ if isinstance(getattr(a, f), (int, float))}
        keys = ("m", "n", "k", "a_mode", "act", "ext_k", "t_rows", "k_split")
        return "gemm " + " ".join(f"{k}={fields.get(k)}" for k in keys if k in fields) + f" tile={op.args[1]} split={op.args[2]}"
    return op.name + " " + " ".join(str(a) for a in op.args if isinstance(a, (int, float)) and abs(a) < 1e6)


class Probe:
    def __init__(self, limit=6):
        self.bad = set()
        self.events = 0
        self.limit = limit

    def run(self, oplist, label):
        first = None
        for i, op in enumerate(oplist):
            op.run()
            if self.events >= self.limit:
                continue
            for name, t in tensors_of(op.keep):
                if t.dtype not in (torch.bfloat16, torch.float32) or id(t) in self.bad or t.numel() >= (1 << 25) and t.dtype == torch.float32 and t.dim() == 1:
                    continue
                if not torch.isfinite(t).all().item():
                    self.bad.add(id(t))
                    n_bad = (~torch.isfinite(t)).sum().item()
                    print(f"[{label}] op {i}/{len(oplist)} {describe(op)} tag={op.tag}: tensor '{name}' {tuple(t.shape)} "
                          f"{t.dtype} has {n_bad} non-finite of {t.numel()}", flush=True)
                    self.events += 1
                    if first is None:
                        first = i
        torch.cuda.synchronize()
        return first


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--arch", default="sd15")
    ap.add_argument("--bs", type=int, default=2)
    ap.add_argument("--res", type=int, default=512)
    ap.add_argument("--rank", type=int, default=4)
    ap.add_argument("--k", type=int, default=2)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--dedup", action="store_true", help="walk the de-duplicated pass structure (FusedStep.dedup)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    tokenizer, text_encoder, unet, sched = model_util.load_models(f"synthetic:{args.arch}", "ddim")
    unet.to(dev, dtype=torch.bfloat16)
    unet.requires_grad_(False)
    unet.eval()
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
    emb = {p: text_encoder([p])[0] for p in ("van gogh", "")}
    pair = prompt_util.PromptEmbedsPair(torch.nn.MSELoss(), emb["van gogh"], emb["van gogh"], emb[""], emb[""], settings)
    fused = FusedStep(unet, net, sched, 50, lr=1e-4, world_size=1, dedup=args.dedup)
    noise_gen = torch.Generator().manual_seed(1000)
    probe = Probe()

    # FusedStep.step with its launch lists walked by the probe instead of unet._run
    def probed_run(plan, which):
        probe.run(plan.lists[which], which)
    fused._run = probed_run
    for s in range(args.steps):
        lat = train_util.get_initial_latents(sched, args.bs, args.res, args.res, 1, generator=noise_gen)
        loss = fused.step(pair, args.k, lat)
        torch.cuda.synchronize()
        st = fused._state[(args.bs, args.res // 8, args.res // 8)]
        print(f"step {s}: loss={loss.item():.6e} grad finite={torch.isfinite(net.grad).all().item()} "
              f"|grad|={net.grad.norm().item():.4e} slab finite={torch.isfinite(net.slab).all().item()} "
              f"x finite={torch.isfinite(st['x']).all().item()} |x|max={st['x'].abs().max().item():.3f} "
              f"pred max={st['last']['plan'].pred.abs().max().item():.4f} "
              f"frozen finite={ {n: bool(torch.isfinite(t).all().item()) for n, t in st['last']['preds'].items()} }", flush=True)
    print("probe events:", probe.events)


if __name__ == "__main__":
    main()
