"""Experiment (round 6): do the k denoising passes run faster as TWO independent micro-batches on two HIP streams?

The samples of a prompt batch never meet inside `diffusion` (train_util.py:172-193): sample 0 and sample 1 of a bs = 2 step are
two independent chains of k UNet passes (each CFG-doubled: UNet batch 2).  Most launches of a UNet-batch-4 pass fill a fraction of
the 256 CUs or sit at a latency floor, so two chains in flight at once could fill the gaps -- at the price of
reading the weights once per chain and of smaller GEMM M.  Measured here: k passes of the product's B = 4 plan (one graph per
pass, one stream) against k passes of two B = 2 plans, each replayed from its own graph on its own stream.

    python tools/exp_microbatch.py [--k 20] [--arch sd15] [--res 512] [--bs 2]"""
import argparse
import contextlib
import io
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from leco_amd import model_util, ops, prompt_util, train_util  # noqa: E402
from leco_amd.lora import LoRANetwork  # noqa: E402
from leco_amd.train import DENOISE_GUIDANCE, FusedStep  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--k", type=int, default=20)
    ap.add_argument("--arch", default="sd15")
    ap.add_argument("--res", type=int, default=512)
    ap.add_argument("--bs", type=int, default=2)
    ap.add_argument("--rank", type=int, default=4)
    ap.add_argument("--parts", type=int, default=2)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    tok, te, unet, sched = model_util.load_models(f"synthetic:{args.arch}", "ddim")
    unet.to(dev, dtype=torch.bfloat16)
    unet.requires_grad_(False)
    unet.use_graphs = True
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
    fused = FusedStep(unet, net, sched, 50, lr=1e-4)
    h = w = args.res // 8
    lat = train_util.get_initial_latents(sched, args.bs, args.res, args.res, 1, generator=torch.Generator().manual_seed(1))
    fused.step(pair, 2, lat)          # builds the product plans, packs the LoRA operands
    torch.cuda.synchronize()
    st = fused._state[(args.bs, h, w)]
    dplan = st["dplan"]
    eng = unet.engine()
    x0 = st["x"].clone()

    def timed(fn, reps=3):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(reps):
            e0.record()
            fn()
            e1.record()
            e1.synchronize()
            best = min(best, e0.elapsed_time(e1))
        return best

    def whole():
        st["x"].copy_(x0)
        ops.step_begin(st["x"], dplan.x_in, 1.0, st["half_n"], dplan.t_idx).run()
        for _ in range(args.k):
            unet._run(dplan, st["dn"])
    t_whole = timed(whole)
    x_whole = st["x"].clone()

    # ---- micro-batches: `parts` plans of UNet batch 2 bs / parts, each with its own split-K workspace, latents and pass counter
    mb = args.bs // args.parts
    parts = []
    for i in range(args.parts):
        pl = eng.plan(2 * mb, h, w, need_bwd=False, share=2, ws_slot=10 + i, tag=f"mb{i}")
        x = torch.zeros(mb, 4, h, w, dtype=torch.float32, device=dev)
        half_n = mb * 4 * h * w
        tail = [ops.cfg_ddim_step(pl.pred, x, pl.x_in, fused.coef, pl.t_idx, DENOISE_GUIDANCE, half_n), ops.advance(pl.t_idx)]
        pl.lists["ctx_on"] = [op for op in pl.lists["fwd_on"] if op.tag == "ctx"]
        pl.lists["denoise"] = [op for op in pl.lists["fwd_on"] if op.tag != "ctx"] + tail
        pl.t_table[:fused.n_steps].copy_(fused.ts_f)
        ctx = train_util.concat_embeddings(pair.unconditional, pair.target, mb).to(dev, eng.adt).contiguous()
        pl.set_ctx(ctx)
        unet._run(pl, "ctx_on")
        parts.append(dict(plan=pl, x=x, half_n=half_n, stream=torch.cuda.Stream() if i else None))
    torch.cuda.synchronize()

    def split(concurrent=True):
        cur = torch.cuda.current_stream()
        for i, P in enumerate(parts):
            P["x"].copy_(x0[i * mb:(i + 1) * mb])
            ops.step_begin(P["x"], P["plan"].x_in, 1.0, P["half_n"], P["plan"].t_idx).run()
        if concurrent:
            for P in parts[1:]:
                P["stream"].wait_stream(cur)
        for _ in range(args.k):
            for P in parts:
                if concurrent and P["stream"] is not None:
                    with torch.cuda.stream(P["stream"]):
                        unet._run(P["plan"], "denoise")
                else:
                    unet._run(P["plan"], "denoise")
        if concurrent:
            for P in parts[1:]:
                cur.wait_stream(P["stream"])
    t_serial = timed(lambda: split(False))
    t_conc = timed(lambda: split(True))
    x_split = torch.cat([P["x"] for P in parts])
    err = ((x_split - x_whole).norm() / x_whole.norm()).item()
    print(f"# {args.arch} {args.res}^2 bs={args.bs} k={args.k}: denoising chain, hipGraph replay per pass")
    print(f"one chain, UNet batch {2 * args.bs}:                       {t_whole:8.2f} ms  ({t_whole / args.k:6.3f} ms per pass)")
    print(f"{args.parts} chains of UNet batch {2 * mb}, one stream (back to back): {t_serial:8.2f} ms  ({t_serial / args.k:6.3f} ms per pass-equivalent)")
    print(f"{args.parts} chains of UNet batch {2 * mb}, one stream each:          {t_conc:8.2f} ms  ({t_conc / args.k:6.3f} ms per pass-equivalent)   "
          f"x{t_whole / t_conc:.3f} vs one chain")
    print(f"denoised latents, split vs whole: rel {err:.3e}")


if __name__ == "__main__":
    main()
