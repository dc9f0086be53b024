#!/usr/bin/env python
"""Where does one UNet pass go?  Times EVERY launch of a plan's list in isolation (HIP events, back-to-back repeats of
the same op, so L2-warm) and prints them grouped by (kernel, shape) with the achieved TFLOP/s or GB/s of each group.

    python tools/plan_profile.py [--list denoise|fwd_off|fwd_on|bwd|frozen] [--arch sd15] [--bs 2] [--res 512] [--top 40]

`denoise`: the forward-only LoRA-ON plan of the k denoising passes (B = 2 bs); `frozen`: the batched LoRA-OFF plan
(B = 6 bs); `fwd_on` / `bwd`: the training plan."""
import argparse
import contextlib
import io
import os
import sys
from collections import OrderedDict

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from leco_amd import model_util, prompt_util, train_util  # noqa: E402
from leco_amd.lora import LoRANetwork  # noqa: E402
from leco_amd.train import FusedStep  # noqa: E402

AMODE = {0: "plain", 1: "conv3", 2: "conv3s2", 3: "conv3up2", 4: "conv3tr2"}


def describe(op):
    """-> (group key, flops, bytes)"""
    a = op.args
    if op.name == "leco_gemm_ex":
        g = op.keep[0]
        ext = g.ext_k if (g.a_ext or g.t_w) else 0
        key = (f"gemm {AMODE[g.a_mode]} M={g.m} N={g.n} K={g.k}" + (f" +lora{ext}{'(fusedT)' if g.t_w else ''}" if ext else "")
               + (" geglu" if g.act == 2 else "") + (" +res" if g.residual else ""))
        kin = g.k // 9 if g.a_mode else g.k
        rows_in = g.m if g.a_mode == 0 else g.batch * g.h_in * g.w_in
        return key, 2.0 * g.m * g.n * (g.k + ext), 2.0 * (rows_in * kin + g.n * g.k + g.m * g.n * (0.5 if g.act == 2 else 1))
    if op.name == "leco_attention_fwd":
        B, H, sq, skv, d = a[13], a[14], a[15], a[16], a[17]
        return f"attn_fwd B={B} H={H} Sq={sq} Skv={skv} d={d}", 4.0 * B * H * sq * skv * d, 2.0 * B * H * d * (2 * sq + 2 * skv)
    if op.name == "leco_attention_bwd":
        B, H, sq, skv, d = a[26], a[27], a[28], a[29], a[30]
        return f"attn_bwd B={B} H={H} Sq={sq} Skv={skv} d={d}", 10.0 * B * H * sq * skv * d, 2.0 * B * H * d * (4 * sq + 4 * skv)
    if op.name == "leco_groupnorm_fwd":
        B, hw, c = a[7], a[8], a[9]
        return f"groupnorm_fwd B={B} HW={hw} C={c} act={a[12]}", 0.0, 4.0 * B * hw * c
    if op.name == "leco_groupnorm_bwd":
        B, hw, c = a[10], a[11], a[12]
        return f"groupnorm_bwd B={B} HW={hw} C={c}", 0.0, 8.0 * B * hw * c
    if op.name == "leco_layernorm_fwd":
        return f"layernorm_fwd M={a[5]} C={a[6]}", 0.0, 4.0 * a[5] * a[6]
    if op.name == "leco_layernorm_bwd":
        return f"layernorm_bwd M={a[9]} C={a[10]}", 0.0, 8.0 * a[9] * a[10]
    if op.name in ("leco_geglu_fwd", "leco_geglu_bwd"):
        return f"{op.name[5:]} M={a[-2]} F={a[-1]}", 0.0, 6.0 * a[-2] * a[-1]
    if op.name == "leco_xblock_tail":
        A = op.keep[0]
        m, c, po = A.m, A.c, 1 if A.proj_out.w else 0
        return (f"xblock_tail M={m} C={c} d={c // A.heads}" + (" +proj_out" if po else ""),
                m * (2.0 * c * c * (15 + po) + 4.0 * A.skv * c), 2.0 * m * c * (3 + po))
    if op.name == "leco_xblock_head":
        A = op.keep[0]
        return f"xblock_head M={A.m} C={A.c}" + (" +gn" if A.gn_cstats else ""), A.m * 2.0 * A.c * A.c * 4, 2.0 * A.m * A.c * 5
    if op.name == "leco_xgemm":
        A = op.keep[0]
        ext = 32 if A.lin.dn else 0
        key = f"xgemm plain M={A.m} N={A.n} K={A.k}" + (f" +lora{ext}(fusedT)" if ext else "") + (" +res" if A.residual else "")
        return key, 2.0 * A.m * A.n * (A.k + ext), 2.0 * (A.m * A.k + A.n * A.k + A.m * A.n)
    return op.name[5:], 0.0, 0.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--list", default="denoise")
    ap.add_argument("--arch", default="sd15")
    ap.add_argument("--bs", type=int, default=2)
    ap.add_argument("--res", type=int, default=512)
    ap.add_argument("--rank", type=int, default=4)
    ap.add_argument("--top", type=int, default=45)
    ap.add_argument("--iters", type=int, default=12)
    ap.add_argument("--blas", action="store_true",
                    help="also time the vendor library (torch.matmul -> hipBLASLt / rocBLAS, bf16) on the bare [M][K] x [K][N] product "
                         "of every plain GEMM shape: the 'achievable' denominator for a contraction of that size WITHOUT the "
                         "epilogue work (bias / residual / GEGLU / LoRA) the plan's launch carries -- measurement only")
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
    fused = FusedStep(unet, net, sched, 50, lr=1e-4)
    lat = train_util.get_initial_latents(sched, args.bs, args.res, args.res, 1, generator=torch.Generator().manual_seed(1))
    fused.step(pair, 2, lat)          # builds the plans, fills every buffer with real activations
    torch.cuda.synchronize()
    st = fused._state[(args.bs, args.res // 8, args.res // 8)]
    plan, which = {"denoise": (st["dplan"], "denoise"), "frozen": (st["fplan"], "fwd_off"), "fwd_on": (st["plan"], "fwd_on"),
                   "fwd_off": (st["plan"], "fwd_off"), "bwd": (st["plan"], "bwd")}[args.list]
    # the CFG / DDIM update and the timestep advance mutate the step state: leave them out of the repeats
    ops_ = [op for op in plan.lists[which] if op.name not in ("leco_advance", "leco_cfg_ddim_step", "leco_cfg_sched_step")]
    x = torch.randn(4096, 4096, device=dev)
    for _ in range(20):
        (x @ x).sum().item()      # clock ramp
    groups = OrderedDict()
    blas = {}
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def vendor_us(m, n, k):
        a_ = (torch.rand(m, k, device=dev) * 2 - 1).to(torch.bfloat16)
        b_ = (torch.rand(k, n, device=dev) * 2 - 1).to(torch.bfloat16)
        for _ in range(3):
            torch.matmul(a_, b_)
        e0.record()
        for _ in range(args.iters):
            torch.matmul(a_, b_)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / args.iters * 1e3
    for op in ops_:
        for _ in range(3):
            op.run()
        e0.record()
        for _ in range(args.iters):
            op.run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / args.iters * 1e3
        key, fl, by = describe(op)
        if args.blas and key.startswith("gemm plain") and key not in blas:
            g_ = op.keep[0]
            blas[key] = vendor_us(g_.m, g_.n, g_.k)
        gsum = groups.setdefault(key, [0, 0.0, 0.0, 0.0])
        gsum[0] += 1
        gsum[1] += us
        gsum[2] += fl
        gsum[3] += by
    # the whole list back to back, for comparison with the sum of the isolated launches
    for _ in range(2):
        for op in ops_:
            op.run()
    e0.record()
    for _ in range(5):
        for op in ops_:
            op.run()
    e1.record()
    torch.cuda.synchronize()
    whole = e0.elapsed_time(e1) / 5 * 1e3
    tot = sum(v[1] for v in groups.values())
    tfl = sum(v[2] for v in groups.values())
    print(f"# {args.arch} {args.res}^2 bs={args.bs} list={args.list}: {len(ops_)} launches, sum of isolated launches {tot/1e3:.3f} ms, "
          f"list back-to-back (eager) {whole/1e3:.3f} ms, {tfl/1e12:.3f} TFLOP -> {tfl/whole/1e6:.1f} TFLOP/s")
    print(f"{'%':>6} {'n':>4} {'us each':>9} {'TFLOP/s':>8} {'GB/s':>8}  op")
    for key, (n, us, fl, by) in sorted(groups.items(), key=lambda kv: -kv[1][1])[:args.top]:
        print(f"{100*us/tot:6.2f} {n:4d} {us/n:9.1f} {fl/us/1e6 if fl else 0:8.1f} {by/us/1e3:8.0f}  {key}"
              + (f"   | vendor GEMM {blas[key]:.1f} us (x{us / n / blas[key]:.2f})" if key in blas else ""))
    if blas:
        ours = sum(us for key, (n, us, fl, by) in groups.items() if key in blas)
        theirs = sum(n * blas[key] for key, (n, us, fl, by) in groups.items() if key in blas)
        print(f"# plain GEMMs of the list: {ours/1e3:.3f} ms here (with their epilogues and fused LoRA) vs {theirs/1e3:.3f} ms for "
              f"the bare vendor GEMMs of the same M, N, K")
    fam = OrderedDict()
    for key, (n, us, fl, by) in groups.items():
        f = " ".join(key.split()[:2]) if key.startswith("gemm") else key.split()[0]
        a = fam.setdefault(f, [0, 0.0, 0.0])
        a[0] += n
        a[1] += us
        a[2] += fl
    print("# by family")
    for f, (n, us, fl) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
        print(f"# {100*us/tot:6.2f}% {n:4d} launches {us/1e3:8.3f} ms  {fl/us/1e6 if fl else 0:7.1f} TFLOP/s  {f}")


if __name__ == "__main__":
    main()
