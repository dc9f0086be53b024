"""Stripe-kernel ablation / phase timing on the GPU box (measurement only).

    python tools/ablate_stripe.py --build-only [variants...]     (here: hipcc cross-compiles the side libraries)
    python tools/ablate_stripe.py [variants...]                  (GPU box)

Variants: `t` = product kernel + phase stamps of workgroup 0 (LECO_STRIPE_TIMING); an integer = LECO_STRIPE_ABLATE bit mask
(1 no MFMA, 2 no weight loads, 4 no activation-fragment reads, 8 no cross-attention)."""
import ctypes as C
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from leco_amd import build as B, hip, ops  # noqa: E402

bf = torch.bfloat16
PACKED = os.environ.get("STRIPE_ROWMAJOR", "0") == "0"     # (timing only: the values are random either way)
PHASES = ["to_out1", "LN2", "to_q2 + store", "(stamp)", "cross-attn", "to_out2", "LN3", "FF (10 chunks)", "proj_out", "store_out"]


def build_variant(v):
    d = os.path.join(ROOT, "tools", "_ablate")
    os.makedirs(d, exist_ok=True)
    out = os.path.join(d, f"libleco_stripe_{v}.so")
    src = os.path.join(B.CSRC, "stripe.hip")
    if os.path.exists(out) and os.path.getmtime(out) > os.path.getmtime(src):
        return out
    B.build()
    objs = [o for o in open(os.path.join(B.OBJ, "link.stamp")).read().split() if "/stripe.hip." not in o]
    obj = os.path.join(d, f"stripe_{v}.o")
    define = ["-DLECO_STRIPE_TIMING"] if v == "t" else [f"-DLECO_STRIPE_ABLATE={v}"]
    subprocess.run([B.HIPCC, *B.FLAGS, *define, "-x", "hip", "-c", src, "-o", obj], check=True)
    subprocess.run([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, *objs, obj], check=True)
    return out


def make_case(dev, M=16384, hw=4096, heads=8, rank=4, C=320, skv=77):
    torch.manual_seed(0)
    Bn = M // hw
    D = C // heads

    def lin(n, k, bias=True):
        w = (torch.randn(n, k, device=dev) / k ** 0.5).to(bf)
        b = torch.randn(n, device=dev) * 0.1 if bias else None
        if rank:
            dn = torch.zeros(16, k, device=dev); dn[:rank] = torch.randn(rank, k, device=dev) / k ** 0.5
            up = torch.zeros(n, 32, device=dev); up[:, :rank] = torch.randn(n, rank, device=dev) * 0.02
            dn, up = dn.to(bf), up.to(bf)
            return hip.xlin(w, b, dn, up, 16, packed=PACKED), (w, b, dn, up)
        return hip.xlin(w, b, packed=PACKED), (w, b)
    keep = []
    A = hip.XBlockTailArgs()
    A.m, A.c, A.heads, A.skv, A.rows_per_sample = M, C, heads, skv, hw
    t = {n: torch.randn(M, C, device=dev).to(bf) for n in ("a1", "h0", "x")}
    out = torch.zeros(M, C, dtype=bf, device=dev)
    kv = torch.randn(Bn * skv, 2 * C, device=dev).to(bf)
    kp, vt = ops.xattn_buffers(Bn, heads, D, dev)
    ops.xattn_prep(kv.data_ptr(), 2 * C, kp, vt, Bn, heads, skv, D).run()
    A.attn, A.ld_attn, A.h_in, A.ld_h = t["a1"].data_ptr(), C, t["h0"].data_ptr(), C
    for f, (n, k, b) in dict(to_out1=(C, C, True), to_q2=(C, C, False), to_out2=(C, C, True), ff1=(8 * C, C, True),
                             ff2=(C, 4 * C, True), proj_out=(C, C, True)).items():
        xl, kp_ = lin(n, k, b)
        setattr(A, f, xl)
        keep.append(kp_)
    ln = [torch.ones(C, device=dev), torch.zeros(C, device=dev)]
    A.ln2_g, A.ln2_b, A.ln3_g, A.ln3_b, A.ln_eps = ln[0].data_ptr(), ln[1].data_ptr(), ln[0].data_ptr(), ln[1].data_ptr(), 1e-5
    A.kp, A.vt, A.attn_scale = kp.data_ptr(), vt.data_ptr(), D ** -0.5
    A.res, A.ld_res, A.out, A.ld_out = t["x"].data_ptr(), C, out.data_ptr(), C
    cst = torch.zeros(Bn, C // 10, 2, device=dev)
    A.col_stats, A.stats_atom = cst.data_ptr(), 10
    return A, (keep, t, out, kv, kp, vt, ln, cst)


def main():
    variants = [a for a in sys.argv[1:] if not a.startswith("--")] or ["t", "1", "2", "4", "7"]
    if "--build-only" in sys.argv:
        for v in variants:
            print(build_variant(v))
        return
    dev = torch.device("cuda:0")
    x = torch.randn(4096, 4096, device=dev)
    for _ in range(30):
        (x @ x).sum().item()
    for M in (16384, 49152):
        for v in ["prod"] + variants:
            hip._use_library(hip.LIB_PATH if v == "prod" else build_variant(v))
            A, keep = make_case(dev, M=M)
            op = ops.xblock_tail(A)
            for _ in range(5):
                op.run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                op.run()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 20 * 1e3
            fl = M * (2.0 * 320 * 320 * 16 + 4.0 * 77 * 320)
            print(f"M={M} variant={v:>4s}: {us:8.1f} us  ({fl / us / 1e6:6.1f} TFLOP/s nominal)", flush=True)
            if v == "t":
                buf = (C.c_ulonglong * 16)()
                lib = hip.lib()
                lib.leco_xblock_debug_times.argtypes = [C.c_void_p, C.c_int]
                lib.leco_xblock_debug_times(buf, 16)
                ts = list(buf)
                tot = ts[10] - ts[0]
                print(f"   workgroup 0: {tot} shader clocks from the first sweep to the end of store_out ({tot / us / 1e3:.2f} GHz if it spans the launch)")
                for i, name in enumerate(PHASES):
                    print(f"   {name:16s} {ts[i + 1] - ts[i]:8d} clk  {100.0 * (ts[i + 1] - ts[i]) / tot:5.1f} %")
                print(f"   first FF chunk: FF1 {ts[11] - ts[7]} clk, GEGLU {ts[12] - ts[11]} clk, FF2 {ts[13] - ts[12]} clk")


if __name__ == "__main__":
    main()
