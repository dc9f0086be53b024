"""Patch-convolution ablation on the GPU box: libleco_hip variants with -DLECO_CONV_ABLATE=1 (no steady-state LDS fragment
reads) / 2 (no MFMAs) against the product build on the level-0 / level-1 3x3 shapes -- which pipe bounds the tap loop
(DESIGN.md 8.00 item 4 predicts the LDS reads for the 128-row tiles).  Build the variants in the build container first
(`python tools/ablate_conv.py --build`: they travel to the GPU box under tools/_ablate/)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from leco_amd import build as B  # noqa: E402


def build_variant(v):
    d = os.path.join(ROOT, "tools", "_ablate")
    os.makedirs(d, exist_ok=True)
    out = os.path.join(d, f"libleco_convablate{v}.so")
    srcs = [os.path.join(B.CSRC, f) for f in sorted(os.listdir(B.CSRC)) if f.endswith((".hip", ".cpp"))]
    subprocess.run([B.HIPCC, *B.FLAGS, f"-DLECO_CONV_ABLATE={v}", "-shared", "-x", "hip", *srcs, "-o", out], check=True)
    return out


def run_case():
    import math
    import torch
    from leco_amd import hip, ops
    dev = torch.device("cuda:0")
    bf = torch.bfloat16
    x = torch.randn(4096, 4096, device=dev)
    for _ in range(20):
        (x @ x).sum().item()
    for name, B_, hw, cin, cout, tile in (("L0 320->320 @64^2 <128,160>", 4, 64, 320, 320, 8), ("L0 640->320 @64^2 <128,160>", 4, 64, 640, 320, 8),
                                          ("L1 640->640 @32^2 <128,128>", 4, 32, 640, 640, 9), ("L1 640->640 @32^2 <128,160>", 4, 32, 640, 640, 8),
                                          ("L1 1280->640 @32^2 <256,128>", 4, 32, 1280, 640, 7)):
        m, k = B_ * hw * hw, 9 * cin
        a = (torch.randn(m, cin, device=dev) * 0.5).to(bf)
        w = (torch.randn(cout, k, device=dev) / math.sqrt(k)).to(bf)
        y = torch.zeros(m, cout, dtype=bf, device=dev)
        g = hip.gemm_args(a, w, y, m=m, n=cout, k=k, lda=cin, a_mode=hip.A_CONV3_S1, conv=(B_, hw, hw, hw, hw))
        op = ops.gemm(g, keep=(a, w, y), tile=tile, split_k=1)
        for _ in range(5):
            op.run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            op.run()
        e1.record()
        e1.synchronize()
        us = e0.elapsed_time(e1) / 30 * 1e3
        extra = ""
        if os.environ.get("LECO_CONV_STAMPS"):       # a build with LECO_CONV_ABLATE & 32: workgroup 0's shader-clock stamps
            torch.cuda.synchronize()
            pro, loop, epi, rt = y.view(torch.int64).flatten()[:4].tolist()
            extra = (f"   wg0: prologue {pro} / tap loop {loop} / epilogue {epi} shader cycles, {rt / 100.0:.1f} us wall"
                     f" -> {(pro + loop + epi) / max(rt, 1) * 0.1:.2f} GHz")
        print(f"{name:32s} {us:7.1f} us  {2.0 * m * cout * k / us / 1e6:7.1f} TFLOP/s{extra}", flush=True)


WHAT = {0: "product build", 1: "no LDS fragment reads", 2: "no MFMAs", 4: "no wait for the weight DMA", 8: "no barrier in the tap step",
        16: "no steady-state DMA", 32: "clock stamps"}


def describe(v):
    return " + ".join(WHAT[b] for b in (1, 2, 4, 8, 16, 32) if v & b) or WHAT[0]


if __name__ == "__main__":
    if "--build" in sys.argv:
        i = sys.argv.index("--build")
        vs = [int(x) for x in sys.argv[i + 1].split(",")] if len(sys.argv) > i + 1 else [1, 2]
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(4) as ex:
            for out in ex.map(build_variant, vs):
                print(out)
    elif "--case" in sys.argv:
        run_case()
    else:
        d = os.path.join(ROOT, "tools", "_ablate")
        have = sorted(int(f[len("libleco_convablate"):-3]) for f in os.listdir(d) if f.startswith("libleco_convablate"))
        for lw in ("0", "1"):
            for v in [0] + have:
                env = dict(os.environ, LECO_CONV_LW=lw)
                if v:
                    env["LECO_HIP_LIB"] = os.path.join(d, f"libleco_convablate{v}.so")
                if v & 32:
                    env["LECO_CONV_STAMPS"] = "1"
                print(f"# LECO_CONV_LW={lw}  {v}: {describe(v)}", flush=True)
                subprocess.run([sys.executable, os.path.abspath(__file__), "--case"], env=env)
