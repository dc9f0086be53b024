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
        print(f"{name:32s} {us:7.1f} us  {2.0 * m * cout * k / us / 1e6:7.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    if "--build" in sys.argv:
        for v in (1, 2):
            print(build_variant(v))
    elif "--case" in sys.argv:
        run_case()
    else:
        for v, what in ((0, "product build"), (1, "no LDS fragment reads"), (2, "no MFMAs")):
            env = dict(os.environ)
            if v:
                env["LECO_HIP_LIB"] = os.path.join(ROOT, "tools", "_ablate", f"libleco_convablate{v}.so")
            print(f"# {what}", flush=True)
            subprocess.run([sys.executable, os.path.abspath(__file__), "--case"], env=env)
