"""The plain (Linear / 1x1) GEMM launches of one SD1.5 512^2 denoising pass (UNet batch 4) in isolation: fused LoRA rank-4
down-projection + K-extension, bias, residual or fused GEGLU as the plan has them, launch shape from the tuner table; each op
captured 16x into ONE hipGraph (no eager launch floor).  us per launch, TFLOP/s, and the L2 -> LDS operand traffic the tiling
implies per second (the quantity that bounds these kernels: DESIGN section 8.000).  Extra argv: other libleco_hip builds.
    python tools/bench_gemm_plain.py [lib.so ...]"""
import math
import os
import re
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from leco_amd import hip, ops  # noqa: E402
from tools.libswitch import graph_us, use_lib  # noqa: E402

bf = torch.bfloat16
dev = torch.device("cuda:0")
# (M, N, K, residual, geglu, launches per pass)
SHAPES = [(4096, 5120, 640, 0, 1, 5), (1024, 10240, 1280, 0, 1, 5), (1024, 1280, 1280, 1, 0, 15), (1024, 1280, 5120, 1, 0, 5),
          (4096, 640, 640, 1, 0, 15), (4096, 640, 2560, 1, 0, 5), (1024, 1280, 1280, 0, 0, 10), (4096, 640, 640, 0, 0, 10),
          (4096, 1920, 640, 0, 0, 5), (1024, 3840, 1280, 0, 0, 5), (1024, 1280, 2560, 0, 0, 2), (256, 1280, 1280, 1, 0, 3),
          (256, 10240, 1280, 0, 1, 1), (256, 1280, 5120, 1, 0, 1)]
# training-plan / frozen-pass shapes (UNet batch 4 level 0, batch 12)
EXTRA = [(16384, 320, 320, 1, 0, 0), (16384, 960, 320, 0, 0, 0), (16384, 2560, 320, 0, 0, 0), (16384, 320, 1280, 1, 0, 0),
         (12288, 5120, 640, 0, 1, 0), (3072, 10240, 1280, 0, 1, 0), (3072, 1280, 1280, 1, 0, 0), (12288, 640, 640, 1, 0, 0)]


def build(m, n, k, res, geglu):
    a = (torch.randn(m, k, device=dev) * 0.5).to(bf)
    w = (torch.randn(n, k, device=dev) / math.sqrt(k)).to(bf)
    bias = torch.randn(n, device=dev) * 0.1
    r = torch.randn(m, n, device=dev).to(bf) if res else None
    dn = torch.zeros(32, k, device=dev, dtype=bf)
    dn[:4] = (torch.randn(4, k, device=dev) / math.sqrt(k)).to(bf)
    up = torch.zeros(n, 32, device=dev, dtype=bf)
    up[:, :4] = (torch.randn(n, 4, device=dev) * 0.2).to(bf)
    c = torch.zeros(m, n // 2 if geglu else n, dtype=bf, device=dev)
    ws = torch.empty(32 << 20, dtype=torch.float32, device=dev)
    kw = dict(m=m, n=n, k=k, bias=bias, w_ext=up, ext_k=32, ld_wext=32, t_w=dn, t_rows=16)
    if res:
        kw.update(residual=r, ldr=n)
    if geglu:
        kw.update(act=hip.ACT_GEGLU, ldc=n // 2)
    g = hip.gemm_args(a, w, c, **kw)
    return ops.gemm(g, keep=(a, w, c, r, dn, up, bias), ws=ws), g, ws, c


def traffic(desc, m, n, k):
    """bytes DMA'd L2 -> LDS by the launch `desc` names (tile rows incl. the 16 stacked lora_down rows)"""
    mt = re.search(r"gemm_kernel<(\d+), (\d+), \w+, \d+, \d+, (\d+)", desc)
    if not mt:
        return 0.0
    bm, bn, tf = int(mt.group(1)), int(mt.group(2)), int(mt.group(3))
    tiles = -(-m // bm) * -(-n // bn)
    return tiles * (bm + bn + 16 * tf) * k * 2.0


def main():
    x = torch.randn(4096, 4096, device=dev)
    for _ in range(20):
        (x @ x).sum().item()      # clock ramp
    libs = [hip.LIB_PATH] + sys.argv[1:]
    print("# libs: " + "  |  ".join(os.path.relpath(l, ROOT) for l in libs))
    tot = [0.0] * len(libs)
    for (m, n, k, res, geglu, cnt) in SHAPES + EXTRA:
        cols, ref = [], None
        for li, lib in enumerate(libs):
            use_lib(lib)
            torch.manual_seed(1)
            op, g, ws, c = build(m, n, k, res, geglu)
            desc = hip.gemm_describe(g, op.args[1], op.args[2], ws.data_ptr(), ws.numel() * 4)
            op.run()
            torch.cuda.synchronize()
            if ref is None:
                ref = c.float().clone()
                err = 0.0
            else:
                err = ((c.float() - ref).norm() / ref.norm()).item()
            t = graph_us([op] * 16)
            tot[li] += t * cnt
            fl = 2.0 * m * n * (k + 32)
            tb = traffic(desc, m, n, k) / t / 1e6
            cols.append(f"{t:6.1f} us {fl / t / 1e6:6.0f} TF/s" + (f" L2>LDS {tb:4.1f} TB/s" if tb else "") + (f" d={err:.0e}" if li else ""))
            last = desc
        print(f"M={m:5d} N={n:5d} K={k:4d}{' +res' if res else ''}{' geglu' if geglu else ''} x{cnt:<2d} " + "  |  ".join(cols)
              + f"   [{last.split(' grid')[0]}]")
    print("# per denoising pass (sum of launches x count): " + "  |  ".join(f"{t:7.1f} us" for t in tot))


if __name__ == "__main__":
    main()
