#!/usr/bin/env python
"""3x3 / stride-1 convolution shapes of the LECO step, implicit-GEMM kernel (gemm.hip, the launch shape the committed
tuner table picks) vs the patch-staged kernel (conv_patch.hip, tile ids 7..10 x split-K), on the GPU box:

    python tools/bench_conv.py [--arch sd15 --batch 4] [--check] [--quick]

Per shape: microseconds and TFLOP/s (algorithmic 2 M N K) of the baseline and of every patch candidate; `--check`
compares each candidate's fp32 output with the baseline's on the same operands (relative L2).  Operands are uniform
random bf16 (never zeros: MI355X_MICROARCH.md, DVFS)."""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from leco_amd import hip, ops, tune  # noqa: E402

bf = torch.bfloat16
dev = torch.device("cuda:0")

# (H = W, Cin0, Cin1 (second source of a skip concat, 0 = none), Cout, launches per UNet pass) -- SD1.5, SURVEY.md Appendix C
SD15 = [
    (64, 320, 0, 320, 7), (64, 320, 320, 320, 2), (64, 640, 320, 320, 1),
    (32, 320, 0, 640, 1), (32, 640, 0, 640, 6), (32, 640, 640, 640, 1), (32, 1280, 640, 640, 1), (32, 640, 320, 640, 1),
    (16, 640, 0, 1280, 1), (16, 1280, 0, 1280, 7), (16, 1280, 1280, 1280, 2), (16, 1280, 640, 1280, 1),
    (8, 1280, 0, 1280, 11), (8, 1280, 1280, 1280, 3),
]
SD21 = [(96, 320, 0, 320, 7), (48, 640, 0, 640, 6), (24, 1280, 0, 1280, 7), (12, 1280, 0, 1280, 11)]
SDXL = [(128, 320, 0, 320, 7), (64, 640, 0, 640, 6), (32, 1280, 0, 1280, 7), (64, 1280, 640, 640, 1)]


def timeit(fn, iters):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e30
    for _ in range(3):
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters * 1e3)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--arch", default="sd15")
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--quick", action="store_true", help="fewer split candidates")
    ap.add_argument("--blas", action="store_true",
                    help="also time the vendor library (torch.matmul -> hipBLASLt / rocBLAS) on the plain [M][K] x [K][N] bf16 GEMM "
                         "of the same size: the 'achievable' denominator for an MFMA contraction of that shape (measurement only; "
                         "never on the product path)")
    a = ap.parse_args()
    shapes = {"sd15": SD15, "sd21": SD21, "sdxl": SDXL}[a.arch]
    s = ops.default_stream()
    x = torch.randn(4096, 4096, device=dev)
    for _ in range(30):
        (x @ x).sum().item()      # clock ramp
    ws = torch.empty(32 * 1024 * 1024, dtype=torch.float32, device=dev)
    fn = hip.lib().leco_gemm_ex
    total_base = total_best = 0.0
    print(f"# {a.arch} UNet batch {a.batch}: us (TFLOP/s); base = tuner-table launch shape of gemm.hip; p<tile>/<split> = conv_patch.hip")
    for (hw, c0, c1, co, cnt) in shapes:
        B, cin = a.batch, c0 + c1
        M, K = B * hw * hw, 9 * (c0 + c1)
        x0 = (torch.rand(M, c0, device=dev) * 2 - 1).to(bf)
        x1 = (torch.rand(M, c1, device=dev) * 2 - 1).to(bf) if c1 else None
        w = ((torch.rand(co, K, device=dev) * 2 - 1) / K ** 0.5).to(bf)
        out = torch.empty(M, co, dtype=bf, device=dev)
        o32 = torch.empty(M, co, device=dev) if a.check else None
        kw = dict(a1=x1, lda1=c1, k_split=c0) if c1 else {}
        g = hip.gemm_args(x0, w, out, m=M, n=co, k=K, lda=c0, a_mode=hip.A_CONV3_S1, conv=(B, hw, hw, hw, hw), out_f32=o32, **kw)
        flops = 2.0 * M * co * K
        iters = 10 if flops > 2e10 else 20
        bt, bs_ = tune.choose(g, ws)

        def run(tile, split):
            rc = fn(C.byref(g), tile, split, ws.data_ptr(), ws.numel() * 4, s)
            assert rc == 0, hip.lib().leco_last_error()
        t_base = timeit(lambda: run(bt, bs_), iters)
        t_blas = None
        if a.blas:
            am = (torch.rand(M, K, device=dev) * 2 - 1).to(bf)
            wm = w.t().contiguous()
            t_blas = timeit(lambda: torch.matmul(am, wm), iters)
            del am, wm
        ref = o32.clone() if a.check else None
        res = {}
        tiles = -(-M // 256)
        for tile in (7, 8, 9, 10):
            bm, bn = {7: (256, 128), 8: (128, 160), 9: (128, 128), 10: (256, 160)}[tile]
            if bn == 160 and co % 160:
                continue
            blocks = -(-M // bm) * -(-co // bn)
            splits = [1] + [sp for sp in ((2, 4, 8) if a.quick else (2, 3, 4, 5, 6, 8, 10, 12, 16, 20))
                            if blocks * sp <= 768 and blocks < 256 and cin // 64 >= sp]
            for sp in splits:
                if "conv_patch" not in hip.gemm_describe(g, tile, sp, ws.data_ptr(), ws.numel() * 4):
                    continue
                res[(tile, sp)] = timeit(lambda: run(tile, sp), iters)
                if a.check:
                    torch.cuda.synchronize()
                    err = ((o32 - ref).norm() / ref.norm()).item()
                    assert err < 2e-5, (tile, sp, err)
        best = min(res, key=res.get) if res else None
        tb = res[best] if best else float("nan")
        total_base += cnt * t_base
        total_best += cnt * min(tb, t_base) if best else cnt * t_base
        top = sorted(res.items(), key=lambda kv: kv[1])[:6]
        print(f"{hw:3d}^2 {c0:4d}+{c1:<4d}->{co:4d} x{cnt:2d}  base({bt},{bs_}) {t_base:7.1f} ({flops / t_base / 1e6:5.0f})  "
              f"best p{best[0]}/{best[1]} {tb:7.1f} ({flops / tb / 1e6:5.0f})  x{t_base / tb:4.2f} | "
              + "  ".join(f"p{t}/{sp} {us:.1f}" for (t, sp), us in top)
              + (f"  | vendor GEMM {t_blas:.1f} us ({flops / t_blas / 1e6:.0f} TF/s)" if t_blas else ""), flush=True)
    print(f"# conv time per UNet pass: base {total_base / 1e3:.3f} ms -> best-of {total_best / 1e3:.3f} ms")


if __name__ == "__main__":
    main()
