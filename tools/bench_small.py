"""Times a few small launches in isolation (HIP events over back-to-back repeats): conv_in / conv_out at the level-0 shape.
    python tools/bench_small.py          (under `rocprofv3 --kernel-trace --stats` the kernel names show which path ran)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from leco_amd import ops

dev = torch.device("cuda:0"); bf = torch.bfloat16


def timeit(op, n=200):
    for _ in range(20): op.run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): op.run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for B in (2, 4, 12):
    H = W = 64; C = 320
    x = torch.randn(B, H, W, C, device=dev).to(bf); w = (torch.randn(4, 3, 3, C, device=dev) * 0.05).to(bf); b4 = torch.randn(4, device=dev)
    y = torch.zeros(B, 4, H, W, device=dev)
    print(f"conv_out B={B}: {timeit(ops.conv_out(x, w, b4, y, B, H, W, C, 4)):.1f} us")
    xi = torch.randn(B, 4, H, W, device=dev).to(bf); wi = torch.randn(4, 3, 3, C, device=dev) * 0.2; bi = torch.randn(C, device=dev)
    yi = torch.zeros(B, H, W, C, device=dev, dtype=bf)
    print(f"conv_in  B={B}: {timeit(ops.conv_in(xi, wi, bi, yi, B, H, W, 4, C)):.1f} us")
