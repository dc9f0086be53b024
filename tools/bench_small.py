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

# self-attention forward: register-staged (LECO_ATTN_DMA=0) vs LDS-DMA staged kernels, q|k|v fused layout as the planner's
print("attention fwd  B  H     S   d   staged us   dma us")
for (B, H, S, D) in [(4, 8, 4096, 40), (2, 8, 4096, 40), (12, 8, 4096, 40), (4, 8, 1024, 80), (12, 8, 1024, 80), (4, 5, 9216, 64),
                     (4, 10, 2304, 64), (2, 10, 4096, 64), (2, 20, 1024, 64), (4, 20, 576, 64)]:
    C = H * D
    qkv = torch.randn(B, S, 3 * C, device=dev).to(bf)
    o = torch.zeros(B, S, C, device=dev, dtype=bf); lse = torch.zeros(B, H, S, device=dev)
    p0 = qkv.data_ptr()
    op = ops.attention_fwd(p0, 3 * C, S * 3 * C, p0 + 2 * C, 3 * C, S * 3 * C, p0 + 4 * C, 3 * C, S * 3 * C, o.data_ptr(), C, S * C, lse, B, H, S, S, D, D ** -0.5)
    ts = []
    for mode in ("0", "1"):
        os.environ["LECO_ATTN_DMA"] = mode
        ts.append(timeit(op, 50))
    fl = 4.0 * B * H * S * S * D
    print(f"              {B:2d} {H:2d} {S:5d} {D:3d}   {ts[0]:8.1f}  {ts[1]:8.1f}   ({fl / ts[0] / 1e6:.0f} -> {fl / ts[1] / 1e6:.0f} TFLOP/s)")
