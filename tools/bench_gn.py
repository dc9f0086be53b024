"""GroupNorm micro-benchmark over the UNet's shapes (SD1.5 @512^2, B = 4 and 12): us per call, effective GB/s
(read + write once).  Optional argv: paths of alternative libleco_hip builds to compare."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from leco_amd import hip, ops  # noqa: E402

bf = torch.bfloat16
dev = torch.device("cuda:0")
SHAPES = [(4096, 320, 0), (4096, 320, 320), (1024, 320, 0), (1024, 640, 0), (1024, 640, 640), (1024, 640, 320),
          (256, 640, 0), (256, 1280, 0), (256, 1280, 1280), (256, 1280, 640), (64, 1280, 0), (64, 1280, 1280)]


_WARM = [False]


def timeit(fn, iters=30):
    if not _WARM[0]:   # the first measurement of a process otherwise runs at ramping clocks (~10-15% slow)
        x = torch.randn(4096, 4096, device=dev)
        t_end = __import__("time").perf_counter() + 0.5
        while __import__("time").perf_counter() < t_end:
            (x @ x).sum().item()
        _WARM[0] = True
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for lib in [hip.LIB_PATH] + sys.argv[1:]:
    hip._use_library(lib)
    tot = {4: 0.0, 12: 0.0}
    for B in (4, 12):
        for hw, c0, c1 in SHAPES:
            C = c0 + c1
            x0 = torch.randn(B * hw, c0, device=dev).to(bf)
            x1 = torch.randn(B * hw, c1, device=dev).to(bf) if c1 else None
            gamma = torch.randn(C, device=dev); beta = torch.randn(C, device=dev)
            stats = torch.zeros(B * 32 * 2 * 257, device=dev); y = torch.empty(B * hw, C, dtype=bf, device=dev)
            op = ops.groupnorm_fwd(x0, c0, x1, c1, c0, gamma, beta, B, hw, C, 32, 1e-5, 1, stats, y, C)
            t = timeit(op.run)
            tot[B] += t
            print(f"{os.path.basename(lib):24s} B={B:2d} hw={hw:5d} C={c0}+{c1:<5d} {t:7.1f} us  "
                  f"{4.0 * B * hw * C / t / 1e3:7.0f} GB/s", flush=True)
    print(f"{os.path.basename(lib):24s} sum B=4 {tot[4]:.1f} us, B=12 {tot[12]:.1f} us")
