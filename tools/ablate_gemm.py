"""GEMM ablation on the GPU box: builds libleco_hip variants with -DLECO_GEMM_ABLATE=1 (no MFMA) / 2 (no DMA)
and times a few shapes with each, to see which pipe bounds a K step."""
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from leco_amd import build as B, hip  # noqa: E402

bf = torch.bfloat16
dev = torch.device("cuda:0")


def build_variant(v):
    d = os.path.join(ROOT, "tools", "_ablate")   # prebuilt variants travel to the GPU box with the snapshot
    os.makedirs(d, exist_ok=True)
    out = os.path.join(d, f"libleco_ablate{v}.so")
    if os.path.exists(out) and os.path.getmtime(out) > os.path.getmtime(os.path.join(B.CSRC, "gemm.hip")):
        return out
    srcs = [os.path.join(B.CSRC, f) for f in sorted(os.listdir(B.CSRC)) if f.endswith((".hip", ".cpp"))]
    cmd = [B.HIPCC, *B.FLAGS, f"-DLECO_GEMM_ABLATE={v}", "-shared", "-x", "hip", *srcs, "-o", out]
    subprocess.run(cmd, check=True)
    return out


_WARM = [False]


def timeit(fn, iters=20):
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


def plain_cases():
    """the short-K / small-M plain projections of one SD1.5 B=4 pass (tools/plan_profile.py), with their LoRA tile"""
    out = []
    for name, M, N, K, geglu, res in [("L0 out 320x320", 16384, 320, 320, 0, 1), ("L0 qkv", 16384, 960, 320, 0, 0),
                                      ("L0 geglu", 16384, 2560, 320, 1, 0), ("L0 ff2", 16384, 320, 1280, 0, 1),
                                      ("L1 out 640x640", 4096, 640, 640, 0, 1), ("L1 geglu", 4096, 5120, 640, 1, 0),
                                      ("L2 out 1280", 1024, 1280, 1280, 0, 1), ("L2 ff2 K5120", 1024, 1280, 5120, 0, 1),
                                      ("L2 geglu", 1024, 10240, 1280, 1, 0)]:
        out.append((name, M, N, K, geglu, res))
    return out


CASES = [("conv L0 320", 16384, 320, 2880, (4, 64, 64, 64, 64), 0), ("conv L1 640", 4096, 640, 5760, (4, 32, 32, 32, 32), 0),
         ("conv L1 640 t4", 4096, 640, 5760, (4, 32, 32, 32, 32), 4),
         ("conv L2 1280", 1024, 1280, 11520, (4, 16, 16, 16, 16), 0), ("conv L2 1280 t4", 1024, 1280, 11520, (4, 16, 16, 16, 16), 4),
         ("conv L3 1280", 256, 1280, 11520, (4, 8, 8, 8, 8), 0), ("conv L3 1280 t4", 256, 1280, 11520, (4, 8, 8, 8, 8), 4),
         ("conv up L2 2560", 1024, 1280, 23040, (4, 16, 16, 16, 16), 0), ("conv up L2 2560 t4", 1024, 1280, 23040, (4, 16, 16, 16, 16), 4),
         ("ff1 L0", 16384, 2560, 320, None, 0), ("ff1 L0 t4", 16384, 2560, 320, None, 4), ("big", 8192, 8192, 1024, None, 1), ("big t4", 8192, 8192, 1024, None, 4),
         ("lin L2 1280", 1024, 1280, 1280, None, 0)]
VARIANTS = [int(a) for a in sys.argv[1:] if a.lstrip("-").isdigit()] or [0, 1, 2, 4, 8]
ONLY = os.environ.get("ABLATE_CASES")
if ONLY:
    CASES = [c for c in CASES if c[0] in ONLY.split(",")]
if "--build-only" in sys.argv:
    for v in VARIANTS:
        if v:
            print(build_variant(v))
    sys.exit(0)
ws = torch.empty(32 * 1024 * 1024, device=dev)
PLAIN = "--plain" in sys.argv

for v in VARIANTS:
    hip._use_library(build_variant(v) if v else hip.LIB_PATH)
    if PLAIN:
        for name, M, N, K, geglu, res in plain_cases():
            x = torch.randn(M, K, device=dev).to(bf)
            w = (torch.randn(N, K, device=dev) / K ** 0.5).to(bf)
            nout = N // 2 if geglu else N
            out = torch.empty(M, nout, dtype=bf, device=dev)
            tw = torch.zeros(32, K, dtype=bf, device=dev); up = torch.zeros(N, 32, dtype=bf, device=dev)
            bias = torch.zeros(N, device=dev); r = torch.zeros(M, N, dtype=bf, device=dev) if res else None
            for lora in (1, 0):
                kw = dict(w_ext=up, ext_k=32, t_w=tw, t_rows=16) if lora else {}
                g = hip.gemm_args(x, w, out, m=M, n=N, k=K, bias=bias, residual=r, act=2 if geglu else 0, ldc=nout, **kw)
                t = timeit(lambda: hip.gemm(g, None, 0, 0, ws))
                print(f"ablate={v:3d} {name:16s} lora={lora} {t:8.1f} us  ({2.0*M*N*K/t/1e6:7.1f} TF/s nominal)", flush=True)
        continue
    for name, M, N, K, conv, tile in CASES:
        x = torch.randn(M if conv is None else conv[0] * conv[3] * conv[4], K if conv is None else K // 9, device=dev).to(bf)
        w = (torch.randn(N, K, device=dev) / K ** 0.5).to(bf)
        out = torch.empty(M, N, dtype=bf, device=dev)
        kw = dict(a_mode=hip.A_CONV3_S1, conv=conv, lda=K // 9) if conv else {}
        g = hip.gemm_args(x, w, out, m=M, n=N, k=K, **kw)
        t = timeit(lambda: hip.gemm(g, None, tile, 0, ws))
        print(f"ablate={v} {name:12s} {t:8.1f} us  ({2.0*M*N*K/t/1e6:7.1f} TF/s nominal)", flush=True)
