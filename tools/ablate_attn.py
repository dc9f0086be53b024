"""Attention-forward ablation on the GPU box.  argv: integers n build -DLECO_ATTN_ABLATE=n, `occN` builds
-DLECO_ATTN_OCC40=N (occupancy floor of the d <= 40 kernels), a path names a prebuilt library.  Variants are cached under tools/_ablate
so they can be built in the (GPU-less) dev container and travel with the snapshot (`--build-only`)."""
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from leco_amd import build as B, hip, ops  # noqa: E402

bf = torch.bfloat16
dev = torch.device("cuda:0")


def build_variant(v):
    if os.path.exists(str(v)):
        return str(v)
    d = os.path.join(ROOT, "tools", "_ablate")
    os.makedirs(d, exist_ok=True)
    out = os.path.join(d, f"libleco_attn_{v}.so")
    if os.path.exists(out) and os.path.getmtime(out) > os.path.getmtime(os.path.join(B.CSRC, "attention.hip")):
        return out
    if str(v).startswith("occ64_"):
        define = f"-DLECO_ATTN_OCC64={str(v)[6:]}"
    elif str(v).startswith("occ"):
        define = f"-DLECO_ATTN_OCC40={str(v)[3:]}"
    else:
        define = f"-DLECO_ATTN_ABLATE={v}"
    # only attention.hip is rebuilt with the switch; the other objects are the product build's
    B.build()
    src = os.path.join(B.CSRC, "attention.hip")
    objs = [o for o in open(os.path.join(B.OBJ, "link.stamp")).read().split() if "/attention.hip." not in o]
    obj = os.path.join(d, f"attention_{v}.o")
    subprocess.run([B.HIPCC, *B.FLAGS, define, "-x", "hip", "-c", src, "-o", obj], check=True)
    subprocess.run([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, *objs, obj], check=True)
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


VARIANTS = [a for a in sys.argv[1:] if a != "--build-only"] or ["0", "1", "2", "3", "4"]
if "--build-only" in sys.argv:
    for v in VARIANTS:
        if v != "0":
            print(build_variant(v))
    sys.exit(0)
for v in VARIANTS:
    hip._use_library(build_variant(v) if v != "0" else hip.LIB_PATH)
    for (Bq, H, Sq, Skv, D) in [(4, 8, 4096, 4096, 40), (12, 8, 4096, 4096, 40), (4, 8, 4096, 77, 40), (4, 8, 1024, 1024, 80),
                                (4, 8, 256, 256, 160), (4, 5, 9216, 9216, 64), (2, 10, 4096, 4096, 64), (2, 20, 1024, 1024, 64)]:
        C = H * D
        q = torch.randn(Bq, Sq, C, device=dev).to(bf); k = torch.randn(Bq, Skv, C, device=dev).to(bf)
        vv = torch.randn(Bq, Skv, C, device=dev).to(bf); o = torch.empty_like(q); lse = torch.empty(Bq, H, Sq, device=dev)
        op = ops.attention_fwd(q.data_ptr(), C, Sq * C, k.data_ptr(), C, Skv * C, vv.data_ptr(), C, Skv * C, o.data_ptr(), C,
                               Sq * C, lse, Bq, H, Sq, Skv, D, D ** -0.5)
        t = timeit(lambda: op.run(None))
        print(f"variant={os.path.basename(str(v)):22s} B={Bq} S={Sq}x{Skv} D={D}: {t:8.1f} us ({4.0*Bq*H*Sq*Skv*D/t/1e6:7.1f} TF/s nominal)", flush=True)
