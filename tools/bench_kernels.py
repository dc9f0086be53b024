"""Per-kernel microbenchmarks at the SD1.5 / 512^2 / B=4 shapes.
Run on the GPU box: python tools/bench_kernels.py [--json out.json]"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from leco_amd import hip, ops  # noqa: E402

bf = torch.bfloat16
dev = torch.device("cuda:0")


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    res = []
    s = ops.default_stream()

    def gemm_case(name, M, N, K, conv=None, tile=0, ext=0):
        x = torch.randn(M if conv is None else conv[0] * conv[3] * conv[4], K if conv is None else K // 9, device=dev).to(bf)
        w = (torch.randn(N, K, device=dev) / K ** 0.5).to(bf)
        out = torch.empty(M, N, dtype=bf, device=dev)
        kw = {}
        if conv is not None:
            kw = dict(a_mode=hip.A_CONV3_S1, conv=conv, lda=K // 9)
        if ext:
            kw.update(a_ext=torch.randn(M, ext, device=dev).to(bf), w_ext=torch.randn(N, ext, device=dev).to(bf), ext_k=ext)
        g = hip.gemm_args(x, w, out, m=M, n=N, k=K, **kw)
        t = timeit(lambda: hip.gemm(g, s, tile))
        tf = 2.0 * M * N * K / t / 1e12
        res.append(dict(kernel=name, M=M, N=N, K=K, tile=tile, ms=t * 1e3, tflops=tf))
        print(f"{name:28s} M={M:6d} N={N:6d} K={K:6d} tile={tile} {t*1e3:8.3f} ms {tf:8.1f} TF/s", flush=True)

    B = 4
    for tile in (0, 1, 2, 3):
        gemm_case("conv3x3 L0 320->320", B * 64 * 64, 320, 2880, conv=(B, 64, 64, 64, 64), tile=tile)
    gemm_case("conv3x3 L1 640->640", B * 32 * 32, 640, 5760, conv=(B, 32, 32, 32, 32))
    gemm_case("conv3x3 L1 640->640", B * 32 * 32, 640, 5760, conv=(B, 32, 32, 32, 32), tile=1)
    gemm_case("conv3x3 L2 1280->1280", B * 16 * 16, 1280, 11520, conv=(B, 16, 16, 16, 16))
    gemm_case("conv3x3 L2 1280->1280", B * 16 * 16, 1280, 11520, conv=(B, 16, 16, 16, 16), tile=1)
    gemm_case("conv3x3 L3 1280->1280", B * 8 * 8, 1280, 11520, conv=(B, 8, 8, 8, 8))
    gemm_case("conv3x3 up 2560->1280 L2", B * 16 * 16, 1280, 23040, conv=(B, 16, 16, 16, 16))
    gemm_case("linear L0 320x320", 16384, 320, 320)
    gemm_case("linear L0 320x320 +lora", 16384, 320, 320, ext=32)
    gemm_case("linear L0 qkv 960x320", 16384, 960, 320)
    gemm_case("ff1 L0 2560x320", 16384, 2560, 320)
    gemm_case("ff2 L0 320x1280", 16384, 320, 1280)
    gemm_case("linear L1 640x640", 4096, 640, 640)
    gemm_case("ff1 L1 5120x640", 4096, 5120, 640)
    gemm_case("ff1 L2 10240x1280", 1024, 10240, 1280)
    gemm_case("ff2 L2 1280x5120", 1024, 1280, 5120)
    gemm_case("big 8192^3/8", 8192, 8192, 1024, tile=1)
    gemm_case("lora_down L0 r32", 16384, 32, 320)

    def attn_case(Bq, H, Sq, Skv, D):
        Cq = H * D
        q = torch.randn(Bq, Sq, Cq, device=dev).to(bf); k = torch.randn(Bq, Skv, Cq, device=dev).to(bf)
        v = torch.randn(Bq, Skv, Cq, device=dev).to(bf); o = torch.empty_like(q)
        lse = torch.empty(Bq, H, Sq, device=dev)
        op = ops.attention_fwd(q.data_ptr(), Cq, Sq * Cq, k.data_ptr(), Cq, Skv * Cq, v.data_ptr(), Cq, Skv * Cq,
                               o.data_ptr(), Cq, Sq * Cq, lse, Bq, H, Sq, Skv, D, D ** -0.5)
        t = timeit(lambda: op.run(s))
        tf = 4.0 * Bq * H * Sq * Skv * D / t / 1e12
        res.append(dict(kernel="attn_fwd", B=Bq, H=H, Sq=Sq, Skv=Skv, D=D, ms=t * 1e3, tflops=tf))
        print(f"attn_fwd B{Bq} H{H} Sq{Sq} Skv{Skv} D{D}: {t*1e3:8.3f} ms {tf:8.1f} TF/s", flush=True)
        do = torch.randn_like(q); dq = torch.empty_like(q); dk = torch.empty_like(k); dv = torch.empty_like(v)
        delta = torch.empty(Bq, H, Sq, device=dev)
        opb = ops.attention_bwd(q.data_ptr(), Cq, Sq * Cq, k.data_ptr(), Cq, Skv * Cq, v.data_ptr(), Cq, Skv * Cq,
                                o.data_ptr(), Cq, Sq * Cq, do.data_ptr(), Cq, Sq * Cq, lse, delta, dq.data_ptr(), Cq,
                                Sq * Cq, dk.data_ptr(), Cq, Skv * Cq, dv.data_ptr(), Cq, Skv * Cq, Bq, H, Sq, Skv, D,
                                D ** -0.5)
        t = timeit(lambda: opb.run(s), iters=5)
        tf = 10.0 * Bq * H * Sq * Skv * D / t / 1e12
        res.append(dict(kernel="attn_bwd", B=Bq, H=H, Sq=Sq, Skv=Skv, D=D, ms=t * 1e3, tflops=tf))
        print(f"attn_bwd B{Bq} H{H} Sq{Sq} Skv{Skv} D{D}: {t*1e3:8.3f} ms {tf:8.1f} TF/s (5-matmul count)", flush=True)

    attn_case(4, 8, 4096, 4096, 40)
    attn_case(4, 8, 1024, 1024, 80)
    attn_case(4, 8, 256, 256, 160)
    attn_case(4, 8, 4096, 77, 40)
    attn_case(4, 5, 9216, 9216, 64)

    # HBM-bound kernels
    M, Cc = 16384, 320
    x = torch.randn(M, Cc, device=dev).to(bf); y = torch.empty_like(x)
    gamma = torch.ones(Cc, device=dev); beta = torch.zeros(Cc, device=dev)
    stats = torch.zeros(4 * 32 * 2 * 257, device=dev)
    op = ops.groupnorm_fwd(x, Cc, None, 0, 0, gamma, beta, 4, 4096, Cc, 32, 1e-5, 1, stats, y, Cc)
    t = timeit(lambda: op.run(s))
    print(f"groupnorm+silu L0: {t*1e6:8.1f} us  {3*M*Cc*2/t/1e9:8.1f} GB/s (2 reads + 1 write)", flush=True)
    res.append(dict(kernel="groupnorm_fwd", ms=t * 1e3, gbps=3 * M * Cc * 2 / t / 1e9))
    mean = torch.empty(M, device=dev); rstd = torch.empty(M, device=dev)
    op = ops.layernorm_fwd(x, Cc, gamma, beta, 1e-5, M, Cc, y, Cc, mean, rstd)
    t = timeit(lambda: op.run(s))
    print(f"layernorm L0: {t*1e6:8.1f} us  {2*M*Cc*2/t/1e9:8.1f} GB/s", flush=True)
    res.append(dict(kernel="layernorm_fwd", ms=t * 1e3, gbps=2 * M * Cc * 2 / t / 1e9))
    u = torch.randn(M, 2560, device=dev).to(bf); yy = torch.empty(M, 1280, dtype=bf, device=dev)
    op = ops.geglu_fwd(u, 2560, yy, 1280, M, 1280)
    t = timeit(lambda: op.run(s))
    print(f"geglu L0: {t*1e6:8.1f} us  {3*M*1280*2/t/1e9:8.1f} GB/s", flush=True)
    res.append(dict(kernel="geglu_fwd", ms=t * 1e3, gbps=3 * M * 1280 * 2 / t / 1e9))

    # hipGraph capture smoke: 20 small GEMMs captured on a side stream, replayed
    lib = hip.lib()
    for nm, at in [("leco_graph_begin_capture", [C.c_void_p]), ("leco_graph_end_capture", [C.c_void_p, C.POINTER(C.c_void_p)]),
                   ("leco_graph_launch", [C.c_void_p, C.c_void_p]), ("leco_graph_destroy", [C.c_void_p])]:
        getattr(lib, nm).argtypes = at
        getattr(lib, nm).restype = C.c_int
    st = torch.cuda.Stream()
    a_ = torch.randn(256, 320, device=dev).to(bf); w_ = torch.randn(320, 320, device=dev).to(bf)
    o_ = torch.zeros(256, 320, dtype=bf, device=dev)
    g = hip.gemm_args(a_, w_, o_, m=256, n=320, k=320)
    torch.cuda.synchronize()
    with torch.cuda.stream(st):
        sp = st.cuda_stream
        hip.check(lib.leco_graph_begin_capture(sp), "begin")
        for _ in range(20):
            hip.gemm(g, sp)
        gh = C.c_void_p()
        hip.check(lib.leco_graph_end_capture(sp, C.byref(gh)), "end")
        o_.zero_()
        hip.check(lib.leco_graph_launch(gh, sp), "launch")
        st.synchronize()
        ok = torch.allclose(o_.float(), (a_.float() @ w_.float().T).to(bf).float(), rtol=2e-2, atol=2e-2)
        t0 = time.perf_counter()
        for _ in range(50):
            lib.leco_graph_launch(gh, sp)
        st.synchronize()
        tg = (time.perf_counter() - t0) / 50
        t0 = time.perf_counter()
        for _ in range(50):
            for _ in range(20):
                hip.gemm(g, sp)
        st.synchronize()
        te = (time.perf_counter() - t0) / 50
    print(f"hipGraph: correct={ok} 20-gemm graph replay {tg*1e6:.1f} us vs eager ctypes {te*1e6:.1f} us", flush=True)
    res.append(dict(kernel="graph20", graph_us=tg * 1e6, eager_us=te * 1e6, ok=bool(ok)))
    if a.json:
        with open(a.json, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
