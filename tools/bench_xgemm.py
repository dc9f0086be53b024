"""leco_xgemm (csrc/xgemm.hip) against the LDS-ring GEMM (leco_gemm_ex, tuned launch shape) on the transformer-block shapes of
the denoising pass, LoRA rank 4 fused, with residual: HIP events around back-to-back launches, both captured into ONE hipGraph
each (no eager launch floor).  LECO_XGEMM_VAR selects the kernel variant.
    python tools/bench_xgemm.py"""
import ctypes as C
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from leco_amd import hip, ops          # noqa: E402
from leco_amd.unet import _graph_api   # noqa: E402

bf = torch.bfloat16


def graph_us(chain, reps=20):
    lib = _graph_api()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    hip.check(lib.leco_graph_begin_capture(side.cuda_stream), "begin")
    ops.run_plan(chain, side.cuda_stream)
    g = C.c_void_p()
    hip.check(lib.leco_graph_end_capture(side.cuda_stream, C.byref(g)), "end")
    cur = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        lib.leco_graph_launch(g, cur)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        lib.leco_graph_launch(g, cur)
    e1.record()
    e1.synchronize()
    lib.leco_graph_destroy(g)
    return e0.elapsed_time(e1) / reps / len(chain) * 1e3


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    x = torch.randn(4096, 4096, device=dev)
    for _ in range(20):
        (x @ x).sum().item()      # clock ramp
    print(f"# LECO_XGEMM_VAR={os.environ.get('LECO_XGEMM_VAR', '(default 1)')}")
    for m, n, k in ((1024, 1280, 1280), (4096, 640, 640), (1024, 3840, 1280), (4096, 1920, 640), (256, 1280, 1280),
                    (3072, 1280, 1280), (12288, 640, 640)):
        a = (torch.randn(m, k, device=dev) * 0.5).to(bf)
        w = (torch.randn(n, k, device=dev) / math.sqrt(k)).to(bf)
        bias = torch.randn(n, device=dev) * 0.1
        r = torch.randn(m, n, device=dev).to(bf)
        dn = torch.zeros(32, k, device=dev, dtype=bf)
        dn[:4] = (torch.randn(4, k, device=dev) / math.sqrt(k)).to(bf)
        up = torch.zeros(n, 32, device=dev, dtype=bf)
        up[:, :4] = (torch.randn(n, 4, device=dev) * 0.2).to(bf)
        wp = hip.pack_fragments(w)
        cx, cg = torch.zeros(m, n, dtype=bf, device=dev), torch.zeros(m, n, dtype=bf, device=dev)
        T = torch.zeros(m, 32, dtype=bf, device=dev)
        lin = hip.xlin(wp, bias, dn, up, 16, packed=True)
        xop = ops.xgemm(a.data_ptr(), k, lin, cx.data_ptr(), n, m, n, k, residual=r.data_ptr(), ldr=n, keep=(a, wp, cx, r))
        ws = torch.empty(32 << 20, dtype=torch.float32, device=dev)
        g = hip.gemm_args(a, w, cg, m=m, n=n, k=k, bias=bias, residual=r, ldr=n, w_ext=up, ext_k=32, ld_wext=32, t_w=dn, t_rows=16,
                          t_out=T, ld_tout=32)
        gop = ops.gemm(g, keep=(a, w, cg, r, T), ws=ws)       # launch shape: the tuned table / the C heuristic
        xop.run(); gop.run()
        torch.cuda.synchronize()
        err = ((cx.float() - cg.float()).norm() / cg.float().norm()).item()
        tx, tg = graph_us([xop] * 16), graph_us([gop] * 16)
        fl = 2.0 * m * n * (k + 32)
        print(f"M={m:6d} N={n:5d} K={k:5d}: xgemm {tx:6.1f} us ({fl / tx / 1e6:6.1f} TFLOP/s)   ring gemm {tg:6.1f} us "
              f"({fl / tg / 1e6:6.1f} TFLOP/s)   x{tg / tx:.2f}   rel diff {err:.2e}")


if __name__ == "__main__":
    main()
