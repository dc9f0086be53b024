#!/usr/bin/env python
"""Stand-alone check of leco_gemm_ex launch shapes at full problem sizes against torch fp32 on the same device.
    python tools/gemm_repro.py            (GPU)    |    LECO_EMU=1 python tools/gemm_repro.py   (host emulator)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from leco_amd import hip, ops  # noqa: E402

if os.environ.get("LECO_EMU"):
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    import build_emu
    hip._use_library(build_emu.build())
    dev = torch.device("cpu")
else:
    dev = torch.device("cuda:0")
bf = torch.bfloat16


def run(tile, t_rows, M, N, K, split=1, reps=3, with_tout=True, bias_res=True, seed=13):
    torch.manual_seed(seed)
    R = 12 if t_rows == 16 else 24
    a = torch.randn(M, K).to(bf).to(dev)
    w = (torch.randn(N, K) / K ** 0.5).to(bf).to(dev)
    tw = torch.zeros(32, K)
    tw[:R] = torch.randn(R, K) / K ** 0.5
    tw = tw.to(bf).to(dev)
    up = torch.zeros(N, 32)
    up[:, :R] = torch.randn(N, R) * 0.3
    up = up.to(bf).to(dev)
    bias = torch.randn(N).to(dev) if bias_res else None
    res = torch.randn(M, N).to(bf).to(dev) if bias_res else None
    ws = torch.empty(32 * 1024 * 1024, device=dev)
    T = (a.float() @ tw.float().T).to(bf)
    ref = a.float() @ w.float().T + T.float() @ up.float().T
    if bias_res:
        ref = ref + bias + res.float()
    for rep in range(reps):
        out = torch.zeros(M, N, dtype=bf, device=dev)
        tout = torch.full((M, 32), 7.0, dtype=bf, device=dev) if with_tout else None
        kw = dict(t_w=tw, t_rows=t_rows, t_out=tout) if t_rows else dict(a_ext=T, ld_aext=32)
        g = hip.gemm_args(a, w, out, m=M, n=N, k=K, w_ext=up, ext_k=32, ld_wext=32, bias=bias, residual=res, **kw)
        hip.gemm(g, ops.default_stream(), tile, split, ws)
        if dev.type == "cuda":
            torch.cuda.synchronize()
        o = out.float()
        nonfin = ~torch.isfinite(o)
        d = (o - ref).abs()
        d[nonfin] = 0
        bad = nonfin | (d > 0.05 * ref.abs().max())
        idx = bad.nonzero()
        where = ""
        if idx.numel():
            rows, cols = idx[:, 0], idx[:, 1]
            where = f" rows {rows.min().item()}..{rows.max().item()} ({rows.unique().numel()} distinct) cols {cols.min().item()}..{cols.max().item()} ({cols.unique().numel()} distinct)"
        terr = ""
        if tout is not None:
            terr = f" T_mismatch={(tout[:, :R].float() - T[:, :R].float()).abs().gt(0.02 * T.float().abs().max()).sum().item()}"
        rel = (d.norm() / ref.norm()).item()
        print(f"tile={tile} t_rows={t_rows} M={M} N={N} K={K} split={split} tout={with_tout} rep={rep}: rel={rel:.3e} "
              f"nonfinite={int(nonfin.sum())} bad={int(bad.sum())}{where}{terr}", flush=True)


if __name__ == "__main__":
    cases = [
        (3, 16, 1024, 1280, 1280), (0, 16, 1024, 1280, 1280), (3, 32, 1024, 1280, 1280), (3, 0, 1024, 1280, 1280),
        (1, 16, 1024, 1280, 1280), (3, 16, 1024, 1280, 320), (3, 16, 256, 1280, 1280), (3, 16, 1024, 3840, 1280),
        (3, 16, 128, 128, 1280), (3, 16, 1024, 1280, 256),
    ]
    for c in cases:
        run(*c)
    run(3, 16, 1024, 1280, 1280, with_tout=False)
    run(3, 16, 1024, 1280, 1280, bias_res=False)
