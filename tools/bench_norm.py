"""GroupNorm / LayerNorm launches of the SD1.5 512^2 passes (UNet batch 4 = denoising, 12 = batched frozen) in isolation, each
captured 16x into ONE hipGraph (no eager launch floor; the ~1.5 us node-to-node floor is part of every figure): us per launch and
GB/s (read + write once) against the 8 TB/s roof.  Extra argv: alternative libleco_hip builds to compare (e.g. the previous
round's, tools/_scratch/libs/libleco_hip_r05.so).
    python tools/bench_norm.py [lib.so ...]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from leco_amd import hip, ops  # noqa: E402
from tools.libswitch import graph_us, use_lib  # noqa: E402

bf = torch.bfloat16
dev = torch.device("cuda:0")
GN = [(64, 1280, 0), (64, 1280, 1280), (256, 1280, 0), (256, 640, 0), (256, 1280, 1280), (256, 1280, 640), (1024, 640, 0),
      (1024, 320, 0), (1024, 640, 640), (1024, 640, 320), (4096, 320, 0), (4096, 320, 320)]
LN = [(16384, 320), (4096, 640), (1024, 1280), (49152, 320), (12288, 640), (3072, 1280)]


def main():
    x = torch.randn(4096, 4096, device=dev)
    for _ in range(20):
        (x @ x).sum().item()      # clock ramp
    libs = [hip.LIB_PATH] + sys.argv[1:]
    rows = {}
    for lib in libs:
        use_lib(lib)
        for B in (4, 12):
            for hw, c0, c1 in GN:
                C = c0 + c1
                x0 = torch.randn(B * hw, c0, device=dev).to(bf)
                x1 = torch.randn(B * hw, c1, device=dev).to(bf) if c1 else None
                gamma, beta = torch.randn(C, device=dev), torch.randn(C, device=dev)
                stats = torch.zeros(B * 32 * 2 * 257, device=dev)
                y = torch.empty(B * hw, C, dtype=bf, device=dev)
                op = ops.groupnorm_fwd(x0, c0, x1, c1, c0, gamma, beta, B, hw, C, 32, 1e-5, 1, stats, y, C)
                op.run()
                t = graph_us([op] * 16)
                rows.setdefault(f"groupnorm_fwd B={B:2d} HW={hw:5d} C={c0}+{c1}", []).append((t, 4.0 * B * hw * C / t / 1e3))
                if hw >= 1024:       # the producer-statistics form the plans use for the large slices
                    atom = (C // 32) if (C // 32) <= 10 else 10
                    while (C // 32) % atom or (c1 and c0 % atom):
                        atom -= 1
                    cs0 = torch.zeros(B, c0 // atom, 2, device=dev)
                    cs1 = torch.zeros(B, max(c1, atom) // atom, 2, device=dev)
                    ops.Op("leco_colstats", (x0.data_ptr(), c0, cs0.data_ptr(), atom, B, hw, c0)).run()
                    if c1:
                        ops.Op("leco_colstats", (x1.data_ptr(), c1, cs1.data_ptr(), atom, B, hw, c1)).run()
                    op2 = ops.Op("leco_groupnorm_apply_stats",
                                 (x0.data_ptr(), c0, x1.data_ptr() if c1 else None, c1, c0 if c1 else 0, cs0.data_ptr(),
                                  cs1.data_ptr() if c1 else None, atom, gamma.data_ptr(), beta.data_ptr(), B, hw, C, 32, 1e-5, 1,
                                  stats.data_ptr(), y.data_ptr(), C), keep=(x0, x1, cs0, cs1, gamma, beta, stats, y))
                    op2.run()
                    t = graph_us([op2] * 16)
                    rows.setdefault(f"gn_apply_stats B={B:2d} HW={hw:5d} C={c0}+{c1}", []).append((t, 4.0 * B * hw * C / t / 1e3))
        for M, C in LN:
            xx = torch.randn(M, C, device=dev).to(bf)
            gamma, beta = torch.randn(C, device=dev), torch.randn(C, device=dev)
            y = torch.empty(M, C, dtype=bf, device=dev)
            mean, rstd = torch.zeros(M, device=dev), torch.zeros(M, device=dev)
            op = ops.layernorm_fwd(xx, C, gamma, beta, 1e-5, M, C, y, C, mean, rstd)
            op.run()
            t = graph_us([op] * 16)
            rows.setdefault(f"layernorm_fwd M={M:5d} C={C}", []).append((t, 4.0 * M * C / t / 1e3))
        torch.cuda.synchronize()
    print("# libs: " + "  |  ".join(os.path.relpath(l, ROOT) for l in libs))
    for k, v in rows.items():
        print(f"{k:44s} " + "  |  ".join(f"{t:6.1f} us {g:6.0f} GB/s ({g / 80:4.1f} % of 8 TB/s)" for t, g in v)
              + (f"   x{v[-1][0] / v[0][0]:.2f}" if len(v) > 1 else ""))
    if len(libs) > 1:
        for pref in ("groupnorm_fwd B= 4", "gn_apply_stats B= 4", "layernorm_fwd"):
            print(f"# sum {pref}: " + "  |  ".join(f"{sum(v[i][0] for k, v in rows.items() if k.startswith(pref)):7.1f} us"
                                                   for i in range(len(libs))))


if __name__ == "__main__":
    main()
