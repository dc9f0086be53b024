"""What one dependent kernel node costs in a replayed hipGraph on this box: a chain of N tiny launches (leco_advance: one
64-thread workgroup; leco_step_begin over 2^20 elements: ~1 k workgroups, 12 MB of traffic) captured once and replayed, timed
with HIP events.  The per-node time of the tiny chain is the launch-to-launch floor every one of the ~9 500 launches of a step
pays; eager launches through ctypes are timed beside it.
    python tools/graph_node_floor.py [N]"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from leco_amd import hip, ops          # noqa: E402
from leco_amd.unet import _graph_api   # noqa: E402


def timed(fn, reps=5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3      # us per call of fn


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    dev = torch.device("cuda:0")
    lib = _graph_api()
    counter = torch.zeros(1, dtype=torch.int32, device=dev)
    x = torch.randn(1 << 20, device=dev)
    x2 = torch.zeros(2 << 20, dtype=torch.bfloat16, device=dev)
    chains = {"advance (1 workgroup)": [ops.advance(counter) for _ in range(n)],
              "step_begin 2^20 elements (1024 workgroups, 8 MB)": [ops.step_begin(x, x2, 1.0, 1 << 20, None) for _ in range(n)]}
    for name, chain in chains.items():
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        hip.check(lib.leco_graph_begin_capture(side.cuda_stream), "begin")
        ops.run_plan(chain, side.cuda_stream)
        g = C.c_void_p()
        hip.check(lib.leco_graph_end_capture(side.cuda_stream, C.byref(g)), "end")
        cur = torch.cuda.current_stream().cuda_stream
        t_graph = timed(lambda: hip.check(lib.leco_graph_launch(g, cur), "launch"))
        t_eager = timed(lambda: ops.run_plan(chain), reps=2)
        print(f"{name}: graph replay {t_graph / n:.2f} us per node, eager {t_eager / n:.2f} us per launch ({n} nodes)")
        lib.leco_graph_destroy(g)


if __name__ == "__main__":
    main()
