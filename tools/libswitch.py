"""A/B helpers for the micro-benchmarks: bind another build of libleco_hip (same C ABI) and time launch chains from ONE
hipGraph.  The graph entry points are declared on whichever library object is bound (leco_amd.unet._graph_api declares them
once, on the first)."""
import ctypes as C
import os

import torch

from leco_amd import hip, ops


def use_lib(path: str) -> None:
    os.environ["LECO_HIP_LIB"] = path          # (hip.is_emulated(): a side build named here is a GPU library)
    hip._use_library(path)


def _graph_lib():
    lib = hip.lib()
    for nm, at in [("leco_graph_begin_capture", [C.c_void_p]), ("leco_graph_end_capture", [C.c_void_p, C.POINTER(C.c_void_p)]),
                   ("leco_graph_launch", [C.c_void_p, C.c_void_p]), ("leco_graph_destroy", [C.c_void_p])]:
        getattr(lib, nm).argtypes = at
        getattr(lib, nm).restype = C.c_int
    return lib


def graph_us(chain, reps=20):
    """us per op of `chain` (a list of ops.Op) replayed from one captured graph"""
    lib = _graph_lib()
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
