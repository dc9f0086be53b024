"""Launch-shape tuner for ``leco_gemm_ex`` ("measure, don't guess").

The GEMM kernel family has four tile shapes, two wave layouts for the 128x128 tile and split-K; which one is fastest
for a given contraction depends on how the grid fills 256 CUs, on K depth and on what rides along (LoRA tile, GEGLU
epilogue).  The C side keeps a heuristic (tile = 0, split_k = 0); this module replaces it per SHAPE by a measurement:
every candidate (tile, split_k) is timed with HIP events on the caller's real operands when a plan is built, the winner
is cached under a shape key, and the cache is a small JSON table (``gemm_tune_gfx950.json``, committed for the BASELINE
shapes so that plans are reproducible; missing shapes are tuned on first use on a GPU and added in memory).

    LECO_GEMM_TUNE=0      never tune, never use the table (C heuristic only)
    LECO_GEMM_TUNE=table  (default) use the table, heuristic for shapes it lacks
    LECO_GEMM_TUNE=1      use the table, measure shapes it lacks
    LECO_GEMM_TUNE=force  re-measure everything (tools/tune_report.py writes the table)
"""
from __future__ import annotations

import ctypes as C
import json
import os
from typing import Dict, Optional, Tuple

import torch

from . import hip
from .hip import GemmArgs

TABLE_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gemm_tune_gfx950.json")
_table: Optional[Dict[str, list]] = None
_measured: Dict[str, dict] = {}      # key -> {"best": (tile, split), "times": {cand: us}} for shapes tuned in this process


def mode() -> str:
    return os.environ.get("LECO_GEMM_TUNE", "table")


def table() -> Dict[str, list]:
    global _table
    if _table is None:
        try:
            with open(TABLE_PATH) as fh:
                _table = json.load(fh)
        except (OSError, ValueError):
            _table = {}
    return _table


def shape_key(g: GemmArgs, has_ws: bool = True) -> str:
    """Everything the best launch shape depends on, including whether the call site supplies a split-K workspace (the
    same contraction appears with and without one; a table entry with split > 1 must not be applied to the latter)."""
    conv = f"c{g.batch}x{g.h_out}x{g.w_out}<{g.h_in}x{g.w_in}" if g.a_mode else ""
    ext = g.ext_k if (g.a_ext or g.t_w) else 0
    return (f"m{g.m}n{g.n}k{g.k}a{g.a_mode}{conv}"
            f"{'s' if g.a1 else ''}e{ext}{'T%d' % g.t_rows if g.t_w else ''}{'o' if g.t_out else ''}"
            f"{'r' if g.residual else ''}{'b' if g.bias else ''}{'B' if g.rowbias else ''}A{g.act}{'f' if g.c_f32 else ''}"
            f"{'S' if g.col_stats else ''}"      # producer-side GroupNorm statistics: extra epilogue work, tiled split-K finish
            f"{'' if has_ws else 'W0'}")


def candidates(g: GemmArgs, has_ws: bool):
    nk = g.k // 64
    plain = g.a_mode == 0
    geglu = g.act == hip.ACT_GEGLU
    tiles = [1, 4, 5, 6] if geglu else [1, 2, 3, 4] + ([5, 6, 11] if plain else [])
    out = [(0, 0)]                                   # the C heuristic itself
    if g.a_mode in (hip.A_CONV3_S1, hip.A_CONV3_UP2) and not g.t_w:      # (K-extension launches included: conv_patch.hip carries a_ext)
        # patch-staged kernel (conv_patch.hip): tile ids 7..10; K splits are whole 64-channel chunks
        chunks = g.k // 9 // 64
        for t, (bm, bn) in {7: (256, 128), 8: (128, 160), 9: (128, 128), 10: (256, 160)}.items():
            if bn == 160 and g.n % 160:
                continue
            blocks = -(-g.m // bm) * -(-g.n // bn)
            out.append((t, 1))
            if has_ws and blocks < 256:
                out += [(t, sp) for sp in (2, 3, 4, 5, 6, 8, 10, 12, 16) if blocks * sp <= 768 and sp <= chunks]
    for t in tiles:
        bm, bn = {1: (128, 128), 2: (128, 160), 3: (64, 64), 4: (256, 128), 5: (128, 128), 6: (128, 128), 11: (128, 64)}[t]
        blocks = -(-g.m // bm) * -(-g.n // bn)
        if t == 2 and g.n % 160 and g.n > 160:
            continue
        out.append((t, 1))
        if has_ws and not geglu and nk >= 16 and blocks <= 192:
            for sp in (2, 3, 4, 6, 8, 12, 16):
                if blocks * sp <= 640 and nk // sp >= 4:
                    out.append((t, sp))
    return out


def _time(fn, args, tile, split, ws_ptr, ws_bytes, stream, iters):
    for _ in range(2):
        rc = fn(args, tile, split, ws_ptr, ws_bytes, stream)
        if rc != 0:
            return None
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e30
    for _ in range(3):
        e0.record()
        for _ in range(iters):
            fn(args, tile, split, ws_ptr, ws_bytes, stream)
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters * 1e3)
    return best


def choose(g: GemmArgs, ws: Optional[torch.Tensor]) -> Tuple[int, int]:
    """(tile, split_k) for this contraction: table / measurement / (0, 0) = C heuristic."""
    m = mode()
    if m == "0" or hip.is_emulated() or not torch.cuda.is_available():
        return 0, 0
    key = shape_key(g, ws is not None)
    if m != "force":
        hit = table().get(key) or (_measured.get(key, {}).get("best"))
        if hit:
            return int(hit[0]), int(hit[1])
        if m != "1":
            return 0, 0
    elif key in _measured:
        return _measured[key]["best"]
    if g.residual and g.residual == g.c:             # in-place accumulation (chained LoRA slices): repeated timing launches
        return 0, 0                                  # would compound garbage on a live plan buffer -- not worth tuning
    fn = hip.lib().leco_gemm_ex
    stream = torch.cuda.current_stream().cuda_stream
    ws_ptr = ws.data_ptr() if ws is not None else None
    ws_bytes = ws.numel() * ws.element_size() if ws is not None else 0
    times = {}
    flops = 2.0 * g.m * g.n * g.k
    iters = 8 if flops > 2e10 else 20
    for tile, split in candidates(g, ws is not None):
        t = _time(fn, C.byref(g), tile, split, ws_ptr, ws_bytes, stream, iters)
        if t is not None:
            times[(tile, split)] = t
    if not times:                                    # every candidate launch failed: leave it to the C side to report
        _measured[key] = {"best": (0, 0), "times": {}}
        return 0, 0
    base = times.get((0, 0))
    best = min(times, key=times.get)
    if base is not None and times[best] > 0.97 * base:      # within noise of the heuristic: keep the heuristic
        best = (0, 0)
    _measured[key] = {"best": best, "times": times}
    return best


def report():
    """[(key, heuristic us, best (tile, split), best us)] for every shape measured in this process."""
    rows = []
    for key, rec in _measured.items():
        t = rec["times"]
        rows.append((key, t.get((0, 0)), rec["best"], t[rec["best"]] if rec["best"] in t else None))
    return rows


def save_table(path: str = TABLE_PATH, merge: bool = True) -> int:
    tab = dict(table()) if merge else {}
    for key, rec in _measured.items():
        if rec["best"] != (0, 0):
            tab[key] = [int(rec["best"][0]), int(rec["best"][1])]
        else:
            tab.pop(key, None)
    with open(path, "w") as fh:
        json.dump(dict(sorted(tab.items())), fh, indent=0)
    return len(tab)
