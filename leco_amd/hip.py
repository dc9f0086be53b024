"""ctypes binding of ``libleco_hip.so`` (C ABI declared in ``include/leco_hip.h``).

The shared object is built in-tree by ``__graft_entry__.build()`` (hipcc, gfx950).  There is
no CPU fallback: if the library is missing this module raises, and every op below goes
through it.  (``tests/emu`` can point the loader at a host-emulated build of the *same kernel
sources* via ``_use_library`` -- a kernel-debugging harness, never used by the product path.)
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libleco_hip.so")

A_PLAIN, A_CONV3_S1, A_CONV3_S2, A_CONV3_UP2, A_CONV3_TR2 = range(5)
ACT_NONE, ACT_SILU, ACT_GEGLU = 0, 1, 2


class GemmArgs(C.Structure):
    _fields_ = [
        ("a0", C.c_void_p), ("a1", C.c_void_p), ("lda0", C.c_int64), ("lda1", C.c_int64),
        ("k_split", C.c_int32), ("a_mode", C.c_int32),
        ("batch", C.c_int32), ("h_out", C.c_int32), ("w_out", C.c_int32), ("h_in", C.c_int32),
        ("w_in", C.c_int32),
        ("w", C.c_void_p), ("ldw", C.c_int64),
        ("m", C.c_int32), ("n", C.c_int32), ("k", C.c_int32),
        ("a_ext", C.c_void_p), ("ld_aext", C.c_int64), ("w_ext", C.c_void_p), ("ld_wext", C.c_int64),
        ("ext_k", C.c_int32),
        ("bias", C.c_void_p), ("rowbias", C.c_void_p), ("rows_per_group", C.c_int32), ("ld_rowbias", C.c_int64),
        ("residual", C.c_void_p), ("ldr", C.c_int64),
        ("act", C.c_int32),
        ("c", C.c_void_p), ("ldc", C.c_int64), ("c_f32", C.c_void_p), ("ldc32", C.c_int64),
        ("t_w", C.c_void_p), ("ld_tw", C.c_int64), ("t_rows", C.c_int32), ("t_out", C.c_void_p), ("ld_tout", C.c_int64),
        ("col_stats", C.c_void_p), ("stats_rows", C.c_int32), ("stats_atom", C.c_int32),
    ]


class LoraSite(C.Structure):
    _fields_ = [
        ("down", C.c_void_p * 3), ("up", C.c_void_p * 3),
        ("groups", C.c_int32), ("r", C.c_int32), ("k", C.c_int32), ("n", C.c_int32),
        ("scale", C.c_float), ("taps", C.c_int32),
        ("dn_s", C.c_void_p), ("up_p", C.c_void_p), ("up_t", C.c_void_p), ("dn_p", C.c_void_p),
        ("up_pg", C.c_void_p), ("rp", C.c_int32),
    ]


class WgradProblem(C.Structure):     # mirrors `leco_wgrad_problem` in include/leco_hip.h
    _fields_ = [
        ("p", C.c_void_p), ("ldp", C.c_int64), ("q", C.c_void_p), ("ldq", C.c_int64),
        ("g", C.c_void_p), ("g_sj", C.c_int64), ("g_sc", C.c_int64),
        ("m", C.c_int32), ("r", C.c_int32), ("cols", C.c_int32), ("scale", C.c_float),
        ("a_mode", C.c_int32), ("h_out", C.c_int32), ("w_out", C.c_int32), ("h_in", C.c_int32), ("w_in", C.c_int32),
        ("kh", C.c_int32), ("kw", C.c_int32), ("block_start", C.c_int32), ("blocks_x", C.c_int32),
    ]


class XLin(C.Structure):          # mirrors `leco_xlin`
    _fields_ = [("w", C.c_void_p), ("ldw", C.c_int64), ("bias", C.c_void_p), ("dn", C.c_void_p), ("ld_dn", C.c_int64),
                ("up", C.c_void_p), ("ld_up", C.c_int64), ("t_rows", C.c_int32), ("packed", C.c_int32)]


class XGemmArgs(C.Structure):          # mirrors `leco_xgemm_args`
    _fields_ = [("m", C.c_int32), ("n", C.c_int32), ("k", C.c_int32), ("a", C.c_void_p), ("lda", C.c_int64), ("lin", XLin),
                ("residual", C.c_void_p), ("ldr", C.c_int64), ("c", C.c_void_p), ("ldc", C.c_int64)]


class XBlockTailArgs(C.Structure):     # mirrors `leco_xblock_tail_args`
    _fields_ = [
        ("m", C.c_int32), ("c", C.c_int32), ("heads", C.c_int32), ("skv", C.c_int32), ("rows_per_sample", C.c_int32),
        ("src_rows", C.c_int32),
        ("attn", C.c_void_p), ("ld_attn", C.c_int64), ("h_in", C.c_void_p), ("ld_h", C.c_int64),
        ("to_out1", XLin), ("to_q2", XLin), ("to_out2", XLin), ("ff1", XLin), ("ff2", XLin), ("proj_out", XLin),
        ("ln2_g", C.c_void_p), ("ln2_b", C.c_void_p), ("ln3_g", C.c_void_p), ("ln3_b", C.c_void_p), ("ln_eps", C.c_float),
        ("kp", C.c_void_p), ("vt", C.c_void_p), ("attn_scale", C.c_float),
        ("res", C.c_void_p), ("ld_res", C.c_int64), ("out", C.c_void_p), ("ld_out", C.c_int64),
        ("col_stats", C.c_void_p), ("stats_atom", C.c_int32),
    ]


class XBlockHeadArgs(C.Structure):     # mirrors `leco_xblock_head_args`
    _fields_ = [
        ("m", C.c_int32), ("c", C.c_int32), ("rows_per_sample", C.c_int32),
        ("x", C.c_void_p), ("ld_x", C.c_int64),
        ("gn_cstats", C.c_void_p), ("stats_atom", C.c_int32), ("groups", C.c_int32),
        ("gn_g", C.c_void_p), ("gn_b", C.c_void_p), ("gn_eps", C.c_float),
        ("proj_in", XLin), ("qkv", XLin),
        ("ln1_g", C.c_void_p), ("ln1_b", C.c_void_p), ("ln_eps", C.c_float),
        ("h_out", C.c_void_p), ("ld_hout", C.c_int64), ("qkv_out", C.c_void_p), ("ld_qkv", C.c_int64),
    ]


def pack_fragments(w: torch.Tensor) -> torch.Tensor:
    """[N][K] -> the MFMA fragment order of `leco_xlin.packed` (same numel): [N / 16][K / 32][k-group 4][row 16][8]."""
    n, k = w.shape
    return w.reshape(n // 16, 16, k // 32, 4, 8).permute(0, 2, 3, 1, 4).contiguous().reshape(n, k)


def xlin(w, bias=None, dn=None, up=None, t_rows: int = 0, ldw: Optional[int] = None, ld_dn: Optional[int] = None,
         ld_up: int = 32, packed: bool = False) -> XLin:
    """One Linear of a stripe chain (include/leco_hip.h `leco_xlin`): ``w`` [N][K] bf16, ``dn`` / ``up`` = the packed LoRA
    operand images `dn_s` / `up_p` of the site (None: LoRA off)."""
    x = XLin()
    x.w, x.ldw = ptr(w), (w.shape[-1] if ldw is None else ldw)
    x.bias = ptr(bias)
    x.dn = ptr(dn)
    x.ld_dn = 0 if dn is None else (dn.shape[-1] if ld_dn is None else ld_dn)
    x.up, x.ld_up = ptr(up), ld_up
    x.t_rows = t_rows
    x.packed = 1 if packed else 0
    return x


_lib: Optional[C.CDLL] = None
_lib_path: Optional[str] = None


def _declare(lib: C.CDLL) -> None:
    vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float
    sig = {
        "leco_version": ([], C.c_int),
        "leco_last_error": ([], C.c_char_p),
        "leco_gemm": ([C.POINTER(GemmArgs), vp], C.c_int),
        "leco_gemm_tile": ([C.POINTER(GemmArgs), C.c_int, vp], C.c_int),
        "leco_gemm_ex": ([C.POINTER(GemmArgs), C.c_int, C.c_int, vp, i64, vp], C.c_int),
        "leco_gemm_describe": ([C.POINTER(GemmArgs), C.c_int, C.c_int, vp, i64, C.c_char_p, C.c_int32], C.c_int),
    }
    for name, (args, res) in sig.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = res
    # remaining entry points are declared by the modules that use them via declare()


def declare(name: str, argtypes, restype=C.c_int):
    fn = getattr(lib(), name)
    fn.argtypes = argtypes
    fn.restype = restype
    return fn


def lib() -> C.CDLL:
    global _lib, _lib_path
    if _lib is None:
        if os.environ.get("LECO_AUTOBUILD", "1") != "0":
            # incremental (content-hashed) in-tree rebuild so a stale .so can never be loaded
            try:
                from . import build as _build
                _build.build()
            except Exception as e:  # no hipcc on this machine: fall through to the existence check
                if not os.path.exists(LIB_PATH):
                    raise RuntimeError(f"leco_amd: cannot build the HIP extension ({e!r}); there is no CPU fallback")
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"leco_amd: HIP extension {LIB_PATH} is missing. Build it with "
                "`python -c 'import __graft_entry__ as g; g.build()'` (hipcc --offload-arch=gfx950). "
                "There is no CPU fallback.")
        # LECO_HIP_LIB: a side build of the SAME sources with a measurement switch (tools/ablate_gemm.py, A/B runs); the
        # product library is what loads otherwise
        path = os.environ.get("LECO_HIP_LIB") or LIB_PATH
        _lib = C.CDLL(path)
        _lib_path = path
        _declare(_lib)
    return _lib


def _use_library(path: str) -> None:
    """Test hook: bind to another build of the same C ABI (the host emulator)."""
    global _lib, _lib_path
    _lib = C.CDLL(path)
    _lib_path = path
    _declare(_lib)


def is_emulated() -> bool:
    """True when bound to the host emulator build (tests/emu): a library that is neither the product .so nor a side
    build named by LECO_HIP_LIB."""
    lib()
    return _lib_path != LIB_PATH and _lib_path != os.environ.get("LECO_HIP_LIB")


class LecoError(RuntimeError):
    pass


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().leco_last_error()
        raise LecoError(f"{what} failed rc={rc}: {msg.decode() if msg else ''}")


def stream_ptr(t: Optional[torch.Tensor] = None) -> Optional[int]:
    """hipStream_t of torch's current stream (None on the emulator / CPU tensors)."""
    if t is not None and t.device.type != "cuda":
        return None
    if not torch.cuda.is_available():
        return None
    return torch.cuda.current_stream().cuda_stream


def ptr(t) -> Optional[int]:
    """Device address of a tensor (ints / None pass through, so views can be given as raw addresses)."""
    if t is None or isinstance(t, int):
        return t
    return t.data_ptr()


def gemm_args(a: torch.Tensor, w: torch.Tensor, out: Optional[torch.Tensor], *, m: int, n: int, k: int,
              lda: Optional[int] = None, a1: Optional[torch.Tensor] = None, lda1: int = 0, k_split: int = 0,
              a_mode: int = A_PLAIN, conv=None, ldw: Optional[int] = None,
              a_ext: Optional[torch.Tensor] = None, w_ext: Optional[torch.Tensor] = None, ext_k: int = 0,
              ld_aext: int = 0, ld_wext: int = 0,
              bias: Optional[torch.Tensor] = None, rowbias: Optional[torch.Tensor] = None,
              rows_per_group: int = 0, ld_rowbias: int = 0, residual: Optional[torch.Tensor] = None, ldr: int = 0,
              act: int = ACT_NONE, ldc: Optional[int] = None, out_f32: Optional[torch.Tensor] = None,
              ldc32: int = 0, t_w: Optional[torch.Tensor] = None, t_rows: int = 0,
              t_out: Optional[torch.Tensor] = None, ld_tout: int = 0, col_stats=None, stats_rows: int = 0,
              stats_atom: int = 1) -> GemmArgs:
    """Build the argument block for leco_gemm.  ``conv`` = (batch, h_out, w_out, h_in, w_in).
    ``t_w`` ([32][k] stacked lora_down rows) selects the fused down-projection (see include/leco_hip.h)."""
    g = GemmArgs()
    g.a0 = ptr(a)
    g.a1 = ptr(a1)
    g.lda0 = k if lda is None else lda
    g.lda1 = lda1
    g.k_split = k_split
    g.a_mode = a_mode
    if conv is not None:
        g.batch, g.h_out, g.w_out, g.h_in, g.w_in = conv
    g.w = ptr(w)
    g.ldw = k if ldw is None else ldw
    g.m, g.n, g.k = m, n, k
    g.a_ext, g.w_ext, g.ext_k = ptr(a_ext), ptr(w_ext), ext_k
    g.ld_aext, g.ld_wext = ld_aext or ext_k, ld_wext or ext_k
    g.bias, g.rowbias, g.rows_per_group = ptr(bias), ptr(rowbias), rows_per_group
    g.ld_rowbias = ld_rowbias or n
    g.residual, g.ldr = ptr(residual), ldr or n
    g.act = act
    g.c, g.ldc = ptr(out), (n if ldc is None else ldc)
    g.c_f32, g.ldc32 = ptr(out_f32), ldc32 or n
    g.t_w, g.ld_tw, g.t_rows = ptr(t_w), k, t_rows
    g.t_out, g.ld_tout = ptr(t_out), ld_tout or 32
    g.col_stats, g.stats_rows, g.stats_atom = ptr(col_stats), stats_rows, stats_atom
    return g


def gemm_describe(args: GemmArgs, tile: int = 0, split_k: int = 1, ws_ptr=None, ws_bytes: int = 0) -> str:
    """The kernel instantiation(s) `leco_gemm_ex` would launch for these arguments, named as rocprofv3 names them."""
    buf = C.create_string_buffer(256)
    check(lib().leco_gemm_describe(C.byref(args), tile, split_k, ws_ptr, ws_bytes, buf, 256), "leco_gemm_describe")
    return buf.value.decode()


def gemm(args: GemmArgs, stream=None, tile: int = 0, split_k: int = 1, ws: Optional[torch.Tensor] = None) -> None:
    check(lib().leco_gemm_ex(C.byref(args), tile, split_k, ptr(ws), 0 if ws is None else ws.numel() * ws.element_size(),
                             stream), "leco_gemm")
