"""Typed Python front for every entry point of ``libleco_hip.so``.

Each function returns an :class:`Op` -- the C function plus its marshalled argument tuple --
so that callers can either run it immediately (``op.run()``) or append it to a static launch
plan that is later replayed eagerly or captured into a hipGraph (``leco_amd.unet``).  Tensors
are only used for their device pointers; nothing here computes in PyTorch.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence

import torch

from . import hip
from .hip import ACT_NONE, ACT_SILU, A_PLAIN, GemmArgs, LoraSite, ptr  # noqa: F401

_vp, _i32, _i64, _f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float

_SIGS = {
    "leco_groupnorm_fwd": [_vp, _i64, _vp, _i64, _i32, _vp, _vp, _i32, _i32, _i32, _i32, _f32, _i32, _vp, _vp, _i64, _vp],
    "leco_groupnorm_bwd": [_vp, _i64, _vp, _i64, _i32, _vp, _i64, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _f32, _i32, _vp, _vp, _i64, _vp],
    "leco_groupnorm_apply_stats": [_vp, _i64, _vp, _i64, _i32, _vp, _vp, _i32, _vp, _vp, _i32, _i32, _i32, _i32, _f32, _i32, _vp, _vp, _i64, _vp],
    "leco_colstats": [_vp, _i64, _vp, _i32, _i32, _i32, _i32, _vp],
    "leco_layernorm_fwd": [_vp, _i64, _vp, _vp, _f32, _i32, _i32, _vp, _i64, _vp, _vp, _vp],
    "leco_layernorm_bwd": [_vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp, _i64, _vp],
    "leco_attention_fwd": [_vp, _i64, _i64, _vp, _i64, _i64, _vp, _i64, _i64, _vp, _i64, _i64, _vp, _i32, _i32, _i32, _i32, _i32, _f32, _vp],
    "leco_attention_bwd": [_vp, _i64, _i64, _vp, _i64, _i64, _vp, _i64, _i64, _vp, _i64, _i64, _vp, _i64, _i64,
                           _vp, _vp, _vp, _i64, _i64, _vp, _i64, _i64, _vp, _i64, _i64,
                           _i32, _i32, _i32, _i32, _i32, _f32, _vp],
    "leco_gemm_ex": [C.POINTER(GemmArgs), _i32, _i32, _vp, _i64, _vp],
    "leco_geglu_fwd": [_vp, _i64, _vp, _i64, _i32, _i32, _vp],
    "leco_geglu_bwd": [_vp, _i64, _vp, _i64, _vp, _i64, _i32, _i32, _vp],
    "leco_add": [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _i32, _i32, _vp],
    "leco_upsample2x_bwd": [_vp, _vp, _i32, _i32, _i32, _i32, _vp],
    "leco_conv_in": [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp],
    "leco_conv_out": [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp],
    "leco_conv_out_bwd": [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp],
    "leco_timestep_embedding": [_vp, _vp, _i32, _i32, _i32, _vp, _vp],
    "leco_advance": [_vp, _vp],
    "leco_cfg_ddim_step": [_vp, _vp, _vp, _vp, _vp, _f32, _i64, _vp],
    "leco_cfg_sched_step": [_vp, _vp, _vp, _vp, _vp, _f32, _i64, _vp, _vp, _i32, _vp],
    "leco_esd_loss": [_vp, _vp, _vp, _vp, _f32, _f32, _f32, _i64, _vp, _vp, _vp],
    "leco_esd_loss_cond": [_vp, _vp, _vp, _vp, _f32, _f32, _i64, _vp, _vp, _vp],
    "leco_adamw": [_vp, _vp, _vp, _vp, _vp, _vp, _f32, _f32, _f32, _f32, _i64, _vp],
    "leco_lion": [_vp, _vp, _vp, _vp, _vp, _f32, _f32, _f32, _i64, _vp],
    "leco_cast_f32_bf16": [_vp, _vp, _i64, _vp],
    "leco_memset": [_vp, _i32, _i64, _vp],
    "leco_repeat": [_vp, _vp, _i64, _i32, _vp],
    "leco_step_begin": [_vp, _vp, _i32, _f32, _i64, _vp, _vp],
    "leco_step_mid": [_vp, _vp, _vp, _i64, _i32, _f32, _vp, _vp, _vp, _vp, _i32, _vp],
    "leco_lora_pack": [_vp, _i32, _vp],
    "leco_lora_wgrad_conv": [_vp, _i64, _vp, _i64, _vp, _i64, _i64, _i32, _i32, _i32, _f32, _i32, _i32, _i32, _i32, _i32,
                             _i32, _i32, _vp, _i64, _vp],
    "leco_rowgroup_sum": [_vp, _i64, _vp, _i64, _i32, _i32, _i32, _vp],
    "leco_lora_wgrad_grouped": [_vp, _i32, _i32, _i32, _vp],
    "leco_lora_wgrad": [_vp, _i64, _vp, _i64, _vp, _i64, _i64, _i32, _i32, _i32, _f32, _vp, _i64, _vp],
    "leco_xattn_prep": [_vp, _i64, _vp, _vp, _i32, _i32, _i32, _i32, _vp],
    "leco_xblock_tail": [_vp, _vp],
    "leco_xgemm": [_vp, _vp],
    "leco_xblock_head": [_vp, _vp],
}
# fp32 compute mode (csrc/f32.hip): the same argument lists behind `leco_f32_` entry points; activations / weights /
# LoRA operand images are float.  While `f32_mode(True)` is active (the plan builder of an fp32 engine), every Op that
# has an fp32 twin is created on it.
_F32_TWINS = ("leco_groupnorm_fwd", "leco_groupnorm_bwd", "leco_layernorm_fwd", "leco_layernorm_bwd", "leco_attention_fwd",
              "leco_attention_bwd", "leco_geglu_fwd", "leco_geglu_bwd", "leco_add", "leco_upsample2x_bwd", "leco_conv_in",
              "leco_conv_out", "leco_conv_out_bwd", "leco_timestep_embedding", "leco_cfg_ddim_step", "leco_cfg_sched_step",
              "leco_cast_f32_bf16", "leco_lora_pack", "leco_lora_wgrad_conv", "leco_rowgroup_sum", "leco_lora_wgrad")
for _n in _F32_TWINS:
    _SIGS["leco_f32_" + _n[len("leco_"):]] = _SIGS[_n]
_SIGS["leco_f32_gemm"] = [C.POINTER(GemmArgs), _vp]
_f32_active = False


class f32_mode:
    """Context manager: Ops created inside are bound to the fp32 kernels."""

    def __init__(self, on: bool = True):
        self.on = bool(on)

    def __enter__(self):
        global _f32_active
        self.prev, _f32_active = _f32_active, self.on
        return self

    def __exit__(self, *exc):
        global _f32_active
        _f32_active = self.prev


_fn_cache = {}
_fn_lib = None


def _fn(name: str):
    global _fn_lib
    lib = hip.lib()
    if _fn_lib is not lib:
        _fn_cache.clear()
        _fn_lib = lib
    f = _fn_cache.get(name)
    if f is None:
        f = getattr(lib, name)
        if name in _SIGS:
            f.argtypes = _SIGS[name]
        f.restype = C.c_int
        _fn_cache[name] = f
    return f


class Op:
    """One enqueue-only C-ABI call: ``fn(*args, stream)``."""
    __slots__ = ("name", "fn", "args", "keep", "tag")

    def __init__(self, name: str, args: tuple, keep=None):
        self.tag = None   # plan builders label ops (e.g. "ctx": depends only on the prompt embeddings)
        if _f32_active:
            if name in _F32_TWINS:
                name = "leco_f32_" + name[len("leco_"):]
            elif name == "leco_lora_wgrad_grouped":
                raise RuntimeError("fp32 mode accumulates the LoRA weight gradients per problem (leco_f32_lora_wgrad)")
        self.name = name
        self.fn = _fn(name)
        self.args = args
        self.keep = keep  # python objects whose memory the args point to

    def run(self, stream=None) -> None:
        if stream is None:
            stream = default_stream()
        rc = self.fn(*self.args, stream)
        if rc != 0:
            hip.check(rc, self.name)


def default_stream():
    """torch's current HIP stream handle (None when bound to the host emulator)."""
    if hip.is_emulated() or not torch.cuda.is_available():
        return None
    return torch.cuda.current_stream().cuda_stream


_TRACE_OPS = os.environ.get("LECO_TRACE_OPS", "0") not in ("", "0")


def _describe_op(op) -> str:
    try:
        if op.name.endswith("gemm_ex"):
            g = op.keep[0]
            return f"m={g.m} n={g.n} k={g.k} a_mode={g.a_mode} ext={g.ext_k} tile={op.args[1]} split={op.args[2]} stats={bool(g.col_stats)}"
        return " ".join(str(a) for a in op.args if isinstance(a, int) and 0 <= a < (1 << 24))
    except Exception:
        return ""


def run_plan(plan: Sequence[Op], stream=None) -> None:
    if stream is None:
        stream = default_stream()
    if _TRACE_OPS:       # LECO_TRACE_OPS=1: name every launch on stderr before it goes out and wait for it (a GPU memory
        import sys        # fault kills the process without a Python error: the last name printed is the faulting launch)
        for op in plan:
            print(f"[leco op] {op.name} {_describe_op(op)}", file=sys.stderr, flush=True)
            rc = op.fn(*op.args, stream)
            if rc != 0:
                hip.check(rc, op.name)
            if torch.cuda.is_available() and not hip.is_emulated():
                torch.cuda.synchronize()
        return
    for op in plan:
        rc = op.fn(*op.args, stream)
        if rc != 0:
            hip.check(rc, op.name)


# ---------------------------------------------------------------------------------------------
def gemm(args: GemmArgs, keep=None, ws: Optional[torch.Tensor] = None, tile: int = 0, split_k: int = 0) -> Op:
    """``ws``: fp32 scratch for split-K partial slabs (without it the GEMM never splits).  ``tile == 0 and
    split_k == 0``: the launch shape comes from the tuner (leco_amd/tune.py: table / measurement), else the C heuristic."""
    if _f32_active:      # one exact-fp32 MFMA kernel, no launch shapes to choose
        return Op("leco_f32_gemm", (C.byref(args),), keep=(args, keep))
    if tile == 0 and split_k == 0:
        from . import tune
        tile, split_k = tune.choose(args, ws)
        if ws is None and split_k == 0:
            split_k = 1
    if ws is None:
        return Op("leco_gemm_ex", (C.byref(args), tile, max(1, split_k), None, 0), keep=(args, keep))
    return Op("leco_gemm_ex", (C.byref(args), tile, split_k, ws.data_ptr(), ws.numel() * ws.element_size()),
              keep=(args, keep, ws))


def groupnorm_fwd(x0, ld0, x1, ld1, c0, gamma, beta, batch, hw, c, groups, eps, act, stats, y, ldy) -> Op:
    return Op("leco_groupnorm_fwd", (ptr(x0), ld0, ptr(x1), ld1, c0, ptr(gamma), ptr(beta), batch, hw, c,
                                     groups, eps, act, ptr(stats), ptr(y), ldy))


def groupnorm_bwd(x0, ld0, x1, ld1, c0, dy, lddy, gamma, beta, stats, batch, hw, c, groups, eps, act,
                  bstats, dx, lddx) -> Op:
    return Op("leco_groupnorm_bwd", (ptr(x0), ld0, ptr(x1), ld1, c0, ptr(dy), lddy, ptr(gamma), ptr(beta),
                                     ptr(stats), batch, hw, c, groups, eps, act, ptr(bstats), ptr(dx), lddx))


def layernorm_fwd(x, ldx, gamma, beta, eps, m, c, y, ldy, mean, rstd) -> Op:
    return Op("leco_layernorm_fwd", (ptr(x), ldx, ptr(gamma), ptr(beta), eps, m, c, ptr(y), ldy, ptr(mean),
                                     ptr(rstd)))


def layernorm_bwd(x, ldx, dy, lddy, gamma, mean, rstd, dres, ldres, m, c, dx, lddx) -> Op:
    return Op("leco_layernorm_bwd", (ptr(x), ldx, ptr(dy), lddy, ptr(gamma), ptr(mean), ptr(rstd), ptr(dres),
                                     ldres, m, c, ptr(dx), lddx))


def attention_fwd(q, ldq, bsq, k, ldk, bsk, v, ldv, bsv, o, ldo, bso, lse, batch, heads, sq, skv, d, scale) -> Op:
    return Op("leco_attention_fwd", (q, ldq, bsq, k, ldk, bsk, v, ldv, bsv, o, ldo, bso, ptr(lse), batch, heads,
                                     sq, skv, d, scale))


def attention_bwd(q, ldq, bsq, k, ldk, bsk, v, ldv, bsv, o, ldo, bso, do, lddo, bsdo, lse, delta,
                  dq, lddq, bsdq, dk, lddk, bsdk, dv, lddv, bsdv, batch, heads, sq, skv, d, scale) -> Op:
    return Op("leco_attention_bwd", (q, ldq, bsq, k, ldk, bsk, v, ldv, bsv, o, ldo, bso, do, lddo, bsdo,
                                     ptr(lse), ptr(delta), dq, lddq, bsdq, dk, lddk, bsdk, dv, lddv,
                                     bsdv, batch, heads, sq, skv, d, scale))


def geglu_fwd(u, ldu, y, ldy, m, f) -> Op:
    return Op("leco_geglu_fwd", (ptr(u), ldu, ptr(y), ldy, m, f))


def geglu_bwd(u, ldu, dy, lddy, du, lddu, m, f) -> Op:
    return Op("leco_geglu_bwd", (ptr(u), ldu, ptr(dy), lddy, ptr(du), lddu, m, f))


def add(a, lda, b, ldb, c, ldc, out, ldo, m, cols) -> Op:
    """a, b, c, out are raw device addresses (ints) so that column-offset views can be passed."""
    return Op("leco_add", (a, lda, b, ldb, c, ldc, out, ldo, m, cols))


def upsample2x_bwd(dy, dx, batch, h, w, c) -> Op:
    return Op("leco_upsample2x_bwd", (ptr(dy), ptr(dx), batch, h, w, c))


def conv_in(x, w, bias, y, batch, h, wd, cin, cout) -> Op:
    return Op("leco_conv_in", (ptr(x), ptr(w), ptr(bias), ptr(y), batch, h, wd, cin, cout))


def conv_out(x, w, bias, y, batch, h, wd, c, cout) -> Op:
    return Op("leco_conv_out", (ptr(x), ptr(w), ptr(bias), ptr(y), batch, h, wd, c, cout))


def conv_out_bwd(dy, w, dx, batch, h, wd, c, cout) -> Op:
    return Op("leco_conv_out_bwd", (ptr(dy), ptr(w), ptr(dx), batch, h, wd, c, cout))


def timestep_embedding(t_table, idx, t_stride, n, dim, out) -> Op:
    return Op("leco_timestep_embedding", (ptr(t_table), ptr(idx), t_stride, n, dim, ptr(out)))


def advance(counter) -> Op:
    return Op("leco_advance", (ptr(counter),))


def cfg_sched_step(pred, x, x2, coef, step, guidance, half_n, noise=None, hist=None, n_hist=0) -> Op:
    return Op("leco_cfg_sched_step", (ptr(pred), ptr(x), ptr(x2), ptr(coef), ptr(step), guidance, half_n, ptr(noise),
                                      ptr(hist), n_hist))


def cfg_ddim_step(pred, x, x2, coef, step, guidance, half_n) -> Op:
    return Op("leco_cfg_ddim_step", (ptr(pred), ptr(x), ptr(x2), ptr(coef), ptr(step), guidance, half_n))


def esd_loss(tgt, pos, neu, unc, g_pred, g_loss, sign, half_n, loss, dpred) -> Op:
    return Op("leco_esd_loss", (ptr(tgt), ptr(pos), ptr(neu), ptr(unc), g_pred, g_loss, sign, half_n, ptr(loss),
                                ptr(dpred)))


def esd_loss_cond(tgt_c, pos_c, neu_c, unc_c, g_loss, sign, half_n, loss, dpred_c) -> Op:
    """The ESD objective on conditional-only predictions (the de-duplicated step: guidance_scale = 1 makes u + 1 (c - u) = c)."""
    return Op("leco_esd_loss_cond", (ptr(tgt_c), ptr(pos_c), ptr(neu_c), ptr(unc_c), g_loss, sign, half_n, ptr(loss), ptr(dpred_c)),
              keep=(tgt_c, pos_c, neu_c, unc_c, loss, dpred_c))


def adamw(p, g, m, v, shadow, hyper, beta1, beta2, eps, wd, n) -> Op:
    return Op("leco_adamw", (ptr(p), ptr(g), ptr(m), ptr(v), ptr(shadow), ptr(hyper), beta1, beta2, eps, wd, n))


def lion(p, g, m, shadow, hyper, beta1, beta2, wd, n) -> Op:
    return Op("leco_lion", (ptr(p), ptr(g), ptr(m), ptr(shadow), ptr(hyper), beta1, beta2, wd, n))


def cast_f32_bf16(x, y, n) -> Op:
    return Op("leco_cast_f32_bf16", (ptr(x), ptr(y), n))


def memset(t: torch.Tensor, value: int = 0) -> Op:
    return Op("leco_memset", (ptr(t), value, t.numel() * t.element_size()))


def repeat(src: int, dst: int, nbytes: int, reps: int, keep=None) -> Op:
    """dst = `reps` back-to-back copies of the `nbytes` at src (raw device addresses)."""
    return Op("leco_repeat", (src, dst, nbytes, reps), keep=keep)


def step_begin(x: torch.Tensor, x2: torch.Tensor, scale: float, half_n: int, t_idx: Optional[torch.Tensor]) -> Op:
    """x2 = cat([scale x] * 2) in x2's dtype (bf16 / fp32), *t_idx = 0: the first UNet input of the denoising passes."""
    return Op("leco_step_begin", (ptr(x), ptr(x2), 1 if x2.dtype == torch.float32 else 0, float(scale), half_n, ptr(t_idx)),
              keep=(x, x2, t_idx))


def step_mid(src: torch.Tensor, dst_a: Optional[torch.Tensor], dst_b: Optional[torch.Tensor], reps_b: int, t_cur: float,
             plan_a, plan_b, slot: int) -> Op:
    """dst_a = src, dst_b = `reps_b` copies of src; both plans' timestep slot `slot` = t_cur and their step index -> slot."""
    nbytes = src.numel() * src.element_size()
    ta = None if plan_a is None else plan_a.t_table.data_ptr() + 4 * slot
    tb = None if plan_b is None else plan_b.t_table.data_ptr() + 4 * slot
    return Op("leco_step_mid", (ptr(src), ptr(dst_a), ptr(dst_b), nbytes, reps_b, float(t_cur), ta, tb,
                                None if plan_a is None else ptr(plan_a.t_idx), None if plan_b is None else ptr(plan_b.t_idx), slot),
              keep=(src, dst_a, dst_b))


def lora_pack(sites_dev: torch.Tensor, nsites: int) -> Op:
    return Op("leco_lora_pack", (ptr(sites_dev), nsites))


def lora_wgrad(p, ldp, q, ldq, g, g_sj, g_sc, m, r, cols, scale, part: Optional[torch.Tensor] = None) -> Op:
    """p, q, g are raw device addresses (ints).  ``part``: fp32 scratch -> deterministic (atomic-free) accumulation."""
    return Op("leco_lora_wgrad", (p, ldp, q, ldq, g, g_sj, g_sc, m, r, cols, scale, ptr(part),
                                  0 if part is None else part.numel() * part.element_size()), keep=(part,))


def lora_wgrad_grouped(problems: list, device) -> Optional[Op]:
    """ONE launch for a list of weight-gradient problems.  Each entry: dict(p, ldp, q, ldq, g, g_sj, g_sc, m, r, cols,
    scale[, a_mode, h_out, w_out, h_in, w_in, kh, kw]) with raw device addresses, as for `lora_wgrad` /
    leco_lora_wgrad_conv.  The problem table is built here and kept on the device."""
    if not problems:
        return None
    tab = (hip.WgradProblem * len(problems))()
    start = 0
    for d, pr in zip(tab, problems):
        for k, v in pr.items():
            setattr(d, k, v)
        d.blocks_x = -(-pr["cols"] // 256)
        d.block_start = start
        start += d.blocks_x * -(-pr["m"] // 128)
    raw = torch.frombuffer(bytearray(bytes(tab)), dtype=torch.uint8).to(device)
    return Op("leco_lora_wgrad_grouped", (raw.data_ptr(), len(problems), start, max(pr["r"] for pr in problems)), keep=(raw,))


# ---- row-stripe fused transformer-block kernels (csrc/stripe.hip; bf16, forward-only plans) ---------------------------------
def xblock_supported(c: int, heads: int, skv: int, rows_per_sample: int) -> bool:
    f = hip.declare("leco_xblock_supported", [_i32, _i32, _i32, _i32])
    return bool(f(c, heads, skv, rows_per_sample))


def xattn_prep(kv_ptr: int, ld_kv: int, kp: torch.Tensor, vt: torch.Tensor, batch: int, heads: int, skv: int, head_dim: int) -> Op:
    """Cross-attention K / V^T operand images for `xblock_tail` (once per step: they depend only on the prompt)."""
    return Op("leco_xattn_prep", (kv_ptr, ld_kv, ptr(kp), ptr(vt), batch, heads, skv, head_dim), keep=(kp, vt))


def xattn_buffers(batch: int, heads: int, head_dim: int, device) -> tuple:
    dv = (head_dim + 15) // 16 * 16
    return (torch.zeros(batch * heads * 80 * 64, dtype=torch.bfloat16, device=device),
            torch.zeros(batch * heads * dv * 96, dtype=torch.bfloat16, device=device))


def xblock_tail(args: "hip.XBlockTailArgs", keep=None) -> Op:
    """Tail of a BasicTransformerBlock (+ proj_out) as ONE launch (include/leco_hip.h `leco_xblock_tail_args`)."""
    return Op("leco_xblock_tail", (C.cast(C.pointer(args), _vp),), keep=(args, keep))


def xblock_head(args: "hip.XBlockHeadArgs", keep=None) -> Op:
    """GroupNorm apply + proj_in + LayerNorm + q|k|v of a Transformer2DModel's first block as ONE launch (`leco_xblock_head_args`)."""
    return Op("leco_xblock_head", (C.cast(C.pointer(args), _vp),), keep=(args, keep))


def xgemm_supported(m: int, n: int, k: int) -> bool:
    f = hip.lib().leco_xgemm_supported
    f.argtypes, f.restype = [_i32, _i32, _i32], C.c_int
    return bool(f(m, n, k))


def xgemm(a_ptr: int, lda: int, lin: "hip.XLin", c_ptr: int, ldc: int, m: int, n: int, k: int, residual: Optional[int] = None,
          ldr: int = 0, keep=None) -> Op:
    """c = a lin.w^T (+ LoRA) + bias (+ residual) on the A-stationary kernel (include/leco_hip.h `leco_xgemm_args`)."""
    A = hip.XGemmArgs()
    A.m, A.n, A.k = m, n, k
    A.a, A.lda = a_ptr, lda
    A.lin = lin
    A.residual, A.ldr = residual, ldr
    A.c, A.ldc = c_ptr, ldc
    return Op("leco_xgemm", (C.cast(C.pointer(A), _vp),), keep=(A, lin, keep))


def deterministic_default() -> bool:
    """LECO_DETERMINISTIC=1: bitwise reproducible steps (LoRA wgrads without atomics; ~1-2 % slower)."""
    import os
    return os.environ.get("LECO_DETERMINISTIC", "0") not in ("", "0")

