"""Every launch shape the UNet plans run at full size, one by one, against fp32 torch.

The whole-step parity tests (`test_fullsize.py`) compare error NORMS of a step; a wrong tile in one of ~1700 launches can
hide under them.  The committed launch-shape table (`leco_amd/gemm_tune_gfx950.json`) lists every distinct GEMM /
convolution problem of the four BASELINE configurations -- sizes, gather mode, epilogue, LoRA form -- together with the
(tile, split-K) the plans launch it with.  This test rebuilds each problem from its key with random operands, launches it
through the C ABI exactly as the table says and compares the result with the same contraction in fp32 torch on the same
device, per 128 x 128 output block (so a single bad tile fails, not only a bad norm).

GPU tier: all entries.  CPU tier (emulator): the smallest entry of every gather mode / epilogue family, which is what keeps
the reference construction in this file honest."""
import json
import os
import re

import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err
from leco_amd import hip, ops

bf = torch.bfloat16
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TABLE = json.load(open(os.path.join(ROOT, "leco_amd", "gemm_tune_gfx950.json")))
KEY = re.compile(r"m(\d+)n(\d+)k(\d+)a(\d)(?:c(\d+)x(\d+)x(\d+)<(\d+)x(\d+))?(s?)e(\d+)(?:T(\d+))?(o?)(r?)(b?)(B?)A(\d)(f?)(S?)(W0)?$")


def parse(key):
    g = KEY.match(key).groups()
    d = dict(m=int(g[0]), n=int(g[1]), k=int(g[2]), a_mode=int(g[3]), two=bool(g[9]), ext=int(g[10]),
             t_rows=int(g[11]) if g[11] else 0, t_out=bool(g[12]), res=bool(g[13]), bias=bool(g[14]), rowbias=bool(g[15]),
             act=int(g[16]), f32=bool(g[17]), stats=bool(g[18]), no_ws=bool(g[19]))
    if d["a_mode"]:
        d["conv"] = tuple(int(x) for x in g[4:9])        # batch, h_out, w_out, h_in, w_in
    return d


def family(d):
    return (d["a_mode"], d["two"], d["ext"] > 0, d["t_rows"], d["res"], d["bias"], d["rowbias"], d["act"], d["f32"], d["stats"])


def run_entry(key, tile, split, dev, seed=0):
    d = parse(key)
    M, N, K = d["m"], d["n"], d["k"]
    gen = torch.Generator().manual_seed(seed)

    def rnd(*shape, scale=1.0):
        return ((torch.rand(*shape, generator=gen) * 2 - 1) * scale).to(dev)
    kw = {}
    if d["a_mode"]:
        B, ho, wo, hi, wi = d["conv"]
        cin = K // 9
        x = rnd(B * hi * wi, cin).to(bf)
        a_src, lda = x, cin
        kw.update(a_mode=d["a_mode"], conv=d["conv"])
    else:
        cin = K
        x = rnd(M, K).to(bf)
        a_src, lda = x, K
    if d["two"]:           # the channel range comes from two tensors (skip concatenations): split at a 64-multiple
        c0 = max(64, (cin // 2) // 64 * 64)
        x0, x1 = x[:, :c0].contiguous(), x[:, c0:].contiguous()
        a_src, lda = x0, c0
        kw.update(a1=x1, lda1=cin - c0, k_split=c0)
    w = rnd(N, K, scale=(3.0 / K) ** 0.5).to(bf)
    # fp32 reference of the gathered contraction
    xf, wf = x.float(), w.float()
    if d["a_mode"] == 0:
        ref = xf @ wf.T
    else:
        # nine shifted matmuls over the NHWC image (no MIOpen: its per-shape kernel search would dominate the GPU tier)
        img = xf.view(B, hi, wi, cin)
        stride = 2 if d["a_mode"] == hip.A_CONV3_S2 else 1
        if d["a_mode"] == hip.A_CONV3_UP2:
            img = img.repeat_interleave(2, 1).repeat_interleave(2, 2)
        elif d["a_mode"] == hip.A_CONV3_TR2:   # dgrad of a stride-2 conv = stride-1 conv over the zero-inserted gradient image
            z = torch.zeros(B, 2 * hi, 2 * wi, cin, device=dev)
            z[:, ::2, ::2] = img
            img = z
        pad = F.pad(img, (0, 0, 1, 1, 1, 1))
        ref = torch.zeros(M, N, device=dev)
        for kh in range(3):
            for kw_ in range(3):
                rows = pad[:, kh:kh + stride * ho:stride, kw_:kw_ + stride * wo:stride, :]
                assert rows.shape[1:3] == (ho, wo), (rows.shape, ho, wo)
                tap = kh * 3 + kw_
                ref += rows.reshape(M, cin) @ wf[:, tap * cin:(tap + 1) * cin].T
    T = None
    if d["ext"]:
        E = d["ext"]
        up = rnd(N, E, scale=0.3).to(bf)
        if d["t_rows"]:        # fused down-projection: T = A t_w^T inside the K sweep, rounded to bf16
            R = 12 if d["t_rows"] == 16 else 24
            tw = torch.zeros(32, K, device=dev)
            tw[:R] = rnd(R, K, scale=(3.0 / K) ** 0.5)
            tw = tw.to(bf)
            up[:, R:] = 0
            T = (xf @ tw.float().T).to(bf)
            kw.update(w_ext=up, ext_k=E, t_w=tw, t_rows=d["t_rows"])
            if d["t_out"]:
                tout = torch.full((M, 32), 7.0, dtype=bf, device=dev)
                kw.update(t_out=tout)
        else:                  # separate low-rank image
            T = rnd(M, E).to(bf)
            kw.update(a_ext=T, w_ext=up, ext_k=E)
        ref = ref + T.float() @ up.float().T
    if d["bias"]:
        bias = rnd(N)
        kw.update(bias=bias)
        ref = ref + bias
    if d["rowbias"]:
        rows = d["conv"][1] * d["conv"][2]
        rb = rnd(M // rows, N)
        kw.update(rowbias=rb, rows_per_group=rows)
        ref = ref + rb.repeat_interleave(rows, 0)
    if d["res"]:               # inside the activation: act(acc + bias + rowbias + residual), include/leco_hip.h:34
        res = rnd(M, N).to(bf)
        kw.update(residual=res)
        ref = ref + res.float()
    if d["act"] == hip.ACT_SILU:
        ref = F.silu(ref)
    n_out = N
    if d["act"] == hip.ACT_GEGLU:      # interleaved value / gate column blocks of 64 -> value * gelu(gate), [M][N / 2]
        u = ref.view(M, N // 128, 2, 64)
        ref = (u[:, :, 0] * F.gelu(u[:, :, 1])).reshape(M, N // 2)
        n_out = N // 2
        kw.update(ldc=n_out)
    cs = None
    if d["stats"]:            # producer-side GroupNorm statistics: {sum, sumsq} per sample and atom of columns (leco_hip.h)
        srows = d["conv"][1] * d["conv"][2] if d["a_mode"] else M
        atom = 10 if N % 10 == 0 else 2
        cs = torch.zeros(M // srows, N // atom, 2, device=dev)
        kw.update(col_stats=cs, stats_rows=srows, stats_atom=atom)
    out = None if d["f32"] else torch.zeros(M, n_out, dtype=bf, device=dev)
    o32 = torch.zeros(M, n_out, device=dev) if d["f32"] else None
    ws = None if d["no_ws"] else torch.empty(max(1, split) * M * N + 1024, device=dev)
    g = hip.gemm_args(a_src, w, out, m=M, n=N, k=K, lda=lda, out_f32=o32, act=d["act"], **kw)
    hip.gemm(g, ops.default_stream(), tile, split, ws)
    if dev.type == "cuda":
        torch.cuda.synchronize()
    got = (o32 if d["f32"] else out).float()
    # per 128 x 128 output block: a single wrong tile must fail
    pm, pn = -(-M // 128) * 128, -(-n_out // 128) * 128
    e2 = F.pad((got - ref) ** 2, (0, pn - n_out, 0, pm - M)).view(pm // 128, 128, pn // 128, 128).sum((1, 3))
    r2 = F.pad(ref ** 2, (0, pn - n_out, 0, pm - M)).view(pm // 128, 128, pn // 128, 128).sum((1, 3))
    worst = float((e2 / r2.clamp_min(1e-20)).sqrt().max())
    tol = 2e-3 if d["f32"] else 1.5e-2
    assert worst < tol, f"{key} tile={tile} split={split}: worst 128x128 block rel {worst:.3g}"
    if d["t_out"]:
        R = 12 if d["t_rows"] == 16 else 24
        assert rel_err(tout[:, :R], T[:, :R]) < 1e-2 and float(tout[:, R:].float().abs().max()) == 0.0, key
    if cs is not None and out is not None:      # the statistics are those of the STORED bf16 values
        y = out.float().view(cs.shape[0], -1, cs.shape[1], N // cs.shape[1])
        want = torch.stack([y.sum((1, 3)), (y * y).sum((1, 3))], dim=-1)
        assert rel_err(cs, want) < 1e-4, f"{key}: column statistics"
    return worst


def test_table_keys_are_what_the_tuner_derives(dev):
    """The parser of this file and `tune.shape_key` agree: a problem rebuilt from a key has that key."""
    from leco_amd import tune
    x = torch.zeros(8, 8, dtype=bf)
    for key in list(TABLE)[::7]:
        d = parse(key)
        kw = dict(a_mode=d["a_mode"], conv=d.get("conv"))
        if d["two"]:
            kw.update(a1=x, lda1=64, k_split=64)
        if d["ext"]:
            kw.update(w_ext=x, ext_k=d["ext"])
            kw.update(dict(t_w=x, t_rows=d["t_rows"]) if d["t_rows"] else dict(a_ext=x))
        if d["t_out"]:
            kw.update(t_out=x)
        g = hip.gemm_args(x, x, None if d["f32"] else x, m=d["m"], n=d["n"], k=d["k"], bias=x if d["bias"] else None,
                          rowbias=x if d["rowbias"] else None, residual=x if d["res"] else None, act=d["act"],
                          out_f32=x if d["f32"] else None, col_stats=x if d["stats"] else None, **kw)
        assert tune.shape_key(g, not d["no_ws"]) == key


def test_every_tuned_launch_shape_matches_fp32_torch(dev):
    entries = sorted(TABLE.items(), key=lambda kv: parse(kv[0])["m"] * parse(kv[0])["n"] * parse(kv[0])["k"])
    if dev.type != "cuda":          # emulator: the smallest problem of each family
        seen, few = set(), []
        for key, ts in entries:
            f = family(parse(key))
            if f not in seen:
                seen.add(f)
                few.append((key, ts))
        entries = [e for e in few if parse(e[0])["m"] * parse(e[0])["n"] * parse(e[0])["k"] <= 4e9]
    worst = {}
    for i, (key, (tile, split)) in enumerate(entries):
        worst[key] = run_entry(key, tile, split, dev, seed=i)
    top = sorted(worst.items(), key=lambda kv: -kv[1])[:3]
    print(f"{len(entries)} launch shapes of the tuned table vs fp32 torch; worst 128x128-block rel errors: "
          + ", ".join(f"{k} {v:.2e}" for k, v in top))


# ---------------------------------------------------------------------------------------------------------------------
# the other kernels at the sizes the BASELINE configurations launch them (GPU only: fp32 torch on the same device)
# ---------------------------------------------------------------------------------------------------------------------
def _gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from conftest import _bind_hip
    _bind_hip()
    return torch.device("cuda:0")


# (B, heads, Sq, Skv, d): SD1.5 512^2 self / cross attention of the three levels + mid block (UNet batch 4; the batched
# frozen pass has 12), SD2.1-768 level 0 (d = 64, S = 9216), SDXL 1024^2 (d = 64, 10 / 20 heads)
FULL_ATTN = [(4, 8, 4096, 4096, 40), (12, 8, 4096, 77, 40), (4, 8, 1024, 1024, 80), (4, 8, 1024, 77, 80), (4, 8, 256, 256, 160),
             (4, 8, 64, 64, 160), (4, 8, 64, 77, 160), (4, 5, 9216, 9216, 64), (2, 10, 4096, 4096, 64), (2, 20, 1024, 77, 64)]


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,Sq,Skv,D", FULL_ATTN)
def test_attention_full_size_fwd_bwd_on_gpu(B, H, Sq, Skv, D):
    dev = _gpu()
    torch.manual_seed(5)
    C = H * D
    q, k, v, do = [torch.randn(B, s, C, device=dev).to(bf) for s in (Sq, Skv, Skv, Sq)]
    o = torch.zeros(B, Sq, C, dtype=bf, device=dev)
    lse = torch.zeros(B, H, Sq, device=dev)
    sc = D ** -0.5
    ops.attention_fwd(q.data_ptr(), C, Sq * C, k.data_ptr(), C, Skv * C, v.data_ptr(), C, Skv * C, o.data_ptr(), C,
                      Sq * C, lse, B, H, Sq, Skv, D, sc).run()
    dq, dk, dv = torch.zeros_like(q), torch.zeros_like(k), torch.zeros_like(v)
    delta = torch.zeros(B, H, Sq, device=dev)
    ops.attention_bwd(q.data_ptr(), C, Sq * C, k.data_ptr(), C, Skv * C, v.data_ptr(), C, Skv * C, o.data_ptr(), C,
                      Sq * C, do.data_ptr(), C, Sq * C, lse, delta, dq.data_ptr(), C, Sq * C, dk.data_ptr(), C,
                      Skv * C, dv.data_ptr(), C, Skv * C, B, H, Sq, Skv, D, sc).run()
    torch.cuda.synchronize()
    # fp32 reference one sample at a time (the S = 9216 score matrix of a whole batch would be 7 GB, twice with autograd)
    worst = {}
    for b in range(B):
        qq, kk, vv = [t[b].float().requires_grad_(True) for t in (q, k, v)]
        qh, kh, vh = [t.reshape(-1, H, D).transpose(0, 1) for t in (qq, kk, vv)]
        s = (qh @ kh.transpose(-1, -2)) * sc
        ref = (torch.softmax(s, -1) @ vh).transpose(0, 1).reshape(Sq, C)
        ref.backward(do[b].float())
        for name, got, want in (("o", o[b], ref.detach()), ("lse", lse[b], torch.logsumexp(s.detach(), -1)),
                                ("dq", dq[b], qq.grad), ("dk", dk[b], kk.grad), ("dv", dv[b], vv.grad)):
            worst[name] = max(worst.get(name, 0.0), rel_err(got, want))
        del s, ref, qq, kk, vv
    print(f"attention B={B} H={H} Sq={Sq} Skv={Skv} d={D}: " + " ".join(f"{n} {e:.2e}" for n, e in worst.items()))
    # d = 40 / 64 / 80 self-attention: the row sum is accumulated by an MFMA from the bf16-ROUNDED probabilities (the values the
    # output is accumulated from), the other kernels sum the fp32 ones
    assert worst["o"] < 1e-2 and worst["lse"] < (2e-4 if D in (40, 64, 80) else 1e-5)
    assert max(worst["dq"], worst["dk"], worst["dv"]) < 6e-3


# (act, B, HW, C0, C1): SD1.5 level 0 (one and two sources), level 1 skip concatenation, the batched frozen pass, SDXL 128^2
FULL_GN = [(1, 4, 4096, 320, 0), (1, 4, 4096, 320, 320), (1, 4, 1024, 640, 320), (0, 4, 1024, 640, 0), (1, 12, 4096, 320, 0),
           (1, 4, 64, 1280, 1280), (1, 2, 16384, 320, 0), (1, 4, 9216, 320, 320)]


@pytest.mark.gpu
@pytest.mark.parametrize("act,B,HW,C0,C1", FULL_GN)
def test_groupnorm_full_size_fwd_bwd_on_gpu(act, B, HW, C0, C1):
    dev = _gpu()
    torch.manual_seed(3)
    C, G = C0 + C1, 32
    x0 = torch.randn(B * HW, C0, device=dev).to(bf)
    x1 = (torch.randn(B * HW, C1, device=dev) * 2 + 0.5).to(bf) if C1 else None
    gamma, beta = torch.randn(C, device=dev), torch.randn(C, device=dev)
    stats = torch.zeros(B * G * 2 * 257, device=dev)
    y = torch.zeros(B * HW, C, dtype=bf, device=dev)
    ops.groupnorm_fwd(x0, C0, x1, C1, C0, gamma, beta, B, HW, C, G, 1e-5, act, stats, y, C).run()
    dy = torch.randn(B * HW, C, device=dev).to(bf)
    bstats = torch.zeros(B * G * 2 * 257, device=dev)
    dx = torch.zeros(B * HW, C, dtype=bf, device=dev)
    ops.groupnorm_bwd(x0, C0, x1, C1, C0, dy, C, gamma, beta, stats, B, HW, C, G, 1e-5, act, bstats, dx, C).run()
    torch.cuda.synchronize()
    xcat = torch.cat([x0, x1], 1) if C1 else x0
    xc = xcat.float().reshape(B, HW, C).permute(0, 2, 1).requires_grad_(True)
    ref = F.group_norm(xc, G, gamma, beta, 1e-5)
    ref = F.silu(ref) if act else ref
    ref.backward(dy.float().reshape(B, HW, C).permute(0, 2, 1))
    e_y = rel_err(y.reshape(B, HW, C).permute(0, 2, 1), ref.detach())
    e_dx = rel_err(dx.reshape(B, HW, C).permute(0, 2, 1), xc.grad)
    print(f"groupnorm act={act} B={B} HW={HW} C={C0}+{C1}: y {e_y:.2e} dx {e_dx:.2e}")
    assert e_y < 5e-3 and e_dx < 8e-3


@pytest.mark.gpu
@pytest.mark.parametrize("M,C", [(16384, 320), (49152, 320), (4096, 640), (1024, 1280), (36864, 320), (8192, 640)])
def test_layernorm_full_size_fwd_bwd_on_gpu(M, C):
    dev = _gpu()
    torch.manual_seed(4)
    x = (torch.randn(M, C, device=dev) * 2 + 0.3).to(bf)
    gamma, beta = torch.randn(C, device=dev), torch.randn(C, device=dev)
    y = torch.zeros(M, C, dtype=bf, device=dev)
    mean, rstd = torch.zeros(M, device=dev), torch.zeros(M, device=dev)
    ops.layernorm_fwd(x, C, gamma, beta, 1e-5, M, C, y, C, mean, rstd).run()
    dy = torch.randn(M, C, device=dev).to(bf)
    dx = torch.zeros(M, C, dtype=bf, device=dev)
    ops.layernorm_bwd(x, C, dy, C, gamma, mean, rstd, None, C, M, C, dx, C).run()
    torch.cuda.synchronize()
    xx = x.float().requires_grad_(True)
    ref = F.layer_norm(xx, (C,), gamma, beta, 1e-5)
    ref.backward(dy.float())
    assert rel_err(y, ref.detach()) < 5e-3 and rel_err(dx, xx.grad) < 8e-3
