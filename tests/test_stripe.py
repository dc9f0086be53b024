"""Row-stripe fused transformer-block kernels (csrc/stripe.hip) vs a plain fp32 PyTorch statement of the same chain of
diffusers operations (BasicTransformerBlock after its self-attention core + Transformer2DModel.proj_out; the head: GroupNorm,
proj_in, LayerNorm, q|k|v) on the same bf16-rounded operands.  Runs on the host emulator of the kernel sources (CPU tier,
both LDS-DMA landing models) and on gfx950 (`-m gpu`).

Tolerance: the kernel rounds the GEMM operands it hands from phase to phase (LayerNorm outputs, q2, a2, the GEGLU chunk, h3)
to bf16 exactly where the per-op kernels round their stored outputs; the reference rounds at the same points, so what is left
is accumulation order and the bf16 rounding of P inside the cross-attention: 3e-3 relative L2 on the bf16 output."""
import math
import os
import subprocess
import sys

import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err
from leco_amd import hip, ops

bf = torch.bfloat16
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _sync(dev):
    if dev.type == "cuda":
        torch.cuda.synchronize()


def _r(t):       # bf16 rounding, back in fp32
    return t.to(bf).float()


def geglu_perm(n):
    i = torch.arange(n)
    j, r = i // 128, i % 128
    return torch.where(r < 64, j * 64 + r, n // 2 + j * 64 + (r - 64))


class Lin:
    """A Linear with an optional rank-R LoRA in the packed form the kernels consume (dn_s [t_rows][K], up_p [N][32])."""

    def __init__(self, n, k, rank, dev, bias=True, wscale=1.0):
        self.w = (torch.randn(n, k) * wscale / k ** 0.5).to(bf)
        self.b = torch.randn(n) * 0.1 if bias else None
        self.rank = rank
        if rank:
            self.t_rows = 16 if rank <= 16 else 32
            dn = torch.zeros(self.t_rows, k)
            dn[:rank] = torch.randn(rank, k) / k ** 0.5
            up = torch.zeros(n, 32)
            up[:, :rank] = torch.randn(n, rank) * 0.05
            self.dn, self.up = dn.to(bf), up.to(bf)
        self.dev = dev

    def ref(self, x):
        y = x @ self.w.float().T
        if self.rank:
            y = y + _r(x @ self.dn.float().T) @ self.up.float().T[:self.t_rows]
        if self.b is not None:
            y = y + self.b
        return y

    def xlin(self, perm=None, packed=True):
        w, b = self.w, self.b
        up = self.up if self.rank else None
        if perm is not None:
            w, b = w[perm], (None if b is None else b[perm])
            up = None if up is None else up[perm]
        if packed:          # frozen weights in MFMA fragment order (what the plan builder hands the kernels)
            w = hip.pack_fragments(w)
        self._keep = (w.contiguous().to(self.dev), None if b is None else b.contiguous().to(self.dev),
                      self.dn.to(self.dev) if self.rank else None, None if up is None else up.contiguous().to(self.dev))
        w, b, dn, up = self._keep
        return hip.xlin(w, b, dn, up, self.t_rows if self.rank else 0, packed=packed)


def _ln(x, g, b, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), g, b, eps)


@pytest.mark.parametrize("heads,rank,proj_out,stats,B,hw", [
    (8, 4, True, True, 2, 64),       # SD1.x level 0 (d = 40), rank 4, Transformer2DModel tail incl. proj_out + GN statistics
    (8, 0, True, False, 1, 128),     # LoRA off (the frozen passes)
    (5, 24, True, True, 1, 64),      # SD2.x (d = 64); a 24-row LoRA stack (32 lora_down rows ride in the tiles)
    (8, 8, False, False, 3, 64),     # a block that is not the last of its Transformer2DModel: h3 is the output
])
def test_xblock_tail_matches_the_per_op_chain(dev, heads, rank, proj_out, stats, B, hw):
    torch.manual_seed(100 + heads + rank)
    C, skv = 320, 77
    D = C // heads
    M = B * hw
    to_out1, to_q2, to_out2 = Lin(C, C, rank, dev), Lin(C, C, rank, dev, bias=False), Lin(C, C, rank, dev)
    ff1, ff2, po = Lin(8 * C, C, rank, dev), Lin(C, 4 * C, rank, dev), Lin(C, C, rank, dev)
    a1 = torch.randn(M, C).to(bf); h0 = torch.randn(M, C).to(bf); x = torch.randn(M, C).to(bf)
    kv = torch.randn(B * skv, 2 * C).to(bf)
    g2, b2, g3, b3 = (1 + 0.2 * torch.randn(C)), 0.1 * torch.randn(C), (1 + 0.2 * torch.randn(C)), 0.1 * torch.randn(C)

    # ---- reference (fp32, rounding where the chain stores bf16 operands)
    h1 = to_out1.ref(a1.float()) + h0.float()
    q2 = _r(to_q2.ref(_r(_ln(h1, g2, b2))))
    kf, vf = kv.float()[:, :C].reshape(B, skv, heads, D), kv.float()[:, C:].reshape(B, skv, heads, D)
    qh = q2.reshape(B, hw, heads, D)
    s = torch.einsum("bqhd,bkhd->bhqk", qh, kf) * D ** -0.5
    a2 = _r(torch.einsum("bhqk,bkhd->bqhd", s.softmax(-1), vf).reshape(M, C))
    h2 = to_out2.ref(a2) + h1
    u = ff1.ref(_r(_ln(h2, g3, b3)))
    gg = _r(u[:, :4 * C] * F.gelu(u[:, 4 * C:]))
    h3 = ff2.ref(gg) + h2
    ref = po.ref(_r(h3)) + x.float() if proj_out else h3

    # ---- kernel
    d = lambda t: t.to(dev)
    a1d, h0d, xd, kvd = d(a1), d(h0), d(x), d(kv)
    kp, vt = ops.xattn_buffers(B, heads, D, dev)
    ops.xattn_prep(kvd.data_ptr(), 2 * C, kp, vt, B, heads, skv, D).run()
    out = torch.zeros(M, C, dtype=bf, device=dev)
    cst = torch.zeros(B, C // 10, 2, device=dev)
    A = hip.XBlockTailArgs()
    A.m, A.c, A.heads, A.skv, A.rows_per_sample = M, C, heads, skv, hw
    A.attn, A.ld_attn, A.h_in, A.ld_h = a1d.data_ptr(), C, h0d.data_ptr(), C
    pk = rank != 8           # (one case keeps row-major weights: both layouts are part of the ABI)
    A.to_out1, A.to_q2, A.to_out2 = to_out1.xlin(packed=pk), to_q2.xlin(packed=pk), to_out2.xlin(packed=pk)
    A.ff1, A.ff2 = ff1.xlin(geglu_perm(8 * C), packed=pk), ff2.xlin(packed=pk)
    if proj_out:
        A.proj_out = po.xlin(packed=pk)
        A.res, A.ld_res = xd.data_ptr(), C
    lnp = [d(t.float().contiguous()) for t in (g2, b2, g3, b3)]
    A.ln2_g, A.ln2_b, A.ln3_g, A.ln3_b = [t.data_ptr() for t in lnp]
    A.ln_eps = 1e-5
    A.kp, A.vt, A.attn_scale = kp.data_ptr(), vt.data_ptr(), D ** -0.5
    A.out, A.ld_out = out.data_ptr(), C
    if stats:
        A.col_stats, A.stats_atom = cst.data_ptr(), 10
    op = ops.xblock_tail(A)
    op.run()
    _sync(dev)
    assert torch.isfinite(out.float()).all()
    assert rel_err(out.cpu(), ref) < 3e-3, rel_err(out.cpu(), ref)
    if stats:
        o = out.float().cpu().reshape(B, hw, C // 10, 10)
        want = torch.stack([o.sum((1, 3)), (o * o).sum((1, 3))], -1)
        assert rel_err(cst.cpu(), want) < 1e-5


def test_xattn_prep_layout(dev):
    """kp / vt are what the tail kernel's fragment loads expect: zero padding, the permuted key order, the row of ones."""
    torch.manual_seed(3)
    B, heads, D, skv, C = 2, 8, 40, 77, 320
    kv = torch.randn(B * skv, 2 * C).to(bf).to(dev)
    kp, vt = ops.xattn_buffers(B, heads, D, dev)
    ops.xattn_prep(kv.data_ptr(), 2 * C, kp, vt, B, heads, skv, D).run()
    _sync(dev)
    kp = kp.float().cpu().reshape(B, heads, 80, 64); vt = vt.float().cpu().reshape(B, heads, 48, 96)
    kvc = kv.float().cpu()
    K = kvc[:, :C].reshape(B, skv, heads, D).permute(0, 2, 1, 3); V = kvc[:, C:].reshape(B, skv, heads, D).permute(0, 2, 1, 3)
    assert torch.equal(kp[:, :, :skv, :D], K) and kp[:, :, skv:].abs().max() == 0 and kp[:, :, :, D:].abs().max() == 0
    pos = torch.arange(96)
    s_, g_, t_ = pos >> 5, (pos >> 3) & 3, pos & 7
    key = 32 * s_ + 16 * (t_ >> 2) + 4 * g_ + (t_ & 3)
    ok = key < skv
    assert torch.equal(vt[:, :, :D][..., ok], V.permute(0, 1, 3, 2)[..., key[ok]])
    assert vt[..., ~ok].abs().max() == 0
    assert torch.equal(vt[:, :, D][..., ok], torch.ones(B, heads, int(ok.sum()))) and vt[:, :, D + 1:].abs().max() == 0


def test_stripe_dma_protocol_under_late_completion():
    """The weight-tile ring of the stripe kernels with the emulator's LATE LDS-DMA model (a copy lands only at the issuing
    lane's counted s_waitcnt vmcnt or a draining barrier): a mis-counted wait reads a stale ring slot and fails parity."""
    env = dict(os.environ, LECO_EMU_DMA="late")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", os.path.abspath(__file__), "-k",
                        "matches_the_per_op_chain and emu and 8-4-True"], env=env, capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


def _stripe_unet(dev, heads=8):
    from leco_amd import model_util
    from leco_amd.unet import UNet2DConditionModel, UNetConfig
    cfg = UNetConfig(block_out_channels=(320, 320), down_block_types=("CrossAttnDownBlock2D", "DownBlock2D"),
                     up_block_types=("UpBlock2D", "CrossAttnUpBlock2D"), layers_per_block=1, attention_head_dim=heads,
                     cross_attention_dim=64, sample_size=8)
    m = model_util.init_synthetic_(UNet2DConditionModel(cfg), seed=7).to(dev, bf)
    m.requires_grad_(False)
    return m


def _run_plan(m, plan, which, x, ctx, t=500.0):
    plan.x_in.copy_(x)
    plan.set_ctx(ctx)
    plan.t_table[:1].fill_(t)
    plan.t_idx.zero_()
    m._run(plan, which)
    return plan.pred.float().cpu().clone()


@pytest.mark.parametrize("heads,rank", [(8, 4), (5, 0)])
def test_forward_only_plans_run_the_stripe_kernels_and_match_the_per_op_plan(dev, heads, rank, monkeypatch):
    """A UNet whose 64-pixel level has 320 channels: the forward-only plan (what the denoising passes and the batched
    frozen pass replay) runs each transformer block's tail as `leco_xblock_tail`; its prediction equals the per-op plan's
    (LECO_STRIPE=0) to bf16 rounding -- LoRA on and off."""
    import contextlib
    import io
    from leco_amd.lora import LoRANetwork
    torch.manual_seed(5)
    m = _stripe_unet(dev, heads)
    if rank:
        with contextlib.redirect_stdout(io.StringIO()):
            net = LoRANetwork(m, rank=rank, multiplier=1.0, alpha=1.0)
        with torch.no_grad():
            for l in net.unet_loras:
                l.lora_up.weight.normal_(0, 0.02)
        net.mark_updated()
    B, h, w = 2, 8, 8
    x = torch.randn(B, 4, h, w).to(dev, bf); ctx = torch.randn(B, 77, 64).to(dev, bf)
    eng = m.engine()
    if rank:
        net.multiplier = 1.0
        m.prepare((B, 4, h, w), lora_on=True)
    fused = eng.plan(B, h, w, need_bwd=False)
    names = [op.name for op in fused.lists["fwd_on"]]
    assert names.count("leco_xblock_tail") == 3 and names.count("leco_xattn_prep") == 3 and names.count("leco_xblock_head") == 3
    # every LayerNorm of the 3 fused blocks lives in the stripe kernels; the mid block (16 pixels) keeps its per-op launches
    assert names.count("leco_layernorm_fwd") == 3
    y_on = _run_plan(m, fused, "fwd_on", x, ctx)
    y_off = _run_plan(m, fused, "fwd_off", x, ctx)
    monkeypatch.setenv("LECO_STRIPE", "0")
    eng.plans.clear()
    plain = eng.plan(B, h, w, need_bwd=False)
    assert "leco_xblock_tail" not in [op.name for op in plain.lists["fwd_on"]]
    p_on = _run_plan(m, plain, "fwd_on", x, ctx)
    p_off = _run_plan(m, plain, "fwd_off", x, ctx)
    _sync(dev)
    e_on, e_off = rel_err(y_on, p_on), rel_err(y_off, p_off)
    print(f"stripe vs per-op plan: LoRA on {e_on:.3g}, off {e_off:.3g}; on-vs-off {rel_err(p_on, p_off):.3g}")
    assert e_on < 1e-2 and e_off < 1e-2
    if rank:
        assert rel_err(p_on, p_off) > 1e-3 and rel_err(y_on, y_off) > 1e-3     # the LoRA term is really there


@pytest.mark.parametrize("rank,gn,B,hw", [(4, True, 2, 64), (0, False, 1, 128), (8, True, 1, 64)])
def test_xblock_head_matches_the_per_op_chain(dev, rank, gn, B, hw):
    """GroupNorm (from producer statistics) -> proj_in -> LayerNorm -> q|k|v: h_out and qkv_out vs the fp32 chain."""
    torch.manual_seed(200 + rank)
    C, G = 320, 32
    M = B * hw
    pin = Lin(C, C, rank, dev)
    qkv = Lin(3 * C, C, 3 * rank, dev, bias=False)        # 3 groups of rank `rank`, stacked (block structure irrelevant here)
    x = (torch.randn(M, C) * 1.5 + 0.3).to(bf)
    gg, gb = 1 + 0.2 * torch.randn(C), 0.1 * torch.randn(C)
    lg, lb = 1 + 0.2 * torch.randn(C), 0.1 * torch.randn(C)
    xf = x.float()
    if gn:
        n = _r(F.group_norm(xf.reshape(B, hw, C).permute(0, 2, 1), G, gg, gb, 1e-6).permute(0, 2, 1).reshape(M, C))
    else:
        n = xf
    p = pin.ref(n)
    q = qkv.ref(_r(_ln(p, lg, lb)))
    d = lambda t: t.to(dev)
    xd = d(x)
    h_out = torch.zeros(M, C, dtype=bf, device=dev); qkv_out = torch.zeros(M, 3 * C, dtype=bf, device=dev)
    A = hip.XBlockHeadArgs()
    A.m, A.c, A.rows_per_sample = M, C, hw
    A.x, A.ld_x = xd.data_ptr(), C
    keep = [d(t.float().contiguous()) for t in (gg, gb, lg, lb)]
    if gn:
        xs = xf.reshape(B, hw, C // 10, 10)
        cst = d(torch.stack([xs.sum((1, 3)), (xs * xs).sum((1, 3))], -1).contiguous())
        A.gn_cstats, A.stats_atom, A.groups = cst.data_ptr(), 10, G
        A.gn_g, A.gn_b, A.gn_eps = keep[0].data_ptr(), keep[1].data_ptr(), 1e-6
    A.proj_in, A.qkv = pin.xlin(packed=rank != 8), qkv.xlin(packed=rank != 8)
    A.ln1_g, A.ln1_b, A.ln_eps = keep[2].data_ptr(), keep[3].data_ptr(), 1e-5
    A.h_out, A.ld_hout, A.qkv_out, A.ld_qkv = h_out.data_ptr(), C, qkv_out.data_ptr(), 3 * C
    ops.xblock_head(A).run()
    _sync(dev)
    e_h, e_q = rel_err(h_out.cpu(), p), rel_err(qkv_out.cpu(), q)
    assert e_h < 3e-3 and e_q < 3e-3, (e_h, e_q)


@pytest.mark.parametrize("share", [2, 6])
def test_batch_shared_prefix_gives_the_same_prediction(dev, share):
    """`Engine.plan(..., share=n)`: the latents are n copies of B / n samples (predict_noise's cat([latents] * 2),
    train_util.py:151; the batched frozen pass: 6 copies), so conv_in, the first ResnetBlock2D and the first transformer up to
    its self-attention run once per distinct sample and the stripe tail kernel re-expands the batch.  Same arithmetic per
    sample: the prediction equals the unshared plan's (up to the run-to-run noise of atomically accumulated GroupNorm
    statistics, absent at this size)."""
    torch.manual_seed(9)
    m = _stripe_unet(dev, 8)
    bs, h, w = 1, 8, 8
    B = share * bs
    x1 = torch.randn(bs, 4, h, w).to(bf)
    x = x1.repeat(share, 1, 1, 1).to(dev)
    ctx = torch.randn(B, 77, 64).to(dev, bf)          # a different prompt per copy: the tail must use the right one
    eng = m.engine()
    shared = eng.plan(B, h, w, need_bwd=False, share=share)
    plain = eng.plan(B, h, w, need_bwd=False)
    n_s, n_p = [op.name for op in shared.lists["fwd_off"]], [op.name for op in plain.lists["fwd_off"]]
    assert n_s.count("leco_repeat") >= 1 and "leco_repeat" not in n_p and len(n_s) <= len(n_p) + 2
    # the shared plan's prefix tensors hold B / share samples
    assert shared.bufs["conv_in"].shape[0] == bs * h * w and plain.bufs["conv_in"].shape[0] == B * h * w
    y_s = _run_plan(m, shared, "fwd_off", x, ctx)
    y_p = _run_plan(m, plain, "fwd_off", x, ctx)
    _sync(dev)
    assert rel_err(y_s, y_p) < 1e-6, rel_err(y_s, y_p)
    assert rel_err(y_s[0], y_s[1]) > 1e-3          # (the copies really differ through their prompts)


@pytest.mark.parametrize("rank,linear", [(4, False), (0, True)])
def test_forward_only_plans_run_the_a_stationary_gemm_and_match_the_ring_gemm(dev, rank, linear, monkeypatch):
    """A UNet with 320- and 640-channel transformer levels below the stripe kernels' reach (LECO_STRIPE=0 keeps level 0 on the
    per-op chain too): the forward-only plan sends attn{1,2}.to_q / to_out.0, q|k|v, proj_in / proj_out and the 1x1
    shortcuts to `leco_xgemm` (csrc/xgemm.hip, K = 320 / 640); its prediction equals the LDS-ring GEMM plan's (LECO_XGEMM=0)
    to bf16 rounding, LoRA on and off -- conv and Linear flavours of proj_in / proj_out."""
    import contextlib
    import io
    from leco_amd import model_util
    from leco_amd.lora import LoRANetwork
    from leco_amd.unet import UNet2DConditionModel, UNetConfig
    torch.manual_seed(9)
    monkeypatch.setenv("LECO_STRIPE", "0")
    monkeypatch.setenv("LECO_XGEMM_MAX_M", "4096")      # (default 256 rows: where the kernel measured faster on MI355X)
    cfg = UNetConfig(block_out_channels=(320, 640), down_block_types=("CrossAttnDownBlock2D", "CrossAttnDownBlock2D"),
                     up_block_types=("CrossAttnUpBlock2D", "CrossAttnUpBlock2D"), layers_per_block=1, attention_head_dim=8,
                     cross_attention_dim=64, sample_size=8, use_linear_projection=linear)
    m = model_util.init_synthetic_(UNet2DConditionModel(cfg), seed=3).to(dev, bf)
    m.requires_grad_(False)
    if rank:
        with contextlib.redirect_stdout(io.StringIO()):
            net = LoRANetwork(m, rank=rank, multiplier=1.0, alpha=1.0)
        with torch.no_grad():
            for l in net.unet_loras:
                l.lora_up.weight.normal_(0, 0.02)
        net.mark_updated()
    B, h, w = 2, 8, 8
    x = torch.randn(B, 4, h, w).to(dev, bf); ctx = torch.randn(B, 77, 64).to(dev, bf)
    eng = m.engine()
    if rank:
        net.multiplier = 1.0
        m.prepare((B, 4, h, w), lora_on=True)
    xp = eng.plan(B, h, w, need_bwd=False)
    names = [op.name for op in xp.lists["fwd_on"]]
    n_x = names.count("leco_xgemm")
    # the four 640-channel transformers (down 1, mid, up 0 x 2): q|k|v, to_out.0, to_q, to_out.0, proj_in, proj_out each
    # (the 320-channel level has N = 320, not a multiple of the 128-column tile: it stays on the ring GEMM / the stripe kernels)
    assert n_x >= 4 * 6, names
    y_on, y_off = _run_plan(m, xp, "fwd_on", x, ctx), _run_plan(m, xp, "fwd_off", x, ctx)
    # the training plan (it must stash T for the LoRA weight gradients) keeps the ring GEMM
    assert "leco_xgemm" not in [op.name for op in eng.plan(B, h, w).lists["fwd_on"]]
    monkeypatch.setenv("LECO_XGEMM", "0")
    eng.plans.clear()
    rp = eng.plan(B, h, w, need_bwd=False)
    assert "leco_xgemm" not in [op.name for op in rp.lists["fwd_on"]]
    p_on, p_off = _run_plan(m, rp, "fwd_on", x, ctx), _run_plan(m, rp, "fwd_off", x, ctx)
    _sync(dev)
    e_on, e_off = rel_err(y_on, p_on), rel_err(y_off, p_off)
    print(f"xgemm vs ring-gemm plan ({n_x} launches moved): LoRA on {e_on:.3g}, off {e_off:.3g}; on-vs-off {rel_err(p_on, p_off):.3g}")
    assert e_on < 1e-2 and e_off < 1e-2
    if rank:
        assert rel_err(p_on, p_off) > 1e-3 and rel_err(y_on, y_off) > 1e-3     # the LoRA term is really there


