"""Per-kernel parity: every C-ABI kernel vs a plain fp32 PyTorch statement of the same op on
the same bf16-rounded inputs.  Runs twice: on the host emulator of the kernel sources (CPU
tier) and on the gfx950 build (`-m gpu`).  Tolerances (relative L2):
  1e-5  where the result is compared in fp32 (MFMA fp32 accumulate vs torch fp32),
  2e-3  where the kernel rounds its result (or P / dS inside attention) to bf16 -- one bf16
        rounding is 2^-9 ~ 2e-3 worst case, ~1.1e-3 RMS -- compared against the un-rounded fp32."""
import math

import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err
from leco_amd import hip, ops

bf = torch.bfloat16
TOL32, TOLBF = 1e-5, 3e-3


def _sync(dev):
    if dev.type == "cuda":
        torch.cuda.synchronize()


@pytest.mark.parametrize("tile,M,N,K", [(3, 100, 72, 128), (1, 300, 136, 192), (2, 200, 320, 64), (0, 70, 64, 64),
                                        (1, 128, 128, 64), (2, 256, 160, 128), (11, 300, 136, 192), (11, 128, 64, 64)])
def test_gemm_plain_full_epilogue(dev, tile, M, N, K):
    torch.manual_seed(0)
    a = torch.randn(M, K).to(bf).to(dev); w = (torch.randn(N, K) / K ** 0.5).to(bf).to(dev)
    ae = torch.randn(M, 32).to(bf).to(dev); we = (torch.randn(N, 32) * 0.1).to(bf).to(dev)
    bias = torch.randn(N).to(dev); rb = torch.randn((M + 49) // 50, N).to(dev); res = torch.randn(M, N).to(bf).to(dev)
    out = torch.zeros(M, N, dtype=bf, device=dev); o32 = torch.zeros(M, N, device=dev)
    g = hip.gemm_args(a, w, out, m=M, n=N, k=K, a_ext=ae, w_ext=we, ext_k=32, bias=bias, rowbias=rb,
                      rows_per_group=50, residual=res, act=hip.ACT_SILU, out_f32=o32)
    hip.gemm(g, ops.default_stream(), tile)
    _sync(dev)
    ref = a.float() @ w.float().T + ae.float() @ we.float().T + bias + rb.repeat_interleave(50, 0)[:M] + res.float()
    ref = F.silu(ref)
    assert rel_err(o32, ref) < TOL32
    assert rel_err(out, ref) < TOLBF


@pytest.mark.parametrize("tile,split", [(1, 3), (3, 4), (2, 2), (0, 0), (4, 2), (4, 1), (11, 2)])
def test_gemm_split_k_with_epilogue_and_lora_tile(dev, tile, split):
    torch.manual_seed(11)
    M, N, K = 200, 320, 1152
    a = torch.randn(M, K).to(bf).to(dev); w = (torch.randn(N, K) / K ** 0.5).to(bf).to(dev)
    ae = torch.randn(M, 64).to(bf).to(dev); we = (torch.randn(N, 64) * 0.1).to(bf).to(dev)
    bias = torch.randn(N).to(dev); res = torch.randn(M, N).to(bf).to(dev)
    out = torch.zeros(M, N, dtype=bf, device=dev); o32 = torch.zeros(M, N, device=dev)
    ws = torch.empty(4 * M * N, device=dev)
    g = hip.gemm_args(a, w, out, m=M, n=N, k=K, a_ext=ae, w_ext=we, ext_k=64, bias=bias, residual=res, out_f32=o32)
    hip.gemm(g, ops.default_stream(), tile, split, ws)
    _sync(dev)
    ref = a.float() @ w.float().T + ae.float() @ we.float().T + bias + res.float()
    assert rel_err(o32, ref) < TOL32 and rel_err(out, ref) < TOLBF


@pytest.mark.parametrize("tile,t_rows,split,M,N,K", [
    (1, 16, 1, 300, 256, 320), (1, 32, 1, 300, 256, 320), (2, 16, 1, 200, 320, 192), (2, 32, 1, 200, 320, 192),
    (3, 16, 1, 77, 64, 128), (3, 32, 1, 150, 72, 128), (4, 16, 1, 520, 256, 256), (4, 32, 1, 300, 128, 256),
    (11, 16, 1, 300, 136, 320), (11, 32, 1, 300, 128, 256),      # 128 x 64 tile (round 6)
    (1, 16, 1, 2600, 640, 128),      # > 384 workgroups: the 2-buffer variant
    (0, 16, 0, 256, 256, 2048),      # heuristic wants split-K: falls back to the separate projection
    (0, 32, 0, 130, 200, 64)])
def test_gemm_fused_lora_down_projection(dev, tile, t_rows, split, M, N, K):
    """t_w: T = A t_w^T is formed inside the main GEMM's K sweep (bf16-rounded like the separate T GEMM),
    used as the A side of the K-extension against scale*up, and optionally written out for the backward."""
    torch.manual_seed(13)
    R = 12 if t_rows == 16 else 24
    a0 = torch.randn(M, 64).to(bf).to(dev); a1 = torch.randn(M, K - 64).to(bf).to(dev)
    w = (torch.randn(N, K) / K ** 0.5).to(bf).to(dev)
    tw = torch.zeros(32, K); tw[:R] = torch.randn(R, K) / K ** 0.5
    tw = tw.to(bf).to(dev)
    up = torch.zeros(N, 32); up[:, :R] = torch.randn(N, R) * 0.3
    up = up.to(bf).to(dev)
    bias = torch.randn(N).to(dev); res = torch.randn(M, N).to(bf).to(dev)
    out = torch.zeros(M, N, dtype=bf, device=dev); o32 = torch.zeros(M, N, device=dev)
    tout = torch.full((M, 32), 7.0, dtype=bf, device=dev)
    ws = torch.empty(8 * M * N, device=dev)
    g = hip.gemm_args(a0, w, out, m=M, n=N, k=K, lda=64, a1=a1, lda1=K - 64, k_split=64, w_ext=up, ext_k=32,
                      bias=bias, residual=res, out_f32=o32, t_w=tw, t_rows=t_rows, t_out=tout)
    hip.gemm(g, ops.default_stream(), tile, split, ws)
    _sync(dev)
    a = torch.cat([a0, a1], 1).float()
    T = (a @ tw.float().T).to(bf)
    ref = a @ w.float().T + T.float() @ up.float().T + bias + res.float()
    assert rel_err(o32, ref) < TOL32 and rel_err(out, ref) < TOLBF
    assert rel_err(tout[:, :R], T[:, :R]) < 1e-2 and float(tout[:, R:].float().abs().max()) == 0.0
    # without t_out (LoRA-on passes that are never differentiated)
    o32.zero_()
    g2 = hip.gemm_args(a0, w, None, m=M, n=N, k=K, lda=64, a1=a1, lda1=K - 64, k_split=64, w_ext=up, ext_k=32,
                       bias=bias, residual=res, out_f32=o32, t_w=tw, t_rows=t_rows)
    hip.gemm(g2, ops.default_stream(), tile, 1, None)
    _sync(dev)
    assert rel_err(o32, ref) < TOL32


@pytest.mark.parametrize("tile,t_rows,M,N,K", [(3, 16, 1024, 3840, 1280), (3, 16, 1024, 1280, 320), (1, 16, 4096, 1920, 640),
                                               (2, 32, 2048, 640, 640), (4, 16, 4096, 1280, 1280)])
def test_gemm_fused_lora_full_grid_deep_k_is_race_free(dev, tile, t_rows, M, N, K):
    """Grids that put two workgroups on a CU, K deep enough for the steady-state loop, repeated launches: the shapes
    the full-size UNet runs and the tiny tests never reached (round-1 NaN: proj_in 1024 x 1280 x 1280 on the 64 x 64
    fused-down-projection tile, wrong on a fraction of the launches only).  The emulator tier checks a cut-down grid."""
    torch.manual_seed(15)
    if dev.type == "cpu":
        M, N = min(M, 256), min(N, 384)
    R = 12 if t_rows == 16 else 24
    a = torch.randn(M, K).to(bf).to(dev); w = (torch.randn(N, K) / K ** 0.5).to(bf).to(dev)
    tw = torch.zeros(32, K); tw[:R] = torch.randn(R, K) / K ** 0.5
    tw = tw.to(bf).to(dev)
    up = torch.zeros(N, 32); up[:, :R] = torch.randn(N, R) * 0.3
    up = up.to(bf).to(dev)
    T = (a.float() @ tw.float().T).to(bf)
    ref = a.float() @ w.float().T + T.float() @ up.float().T
    for rep in range(6 if dev.type == "cuda" else 1):
        out = torch.zeros(M, N, dtype=bf, device=dev); tout = torch.zeros(M, 32, dtype=bf, device=dev)
        g = hip.gemm_args(a, w, out, m=M, n=N, k=K, w_ext=up, ext_k=32, t_w=tw, t_rows=t_rows, t_out=tout)
        hip.gemm(g, ops.default_stream(), tile, 1, None)
        _sync(dev)
        assert torch.isfinite(out.float()).all(), rep
        assert (out.float() - ref).abs().max().item() < 0.05 * ref.abs().max().item(), rep
        assert rel_err(out, ref) < TOLBF, rep


@pytest.mark.parametrize("tile,M,F,K,lora", [(0, 300, 128, 192, False), (1, 200, 256, 320, True), (4, 520, 128, 128, True)])
def test_gemm_fused_geglu_epilogue(dev, tile, M, F, K, lora):
    """LECO_ACT_GEGLU: interleaved value / gate weight rows, value * gelu(gate) written as [M][F] (diffusers GEGLU)."""
    torch.manual_seed(14)
    N = 2 * F
    a = torch.randn(M, K).to(bf).to(dev); w = (torch.randn(N, K) / K ** 0.5).to(bf); bias = torch.randn(N)
    i = torch.arange(N); j, r = i // 128, i % 128
    perm = torch.where(r < 64, j * 64 + r, F + j * 64 + (r - 64))
    wg = w[perm].contiguous().to(dev); bg = bias[perm].contiguous().to(dev)
    out = torch.zeros(M, F, dtype=bf, device=dev)
    kw = {}
    ref_u = a.float().cpu() @ w.float().T + bias   # reference on the host (w, bias were never moved)
    if lora:
        R = 12
        tw = torch.zeros(32, K); tw[:R] = torch.randn(R, K) / K ** 0.5
        up = torch.zeros(N, 32); up[:, :R] = torch.randn(N, R) * 0.3
        tw, up = tw.to(bf), up.to(bf)
        kw = dict(w_ext=up[perm].contiguous().to(dev), ext_k=32, t_w=tw.to(dev), t_rows=16,
                  t_out=torch.zeros(M, 32, dtype=bf, device=dev))
        T = (a.float().cpu() @ tw.float().T).to(bf)
        ref_u = ref_u + T.float() @ up.float().T
    g = hip.gemm_args(a, wg, out, m=M, n=N, k=K, bias=bg, act=hip.ACT_GEGLU, ldc=F, **kw)
    hip.gemm(g, ops.default_stream(), tile)
    _sync(dev)
    ref = ref_u[:, :F] * F_gelu(ref_u[:, F:])
    assert rel_err(out.cpu(), ref) < TOLBF
    with pytest.raises(hip.LecoError):   # needs a 128-column tile
        hip.gemm(g, ops.default_stream(), 2)


def F_gelu(x):
    return F.gelu(x)


def test_gemm_large_grid_two_buffer_variant(dev):
    """> 384 workgroups selects the 4-wave / 2-buffer pipeline (the small cases use the 8-wave / 4-deep ring)."""
    torch.manual_seed(12)
    M, N, K = 2560, 2560, 128
    a = torch.randn(M, K).to(bf).to(dev); w = (torch.randn(N, K) / K ** 0.5).to(bf).to(dev)
    out = torch.zeros(M, N, dtype=bf, device=dev)
    hip.gemm(hip.gemm_args(a, w, out, m=M, n=N, k=K), ops.default_stream(), 1)
    _sync(dev)
    assert rel_err(out, a.float() @ w.float().T) < TOLBF


def test_gemm_mfma_layout_asymmetric(dev):
    """A = I-like / asymmetric-B check (cdna_hip_programming.md: transposes must be caught)."""
    M = N = K = 64
    a = torch.eye(M).to(bf).to(dev)
    w = (torch.arange(N)[:, None] * 0.5 - torch.arange(K)[None, :] * 0.25).to(bf).to(dev)  # w[n][k]
    o32 = torch.zeros(M, N, device=dev)
    hip.gemm(hip.gemm_args(a, w, None, m=M, n=N, k=K, out_f32=o32), ops.default_stream())
    _sync(dev)
    assert torch.equal(o32, w.float().T)


def test_gemm_two_source(dev):
    torch.manual_seed(1)
    M, N, K = 130, 64, 192
    a0 = torch.randn(M, 64).to(bf).to(dev); a1 = torch.randn(M, 128).to(bf).to(dev)
    w = (torch.randn(N, K) / K ** 0.5).to(bf).to(dev); o32 = torch.zeros(M, N, device=dev)
    hip.gemm(hip.gemm_args(a0, w, None, m=M, n=N, k=K, lda=64, a1=a1, lda1=128, k_split=64, out_f32=o32),
             ops.default_stream())
    _sync(dev)
    assert rel_err(o32, torch.cat([a0, a1], 1).float() @ w.float().T) < TOL32


@pytest.mark.parametrize("split,stats", [(2, False), (5, False), (8, True), (11, False), (6, True)])
def test_gemm_splitk_finish_sums_every_slab(dev, split, stats):
    """split-K partial slabs + the finishing kernels (plain and GroupNorm-statistics form): the slab loop keeps four loads in
    flight (round 6) -- every slab count around the unroll boundary gives the single-launch result."""
    torch.manual_seed(split)
    M, N, K = 130, 128, 64 * 12
    a = torch.randn(M, K).to(bf).to(dev); w = (torch.randn(N, K) / K ** 0.5).to(bf).to(dev)
    bias = torch.randn(N).to(dev); res = torch.randn(M, N).to(bf).to(dev)
    ws = torch.zeros(16 * M * N, device=dev)
    outs = []
    for sp in (1, split):
        out = torch.zeros(M, N, dtype=bf, device=dev)
        kw = {}
        cs = None
        if stats:
            cs = torch.zeros(1, N // 4, 2, device=dev)
            kw = dict(col_stats=cs, stats_rows=M, stats_atom=4)
        g = hip.gemm_args(a, w, out, m=M, n=N, k=K, bias=bias, residual=res, act=hip.ACT_SILU, **kw)
        hip.gemm(g, ops.default_stream(), tile=1, split_k=sp, ws=ws)
        _sync(dev)
        outs.append((out.float().cpu(), None if cs is None else cs.cpu().clone()))
    ref = F.silu(a.float().cpu() @ w.float().cpu().T + bias.cpu() + res.float().cpu())
    assert rel_err(outs[1][0], ref) < TOLBF
    assert rel_err(outs[1][0], outs[0][0]) < 3e-3          # same sums up to the fp32 order of the slabs
    if stats:
        assert rel_err(outs[1][1], _col_stats_ref(outs[1][0].to(bf), 1, M, 4)) < 1e-5


def test_gemm_rejects_bad_shapes(dev):
    a = torch.zeros(8, 40, dtype=bf, device=dev)
    with pytest.raises(hip.LecoError):
        hip.gemm(hip.gemm_args(a, a, a, m=8, n=8, k=40), ops.default_stream())


@pytest.mark.parametrize("mode", ["s1", "s2", "up2", "tr2", "s1_dgrad", "concat"])
def test_conv3x3_modes(dev, mode):
    torch.manual_seed(2)
    B, H, W_, Ci, Co = 2, 6, 10, 64, 64
    x = torch.randn(B, Ci, H, W_).to(bf); wt = (torch.randn(Co, Ci, 3, 3) / (9 * Ci) ** 0.5).to(bf)
    xh = x.permute(0, 2, 3, 1).contiguous().to(dev)
    wh = wt.permute(0, 2, 3, 1).contiguous().reshape(Co, 9 * Ci).to(dev)
    wtr = wt.flip(2, 3).permute(1, 2, 3, 0).contiguous().reshape(Ci, 9 * Co).to(dev)

    def run(amode, src, w, cin, cout, ho, wo, hin, win, **kw):
        o32 = torch.zeros(B * ho * wo, cout, device=dev)
        g = hip.gemm_args(src, w, None, m=B * ho * wo, n=cout, k=9 * cin, lda=kw.pop("lda", cin), a_mode=amode,
                          conv=(B, ho, wo, hin, win), out_f32=o32, **kw)
        hip.gemm(g, ops.default_stream())
        _sync(dev)
        return o32.reshape(B, ho, wo, cout).permute(0, 3, 1, 2).cpu()

    xf, wf = x.float(), wt.float()
    if mode == "s1":
        got, ref = run(hip.A_CONV3_S1, xh, wh, Ci, Co, H, W_, H, W_), F.conv2d(xf, wf, padding=1)
    elif mode == "s2":
        got, ref = run(hip.A_CONV3_S2, xh, wh, Ci, Co, H // 2, W_ // 2, H, W_), F.conv2d(xf, wf, padding=1, stride=2)
    elif mode == "up2":
        got = run(hip.A_CONV3_UP2, xh, wh, Ci, Co, 2 * H, 2 * W_, H, W_)
        ref = F.conv2d(F.interpolate(xf, scale_factor=2.0, mode="nearest"), wf, padding=1)
    elif mode in ("tr2", "s1_dgrad"):
        stride = 2 if mode == "tr2" else 1
        dy = torch.randn(B, Co, H // stride, W_ // stride).to(bf)
        xx = xf.clone().requires_grad_(True)
        F.conv2d(xx, wf, padding=1, stride=stride).backward(dy.float())
        dyh = dy.permute(0, 2, 3, 1).contiguous().to(dev)
        amode = hip.A_CONV3_TR2 if mode == "tr2" else hip.A_CONV3_S1
        got, ref = run(amode, dyh, wtr, Co, Ci, H, W_, H // stride, W_ // stride), xx.grad
    else:
        x2 = torch.randn(B, 128, H, W_).to(bf); wt2 = (torch.randn(Co, 192, 3, 3) / (9 * 192) ** 0.5).to(bf)
        x2h = x2.permute(0, 2, 3, 1).contiguous().to(dev)
        wh2 = wt2.permute(0, 2, 3, 1).contiguous().reshape(Co, 9 * 192).to(dev)
        got = run(hip.A_CONV3_S1, xh, wh2, 192, Co, H, W_, H, W_, lda=64, a1=x2h, lda1=128, k_split=64)
        ref = F.conv2d(torch.cat([x, x2], 1).float(), wt2.float(), padding=1)
    assert rel_err(got, ref) < TOL32


# (tile id, B, H, W, Cin, Cout, two-source split, split_k): TW = 16 and TW = 8 geometries, tiles that span several
# images (the zero separator rows of the virtual row space), ragged rows / columns / output channels, the padding piece
# of the 160-column weight tile, a channel range from two tensors, K slices
PATCH_CASES = [
    (7, 1, 16, 16, 64, 128, 0, 1),      # one 16x16 tile = one image
    (7, 3, 8, 8, 128, 136, 0, 1),       # TW = 8: a 32-row tile over three 8x8 images (+ ragged rows, ragged n)
    (7, 2, 24, 24, 64, 64, 0, 1),       # W % 16 != 0 -> TW = 8; tiles cross the image boundary mid-tile
    (8, 2, 6, 10, 64, 320, 0, 1),       # 128x160: ragged columns (W = 10 in 16-wide tiles), 8-row tiles over 2 images
    (8, 1, 32, 32, 192, 160, 64, 1),    # two-source channel range (64 + 128), three chunks
    (9, 2, 12, 20, 128, 64, 0, 2),      # 128x128, split-K over the two chunks
    (10, 1, 20, 16, 256, 200, 128, 2),  # 256x160 (3-slot ring), two sources, split-K, ragged n
    (7, 4, 5, 7, 64, 64, 0, 1),         # odd sizes: many tiny images per tile
]


@pytest.mark.parametrize("tile,B,H,W_,Ci,Co,ks,split", PATCH_CASES)
def test_conv3x3_patch_staged(dev, tile, B, H, W_, Ci, Co, ks, split):
    """conv_patch.hip (tile ids 7..10 of leco_gemm_ex) vs F.conv2d on the same bf16-rounded operands, with the full
    epilogue (bias + per-sample row bias + residual + SiLU, bf16 and fp32 outputs)."""
    torch.manual_seed(tile * 100 + H)
    x = torch.randn(B, Ci, H, W_).to(bf)
    wt = (torch.randn(Co, Ci, 3, 3) / (9 * Ci) ** 0.5).to(bf)
    bias, rowb = torch.randn(Co), torch.randn(B, Co)
    res = torch.randn(B, Co, H, W_).to(bf)
    xh = x.permute(0, 2, 3, 1).contiguous().to(dev)
    wh = wt.permute(0, 2, 3, 1).contiguous().reshape(Co, 9 * Ci).to(dev)
    resh = res.permute(0, 2, 3, 1).contiguous().reshape(B * H * W_, Co).to(dev)
    M = B * H * W_
    out = torch.zeros(M, Co, dtype=bf, device=dev)
    o32 = torch.zeros(M, Co, device=dev)
    kw = {}
    if ks:       # channels [0, ks) from one tensor, the rest from another (skip-connection concat)
        x0h = x[:, :ks].permute(0, 2, 3, 1).contiguous().to(dev)
        x1h = x[:, ks:].permute(0, 2, 3, 1).contiguous().to(dev)
        kw = dict(lda=ks, a1=x1h, lda1=Ci - ks, k_split=ks)
        xh = x0h
    bias_d, rowb_d = bias.to(dev), rowb.to(dev)      # (the argument block only holds raw pointers: keep them alive)
    g = hip.gemm_args(xh, wh, out, m=M, n=Co, k=9 * Ci, a_mode=hip.A_CONV3_S1, conv=(B, H, W_, H, W_), out_f32=o32,
                      bias=bias_d, rowbias=rowb_d, rows_per_group=H * W_, residual=resh, act=hip.ACT_SILU,
                      **{"lda": Ci, **kw})
    desc = hip.gemm_describe(g, tile, split, None, 0)
    assert "conv_patch_kernel" in desc, desc
    ws = torch.zeros(split * M * Co, device=dev) if split > 1 else None
    hip.gemm(g, ops.default_stream(), tile=tile, split_k=split, ws=ws)
    _sync(dev)
    ref = F.conv2d(x.float(), wt.float(), padding=1) + bias[None, :, None, None] + rowb[:, :, None, None] + res.float()
    ref = F.silu(ref).permute(0, 2, 3, 1).reshape(M, Co)
    assert rel_err(o32.cpu(), ref) < TOL32
    assert rel_err(out.cpu(), ref) < TOLBF


@pytest.mark.parametrize("tile,B,H,W_,Ci,Co,split,ext_k,up2", [
    (7, 1, 16, 16, 64, 128, 1, 32, False), (8, 2, 6, 10, 64, 320, 1, 64, False), (9, 2, 12, 20, 128, 64, 2, 32, False),
    (10, 1, 20, 16, 256, 200, 2, 64, False), (7, 4, 5, 7, 64, 64, 1, 32, False), (8, 2, 10, 24, 128, 160, 1, 32, True)])
def test_conv3x3_patch_staged_with_k_extension(dev, tile, B, H, W_, Ci, Co, split, ext_k, up2):
    """The c3lier LoRA branch on the patch-staged convolution: C = conv3x3(x, W) + a_ext w_ext^T (a_ext = the low-rank image T
    per OUTPUT pixel, w_ext = scale * up; lora.py:102-106) -- one extra step behind the tap loop, carried by the first K
    split; stride 1 and the upsampler form; 32 and 64 low-rank columns; ragged tiles."""
    torch.manual_seed(tile + ext_k)
    Hi, Wi = (H // 2, W_ // 2) if up2 else (H, W_)
    x = torch.randn(B, Ci, Hi, Wi).to(bf)
    wt = (torch.randn(Co, Ci, 3, 3) / (9 * Ci) ** 0.5).to(bf)
    M = B * H * W_
    T = torch.randn(M, ext_k).to(bf); up = (torch.randn(Co, ext_k) * 0.2).to(bf)
    res = torch.randn(M, Co).to(bf)
    xh = x.permute(0, 2, 3, 1).contiguous().to(dev)
    wh = wt.permute(0, 2, 3, 1).contiguous().reshape(Co, 9 * Ci).to(dev)
    o32 = torch.zeros(M, Co, device=dev)
    Td, upd, resd = T.to(dev), up.to(dev), res.to(dev)
    g = hip.gemm_args(xh, wh, None, m=M, n=Co, k=9 * Ci, lda=Ci, a_mode=hip.A_CONV3_UP2 if up2 else hip.A_CONV3_S1,
                      conv=(B, H, W_, Hi, Wi), out_f32=o32, residual=resd, a_ext=Td, w_ext=upd, ext_k=ext_k)
    assert "conv_patch_kernel" in hip.gemm_describe(g, tile, split, None, 0)
    ws = torch.zeros(split * M * Co, device=dev) if split > 1 else None
    hip.gemm(g, ops.default_stream(), tile=tile, split_k=split, ws=ws)
    _sync(dev)
    xin = F.interpolate(x.float(), scale_factor=2.0, mode="nearest") if up2 else x.float()
    ref = F.conv2d(xin, wt.float(), padding=1).permute(0, 2, 3, 1).reshape(M, Co) + T.float() @ up.float().t() + res.float()
    assert rel_err(o32.cpu(), ref) < TOL32


@pytest.mark.parametrize("tile,B,Hi,Wi,Ci,Co,split", [(7, 1, 8, 8, 64, 128, 1), (8, 2, 5, 12, 128, 160, 1), (9, 3, 4, 4, 192, 72, 3),
                                                      (10, 2, 12, 8, 64, 200, 1), (7, 2, 16, 16, 64, 64, 1)])
def test_conv3x3_patch_staged_on_upsampled_input(dev, tile, B, Hi, Wi, Ci, Co, split):
    """conv_patch.hip on LECO_A_CONV3_UP2 (Upsample2D: nearest-2x, then 3x3): the patch is staged at INPUT resolution;
    vs F.conv2d(F.interpolate(x, 2x)) -- tiles over several images, ragged rows / columns, TW = 8 and 16, split-K."""
    torch.manual_seed(tile * 10 + Hi)
    H, W_ = 2 * Hi, 2 * Wi
    x = torch.randn(B, Ci, Hi, Wi).to(bf)
    wt = (torch.randn(Co, Ci, 3, 3) / (9 * Ci) ** 0.5).to(bf)
    bias = torch.randn(Co)
    xh = x.permute(0, 2, 3, 1).contiguous().to(dev)
    wh = wt.permute(0, 2, 3, 1).contiguous().reshape(Co, 9 * Ci).to(dev)
    M = B * H * W_
    o32 = torch.zeros(M, Co, device=dev)
    bias_d = bias.to(dev)
    g = hip.gemm_args(xh, wh, None, m=M, n=Co, k=9 * Ci, lda=Ci, a_mode=hip.A_CONV3_UP2, conv=(B, H, W_, Hi, Wi), out_f32=o32,
                      bias=bias_d)
    assert "conv_patch_kernel" in hip.gemm_describe(g, tile, 1, None, 0) and ", true>" in hip.gemm_describe(g, tile, 1, None, 0)
    ws = torch.zeros(split * M * Co, device=dev) if split > 1 else None
    hip.gemm(g, ops.default_stream(), tile=tile, split_k=split, ws=ws)
    _sync(dev)
    ref = F.conv2d(F.interpolate(x.float(), scale_factor=2.0, mode="nearest"), wt.float(), padding=1) + bias[None, :, None, None]
    assert rel_err(o32.cpu(), ref.permute(0, 2, 3, 1).reshape(M, Co)) < TOL32


def test_conv3x3_patch_falls_back_when_not_applicable(dev):
    """Tile ids 7..10 on a problem the patch kernel does not cover (stride 2) run the implicit-GEMM kernel instead."""
    g = hip.gemm_args(torch.zeros(2 * 8 * 8, 64, dtype=bf, device=dev), torch.zeros(64, 576, dtype=bf, device=dev),
                      torch.zeros(2 * 4 * 4, 64, dtype=bf, device=dev), m=32, n=64, k=576, lda=64, a_mode=hip.A_CONV3_S2,
                      conv=(2, 4, 4, 8, 8))
    assert "gemm_kernel" in hip.gemm_describe(g, 7, 1, None, 0)


def _col_stats_ref(y_bf, B, HW, atom=1):
    """{sum, sumsq} per (sample, atom of adjacent columns) of the STORED bf16 values."""
    y = y_bf.float().cpu().reshape(B, HW, -1, atom)
    return torch.stack([y.sum((1, 3)), (y * y).sum((1, 3))], dim=-1)          # [B][C / atom][2]


@pytest.mark.parametrize("atom", [1, 6])
@pytest.mark.parametrize("kind", ["plain", "plain_split", "conv_old", "conv_patch", "conv_patch_split", "conv_patch_big"])
def test_producer_side_groupnorm_statistics(dev, kind, atom):
    """leco_gemm_args.col_stats: the GEMM / conv epilogues (gemm.hip, conv_patch.hip) and the split-K finish leave
    {sum, sumsq} per (sample, output column) of the bf16 values they store -- tiles that span samples, ragged rows /
    columns, residual + bias + SiLU in front of the rounding."""
    torch.manual_seed(5)
    B, H, W_ = (2, 16, 16) if kind == "conv_patch_big" else (3, 6, 10)      # big: single-sample tiles (the LDS fast path)
    HW, Ci, Co = H * W_, 128, 72
    M = B * HW
    bias = torch.randn(Co).to(dev)
    res = torch.randn(M, Co).to(bf).to(dev)
    out = torch.zeros(M, Co, dtype=bf, device=dev)
    cs = torch.zeros(B, Co // atom, 2, device=dev)
    ws = torch.zeros(4 * M * Co, device=dev)
    if kind.startswith("plain"):
        a = torch.randn(M, Ci).to(bf).to(dev); w = (torch.randn(Co, Ci) / Ci ** 0.5).to(bf).to(dev)
        g = hip.gemm_args(a, w, out, m=M, n=Co, k=Ci, bias=bias, residual=res, act=hip.ACT_SILU, col_stats=cs, stats_rows=HW,
                          stats_atom=atom)
        tile, split = (1, 2) if kind == "plain_split" else (3, 1)
    else:
        x = torch.randn(B, H, W_, Ci).to(bf).to(dev); w = (torch.randn(Co, 9 * Ci) / (9 * Ci) ** 0.5).to(bf).to(dev)
        g = hip.gemm_args(x, w, out, m=M, n=Co, k=9 * Ci, lda=Ci, a_mode=hip.A_CONV3_S1, conv=(B, H, W_, H, W_), bias=bias,
                          residual=res, act=hip.ACT_SILU, col_stats=cs, stats_rows=HW, stats_atom=atom)
        tile, split = {"conv_old": (-1, 1), "conv_patch": (9, 1), "conv_patch_split": (7, 2), "conv_patch_big": (8, 1)}[kind]
    hip.gemm(g, ops.default_stream(), tile=tile, split_k=split, ws=ws)
    _sync(dev)
    ref = _col_stats_ref(out, B, HW, atom)
    assert out.float().abs().sum() > 0
    assert rel_err(cs.cpu(), ref) < 1e-5


@pytest.mark.parametrize("act,B,HW,C0,C1,G,atom", [(1, 2, 70, 64, 0, 32, 2), (0, 3, 33, 64, 128, 32, 2), (1, 1, 300, 320, 0, 32, 10),
                                                   (1, 2, 16, 1280, 640, 32, 10), (0, 2, 20, 64, 0, 16, 1),
                                                   (1, 1, 11, 1280, 1280, 32, 10)])      # 320 channel vectors: two column sweeps
def test_groupnorm_from_channel_statistics(dev, act, B, HW, C0, C1, G, atom):
    """leco_colstats + leco_groupnorm_apply_stats == F.group_norm(+SiLU) on the (optionally two-source) input; the group
    sums it leaves in `stats` are what the backward kernel expects (same dx as with leco_groupnorm_fwd's statistics)."""
    torch.manual_seed(HW)
    C = C0 + C1
    x0 = (torch.randn(B * HW, C0) * 2 + 0.5).to(bf).to(dev)
    x1 = (torch.randn(B * HW, C1) - 0.3).to(bf).to(dev) if C1 else None
    gamma, beta = torch.randn(C).to(dev), torch.randn(C).to(dev)
    cs0 = torch.zeros(B, C0 // atom, 2, device=dev)
    cs1 = torch.zeros(B, C1 // atom, 2, device=dev) if C1 else None
    ops.Op("leco_colstats", (x0.data_ptr(), C0, cs0.data_ptr(), atom, B, HW, C0)).run()
    if C1:
        ops.Op("leco_colstats", (x1.data_ptr(), C1, cs1.data_ptr(), atom, B, HW, C1)).run()
    y = torch.zeros(B * HW, C, dtype=bf, device=dev)
    stats = torch.zeros(B * G * 2 * 257, device=dev)
    ops.Op("leco_groupnorm_apply_stats", (x0.data_ptr(), C0, x1.data_ptr() if C1 else None, C1, C0 if C1 else 0, cs0.data_ptr(),
                                          cs1.data_ptr() if C1 else None, atom, gamma.data_ptr(), beta.data_ptr(), B, HW, C, G, 1e-5,
                                          act, stats.data_ptr(), y.data_ptr(), C)).run()
    _sync(dev)
    xin = torch.cat([x0] + ([x1] if C1 else []), 1).float().cpu().reshape(B, HW, C).permute(0, 2, 1)
    ref = F.group_norm(xin, G, gamma.cpu(), beta.cpu(), 1e-5)
    ref = (F.silu(ref) if act else ref).permute(0, 2, 1).reshape(B * HW, C)
    assert rel_err(y.cpu(), ref) < TOLBF
    # the statistics match what the reducing forward kernel leaves for the backward
    y2 = torch.zeros_like(y); stats2 = torch.zeros_like(stats)
    ops.groupnorm_fwd(x0, C0, x1, C1, C0 if C1 else 0, gamma, beta, B, HW, C, G, 1e-5, act, stats2, y2, C).run()
    _sync(dev)
    assert rel_err(stats[:B * G * 2].cpu(), stats2[:B * G * 2].cpu()) < 1e-5 and rel_err(y.cpu(), y2.cpu()) < 2e-3


@pytest.mark.parametrize("act,B,HW,C0,C1,G", [
    (0, 2, 37, 64, 128, 32),      # cg = 6 -> 4 groups per block, two-source
    (1, 2, 37, 64, 128, 32),
    (1, 3, 300, 320, 0, 32),      # cg = 10 (SD1.5 level 0): 40-channel blocks, more pixels than pixel lanes
    (1, 1, 70, 640, 320, 32),     # cg = 30, concat split inside a block's channel run
    (0, 2, 16, 64, 0, 32),        # cg = 2 (tiny config): 4 groups inside one 8-channel vector
    (1, 2, 9, 1280, 1280, 32),    # cg = 80: one group per block
    (1, 2, 1100, 320, 0, 32),     # 6 pixels per thread: the register-resident forward (one read of the tensor), NVR = 6
    (0, 1, 1024, 640, 320, 32),   # 16 pixels per thread (cg = 30: 68 pixel lanes), NVR = 16, two-source
    (1, 1, 1150, 480, 480, 32),   # 17 pixels per thread: back to two passes
    (1, 1, 2600, 320, 0, 32),     # > 200 KB per (sample, group run) on few blocks: pixel-parallel three-launch path
    (1, 2, 300, 640, 0, 32),      # cg = 20, >= 256 pixels, < 128 blocks: 8-byte vectors, one group per block (round 6)
    (0, 2, 260, 320, 320, 32)])   # the same with the concat split between two of the 4-channel vectors' groups
def test_groupnorm_fwd_bwd(dev, act, B, HW, C0, C1, G):
    torch.manual_seed(3)
    C = C0 + C1
    x0 = torch.randn(B * HW, C0).to(bf).to(dev)
    x1 = (torch.randn(B * HW, C1) * 2 + 0.5).to(bf).to(dev) if C1 else None
    gamma = torch.randn(C).to(dev); beta = torch.randn(C).to(dev)
    stats = torch.zeros(B * G * 2 * 257, device=dev); y = torch.zeros(B * HW, C, dtype=bf, device=dev)
    ops.groupnorm_fwd(x0, C0, x1, C1, C0, gamma, beta, B, HW, C, G, 1e-5, act, stats, y, C).run()
    xcat = torch.cat([x0, x1], 1) if C1 else x0
    xc = xcat.float().cpu().reshape(B, HW, C).permute(0, 2, 1).requires_grad_(True)
    ref = F.group_norm(xc, G, gamma.cpu(), beta.cpu(), 1e-5)
    ref = F.silu(ref) if act else ref
    dy = torch.randn(B * HW, C).to(bf).to(dev); bstats = torch.zeros(B * G * 2 * 257, device=dev)
    dx = torch.zeros(B * HW, C, dtype=bf, device=dev)
    ops.groupnorm_bwd(x0, C0, x1, C1, C0, dy, C, gamma, beta, stats, B, HW, C, G, 1e-5, act, bstats, dx, C).run()
    _sync(dev)
    ref.backward(dy.float().cpu().reshape(B, HW, C).permute(0, 2, 1))
    assert rel_err(y.cpu().reshape(B, HW, C).permute(0, 2, 1), ref) < TOLBF
    assert rel_err(dx.cpu().reshape(B, HW, C).permute(0, 2, 1), xc.grad) < TOLBF
    # statistics are published for the backward: {sum x, sum x^2} per (sample, group)
    xs = xcat.float().cpu().reshape(B, HW, G, C // G)
    s_ref = torch.stack([xs.sum((1, 3)), (xs * xs).sum((1, 3))], -1).reshape(-1)
    assert rel_err(stats[:B * G * 2].cpu(), s_ref) < 1e-4


@pytest.mark.parametrize("M,C", [(37, 320), (9, 1280), (5, 64), (77, 640), (16390, 320), (4100, 640), (2050, 1280)])
def test_layernorm_fwd_bwd(dev, M, C):
    torch.manual_seed(4)
    x = (torch.randn(M, C) * 1.5 + 0.3).to(bf).to(dev); gamma = torch.randn(C).to(dev); beta = torch.randn(C).to(dev)
    y = torch.zeros(M, C, dtype=bf, device=dev); mean = torch.zeros(M, device=dev); rstd = torch.zeros(M, device=dev)
    ops.layernorm_fwd(x, C, gamma, beta, 1e-5, M, C, y, C, mean, rstd).run()
    xx = x.float().cpu().requires_grad_(True)
    ref = F.layer_norm(xx, (C,), gamma.cpu(), beta.cpu(), 1e-5)
    dy = torch.randn(M, C).to(bf).to(dev); dres = torch.randn(M, C).to(bf).to(dev)
    dx = torch.zeros(M, C, dtype=bf, device=dev)
    ops.layernorm_bwd(x, C, dy, C, gamma, mean, rstd, dres, C, M, C, dx, C).run()
    _sync(dev)
    ref.backward(dy.float().cpu())
    assert rel_err(y.cpu(), ref) < TOLBF
    assert rel_err(dx.cpu(), xx.grad + dres.float().cpu()) < TOLBF


ATTN_CASES = [(2, 2, 70, 77, 40), (1, 3, 130, 130, 80), (1, 2, 64, 200, 160), (2, 2, 50, 64, 64), (1, 2, 96, 96, 32),
              (1, 1, 1100, 1100, 40)]


@pytest.mark.parametrize("B,H,Sq,Skv,D", ATTN_CASES)
def test_attention_fwd_bwd(dev, B, H, Sq, Skv, D):
    torch.manual_seed(5)
    C = H * D
    q = torch.randn(B, Sq, C).to(bf).to(dev); k = torch.randn(B, Skv, C).to(bf).to(dev)
    v = torch.randn(B, Skv, C).to(bf).to(dev)
    o = torch.zeros(B, Sq, C, dtype=bf, device=dev); lse = torch.zeros(B, H, Sq, device=dev)
    sc = D ** -0.5
    ops.attention_fwd(q.data_ptr(), C, Sq * C, k.data_ptr(), C, Skv * C, v.data_ptr(), C, Skv * C, o.data_ptr(), C,
                      Sq * C, lse, B, H, Sq, Skv, D, sc).run()
    do = torch.randn(B, Sq, C).to(bf).to(dev)
    dq = torch.zeros_like(q); dk = torch.zeros_like(k); dv = torch.zeros_like(v)
    delta = torch.zeros(B, H, Sq, device=dev)
    ops.attention_bwd(q.data_ptr(), C, Sq * C, k.data_ptr(), C, Skv * C, v.data_ptr(), C, Skv * C, o.data_ptr(), C,
                      Sq * C, do.data_ptr(), C, Sq * C, lse, delta, dq.data_ptr(), C, Sq * C, dk.data_ptr(), C,
                      Skv * C, dv.data_ptr(), C, Skv * C, B, H, Sq, Skv, D, sc).run()
    _sync(dev)
    qq, kk, vv = [t.float().cpu().requires_grad_(True) for t in (q, k, v)]
    qh = qq.reshape(B, Sq, H, D).transpose(1, 2); kh = kk.reshape(B, Skv, H, D).transpose(1, 2)
    vh = vv.reshape(B, Skv, H, D).transpose(1, 2)
    s = (qh @ kh.transpose(-1, -2)) * sc
    ref = (torch.softmax(s, -1) @ vh).transpose(1, 2).reshape(B, Sq, C)
    ref.backward(do.float().cpu())
    assert rel_err(o.cpu(), ref) < TOLBF
    # d = 40: the row sum comes out of the PV MFMA (a 1.0 column in the padded V tile), i.e. it is the sum of the
    # bf16-ROUNDED probabilities -- the same values the output was accumulated from
    assert rel_err(lse.cpu(), torch.logsumexp(s, -1)) < (2e-4 if D == 40 else 1e-5)
    assert rel_err(dq.cpu(), qq.grad) < 4e-3
    assert rel_err(dk.cpu(), kk.grad) < 4e-3
    assert rel_err(dv.cpu(), vv.grad) < 4e-3


@pytest.mark.parametrize("B,H,S,fused,gain,D", [(1, 2, 256, False, 1.0, 40), (2, 1, 384, True, 1.0, 40), (1, 1, 128, True, 1.0, 40),
                                                (1, 1, 512, False, 3.0, 40),     # scores x 9: the deferred maximum has to move mid-way
                                                (1, 1, 192, True, 1.0, 40),      # 64-query workgroups (one query fragment per wave)
                                                (1, 2, 256, True, 1.0, 64), (1, 1, 192, False, 2.0, 64),     # 2-deep ring
                                                (1, 1, 256, False, 1.0, 80), (2, 1, 192, True, 1.0, 80)])    # three k-steps, 5 column fragments
def test_attention_dma_staged_forward(dev, monkeypatch, B, H, S, fused, gain, D):
    """attn_fwd_dma_kernel (d = 40 / 64 / 80 self-attention with the K / V tiles brought in by LDS-DMA into a 2- or 3-deep ring;
    the planner's large self-attention shapes take it by default, LECO_ATTN_DMA=2 forces it for small grids) against fp32 torch and against
    the register-staged kernel (LECO_ATTN_DMA=0); 2 / 4 / 6 key tiles: prologue only, one ring wrap, two ring wraps; separate
    and q|k|v-fused layouts.  tests/test_host.py re-runs this under the emulator's late-DMA model."""
    torch.manual_seed(S + H)
    C = H * D
    if fused:
        qkv = (torch.randn(B, S, 3 * C) * gain).to(bf).to(dev)
        ptrs = (qkv.data_ptr(), qkv.data_ptr() + 2 * C, qkv.data_ptr() + 4 * C)
        ld, bs = 3 * C, S * 3 * C
        q, k, v = [t.float().cpu() for t in qkv.chunk(3, -1)]
    else:
        qs = [(torch.randn(B, S, C) * gain).to(bf).to(dev) for _ in range(3)]
        ptrs, ld, bs = tuple(t.data_ptr() for t in qs), C, S * C
        q, k, v = [t.float().cpu() for t in qs]
    outs = {}
    for mode in ("2", "0"):
        monkeypatch.setenv("LECO_ATTN_DMA", mode)
        o = torch.zeros(B, S, C, dtype=bf, device=dev); lse = torch.zeros(B, H, S, device=dev)
        ops.attention_fwd(ptrs[0], ld, bs, ptrs[1], ld, bs, ptrs[2], ld, bs, o.data_ptr(), C, S * C, lse, B, H, S, S, D, D ** -0.5).run()
        _sync(dev)
        outs[mode] = (o.cpu(), lse.cpu())
    qh, kh, vh = [t.reshape(B, S, H, D).transpose(1, 2) for t in (q, k, v)]
    sc = qh @ kh.transpose(-1, -2) * D ** -0.5
    ref = (torch.softmax(sc, -1) @ vh).transpose(1, 2).reshape(B, S, C)
    assert rel_err(outs["2"][0], ref) < TOLBF and rel_err(outs["2"][1], torch.logsumexp(sc, -1)) < 2e-4
    # (different reference maxima -> different bf16 roundings of P: the two kernels agree to the rounding, not bit for bit)
    assert rel_err(outs["2"][0], outs["0"][0]) < 4e-3 and rel_err(outs["2"][1], outs["0"][1]) < 1e-4


def test_attention_strided_fused_qkv(dev):
    """q|k|v packed in one [B][S][3C] buffer (how the UNet engine lays them out)."""
    torch.manual_seed(6)
    B, H, S, D = 2, 2, 48, 40
    C = H * D
    qkv = torch.randn(B, S, 3 * C).to(bf).to(dev)
    o = torch.zeros(B, S, C, dtype=bf, device=dev); lse = torch.zeros(B, H, S, device=dev)
    p0 = qkv.data_ptr()
    ops.attention_fwd(p0, 3 * C, S * 3 * C, p0 + 2 * C, 3 * C, S * 3 * C, p0 + 4 * C, 3 * C, S * 3 * C, o.data_ptr(), C,
                      S * C, lse, B, H, S, S, D, D ** -0.5).run()
    _sync(dev)
    q, k, v = [t.float().cpu().reshape(B, S, H, D).transpose(1, 2) for t in qkv.chunk(3, -1)]
    ref = (torch.softmax(q @ k.transpose(-1, -2) * D ** -0.5, -1) @ v).transpose(1, 2).reshape(B, S, C)
    assert rel_err(o.cpu(), ref) < TOLBF


def test_elementwise_family(dev):
    torch.manual_seed(7)
    M, Fd = 33, 128
    u = torch.randn(M, 2 * Fd).to(bf).to(dev); y = torch.zeros(M, Fd, dtype=bf, device=dev)
    ops.geglu_fwd(u, 2 * Fd, y, Fd, M, Fd).run()
    uu = u.float().cpu().requires_grad_(True)
    a, g = uu.chunk(2, -1)
    ref = a * F.gelu(g)
    dy = torch.randn(M, Fd).to(bf).to(dev); du = torch.zeros(M, 2 * Fd, dtype=bf, device=dev)
    ops.geglu_bwd(u, 2 * Fd, dy, Fd, du, 2 * Fd, M, Fd).run()
    ref.backward(dy.float().cpu())
    _sync(dev)
    assert rel_err(y.cpu(), ref) < TOLBF and rel_err(du.cpu(), uu.grad) < TOLBF
    # strided add
    a = torch.randn(10, 64).to(bf).to(dev); b = torch.randn(10, 128).to(bf).to(dev); c = torch.randn(10, 64).to(bf).to(dev)
    o = torch.zeros(10, 64, dtype=bf, device=dev)
    ops.add(a.data_ptr(), 64, b.data_ptr() + 64 * 2, 128, c.data_ptr(), 64, o.data_ptr(), 64, 10, 64).run()
    _sync(dev)
    assert rel_err(o, a.float() + b[:, 64:].float() + c.float()) < TOLBF
    # upsample dgrad
    B, H, W, C = 2, 3, 5, 64
    dyh = torch.randn(B, 2 * H, 2 * W, C).to(bf).to(dev); dx = torch.zeros(B, H, W, C, dtype=bf, device=dev)
    ops.upsample2x_bwd(dyh, dx, B, H, W, C).run()
    xx = torch.zeros(B, C, H, W, requires_grad=True)
    F.interpolate(xx, scale_factor=2.0, mode="nearest").backward(dyh.float().cpu().permute(0, 3, 1, 2))
    _sync(dev)
    assert rel_err(dx.cpu().permute(0, 3, 1, 2), xx.grad) < TOLBF


@pytest.mark.parametrize("B,H,W,Co,C", [(2, 6, 7, 64, 128),     # one output fragment per wave; ragged last pixel group
                                        (1, 16, 20, 320, 320),   # SD level-0 widths: 5 fragments per wave, 90 k-steps over 4 waves
                                        (3, 5, 5, 48, 96),       # 3 fragments: an idle wave; 27 k-steps
                                        (2, 4, 4, 24, 72)])      # widths the MFMA forms do not take: the scalar kernels
def test_conv_in_out(dev, B, H, W, Co, C):
    """leco_conv_in (NCHW bf16 -> channels-last, fp32 weights) and leco_conv_out (channels-last -> NCHW fp32) against
    F.conv2d; both run on the matrix cores where the widths allow (hi + lo split of the fp32 conv_in weights)."""
    torch.manual_seed(8)
    Ci = 4
    x = torch.randn(B, Ci, H, W).to(bf).to(dev); w = (torch.randn(Co, Ci, 3, 3) * 0.2).to(dev); bias = torch.randn(Co).to(dev)
    y = torch.zeros(B, H, W, Co, dtype=bf, device=dev)
    wt_in = w.permute(1, 2, 3, 0).contiguous()
    ops.conv_in(x, wt_in, bias, y, B, H, W, Ci, Co).run()
    _sync(dev)
    assert rel_err(y.cpu().permute(0, 3, 1, 2), F.conv2d(x.float().cpu(), w.cpu(), bias.cpu(), padding=1)) < TOLBF
    xh = torch.randn(B, H, W, C).to(bf).to(dev); w4 = (torch.randn(4, C, 3, 3) * 0.05).to(bf); b4 = torch.randn(4).to(dev)
    wl = w4.permute(0, 2, 3, 1).contiguous().to(dev); yo = torch.zeros(B, 4, H, W, device=dev)
    ops.conv_out(xh, wl, b4, yo, B, H, W, C, 4).run()
    xx = xh.float().cpu().permute(0, 3, 1, 2).requires_grad_(True)
    ref = F.conv2d(xx, w4.float(), b4.cpu(), padding=1)
    dyo = torch.randn(B, 4, H, W).to(dev); dxh = torch.zeros(B, H, W, C, dtype=bf, device=dev)
    ops.conv_out_bwd(dyo, wl, dxh, B, H, W, C, 4).run()
    ref.backward(dyo.cpu())
    _sync(dev)
    assert rel_err(yo.cpu(), ref) < TOL32
    assert rel_err(dxh.cpu().permute(0, 3, 1, 2), xx.grad) < TOLBF


@pytest.mark.parametrize("m,n,k,rank,res", [(200, 256, 320, 4, True), (96, 128, 640, 0, True), (160, 384, 640, 8, False),
                                            (72, 256, 1280, 4, True), (64, 128, 1280, 0, False)])
def test_xgemm_a_stationary_linear(dev, m, n, k, rank, res):
    """leco_xgemm (csrc/xgemm.hip): c = a W^T (+ (a down^T)(scale up)^T) + bias (+ residual) with W in MFMA fragment order,
    against fp32 torch on the same bf16 operands; row counts that are not a multiple of the tile, both stack heights of the
    fused down-projection (rank 4 x 3 groups -> 16 rows; rank 8 x 3 -> 32 rows), every supported K."""
    torch.manual_seed(m + n + k)
    a = (torch.randn(m, k) * 0.5).to(bf)
    w = (torch.randn(n, k) / math.sqrt(k)).to(bf)
    bias = torch.randn(n) * 0.1
    r = torch.randn(m, n).to(bf) if res else None
    ref = a.float() @ w.float().t() + bias
    dn = up = None
    t_rows = 0
    if rank:
        R = 3 * rank                                   # q|k|v-style stacking: 3 groups
        t_rows = 16 if R <= 16 else 32
        dn = torch.zeros(t_rows, k)
        dn[:R] = torch.randn(R, k) / math.sqrt(k)
        up = torch.zeros(n, 32)
        up[:, :R] = torch.randn(n, R) * 0.2
        dn, up = dn.to(bf), up.to(bf)
        T = (a.float() @ dn.float().t()).to(bf).float()          # the kernel rounds T to bf16 before the K-extension
        ref = ref + T @ up.float()[:, :t_rows].t()
    if res:
        ref = ref + r.float()
    ad, wd = a.to(dev), hip.pack_fragments(w).to(dev)
    c = torch.zeros(m, n, dtype=bf, device=dev)
    # (`leco_xlin` holds raw addresses: every operand tensor must stay alive until the launch has run)
    bd, dnd, upd = bias.to(dev), None if dn is None else dn.to(dev), None if up is None else up.to(dev)
    lin = hip.xlin(wd, bd, dnd, upd, t_rows, packed=True)
    rd = r.to(dev) if res else None
    keep = (ad, wd, c, bd, dnd, upd, rd)
    assert ops.xgemm_supported(m, n, k) and not ops.xgemm_supported(m, n + 64, k) and not ops.xgemm_supported(m, n, 768)
    ops.xgemm(ad.data_ptr(), k, lin, c.data_ptr(), n, m, n, k, residual=None if rd is None else rd.data_ptr(), ldr=n, keep=keep).run()
    _sync(dev)
    assert rel_err(c.float().cpu(), ref) < TOLBF
    # rows beyond m are never written: a guard row behind the output stays untouched
    c2 = torch.full((m + 8, n), 7.0, dtype=bf, device=dev)
    ops.xgemm(ad.data_ptr(), k, lin, c2.data_ptr(), n, m, n, k, keep=keep).run()
    _sync(dev)
    assert (c2[m:].float() == 7.0).all() and torch.isfinite(c2.float()).all()


@pytest.mark.parametrize("adt", [bf, torch.float32])
def test_step_glue_launches(dev, adt):
    """leco_step_begin / leco_step_mid (the tensor moves between the launch plans of a step, train_lora.py:175-199) against
    the framework ops they replace, bit for bit."""
    class P:        # the two fields of a launch plan the op touches
        def __init__(self):
            self.t_table = torch.zeros(1024, device=dev)
            self.t_idx = torch.full((1,), 7, dtype=torch.int32, device=dev)
    torch.manual_seed(4)
    bs, n = 2, 4 * 8 * 24
    x = torch.randn(bs * n, device=dev)
    x2 = torch.zeros(2 * bs * n, dtype=adt, device=dev)
    t_idx = torch.full((1,), 5, dtype=torch.int32, device=dev)
    ops.step_begin(x, x2, 0.37, bs * n, t_idx).run()
    _sync(dev)
    want = (x * 0.37).to(adt)
    assert torch.equal(x2.cpu(), torch.cat([want, want]).cpu()) and t_idx.item() == 0
    a, b = torch.zeros_like(x2), torch.zeros(3 * x2.numel(), dtype=adt, device=dev)
    pa, pb = P(), P()
    ops.step_mid(x2, a, b, 3, 439.0, pa, pb, 512).run()
    _sync(dev)
    assert torch.equal(a.cpu(), x2.cpu()) and torch.equal(b.cpu(), x2.repeat(3).cpu())
    for q in (pa, pb):
        assert q.t_idx.item() == 512 and q.t_table[512].item() == 439.0 and q.t_table.sum().item() == 439.0
    a.zero_()
    ops.step_mid(x2, None, b, 1, 3.0, None, pb, 9).run()       # generic schedulers: only the batched frozen pass is filled
    _sync(dev)
    assert a.abs().sum().item() == 0 and pb.t_idx.item() == 9 and pb.t_table[9].item() == 3.0 and pa.t_idx.item() == 512


def test_timestep_ddim_loss_adamw(dev):
    from oracle.unet_ref import timestep_sinusoid
    torch.manual_seed(9)
    tt = torch.tensor([999., 980., 0., 1., 500.], device=dev); idx = torch.tensor([1], dtype=torch.int32, device=dev)
    out = torch.zeros(2, 320, dtype=bf, device=dev)
    ops.timestep_embedding(tt, idx, 0, 2, 320, out).run()
    ops.advance(idx).run()
    _sync(dev)
    assert rel_err(out.cpu(), timestep_sinusoid(torch.tensor([980., 980.]), 320)) < TOLBF
    assert idx.item() == 2
    bs, n = 2, 4 * 8 * 8
    pred = torch.randn(2 * bs * n, device=dev); x = torch.randn(bs * n, device=dev)
    x2 = torch.zeros(2 * bs * n, dtype=bf, device=dev)
    coef = torch.tensor([0., 0., 1.01, -0.03, 5, 5], device=dev); st = torch.tensor([1], dtype=torch.int32, device=dev)
    x_ref = 1.01 * x - 0.03 * (pred[:bs * n] + 3 * (pred[bs * n:] - pred[:bs * n]))
    ops.cfg_ddim_step(pred, x, x2, coef, st, 3.0, bs * n).run()
    _sync(dev)
    assert rel_err(x, x_ref) < 1e-6 and rel_err(x2, torch.cat([x_ref, x_ref])) < TOLBF
    t, p, nn_, u = [torch.randn(2 * bs * n, device=dev) for _ in range(4)]
    loss = torch.zeros(1, device=dev); dp = torch.zeros(2 * bs * n, device=dev)
    ops.esd_loss(t, p, nn_, u, 1.0, 1.5, -1.0, bs * n, loss, dp).run()
    _sync(dev)
    tt_ = t.cpu().clone().requires_grad_(True)

    def gd(z):
        return z[:bs * n] + 1.0 * (z[bs * n:] - z[:bs * n])
    ref = F.mse_loss(gd(tt_), gd(nn_.cpu()) - 1.5 * (gd(p.cpu()) - gd(u.cpu())))
    ref.backward()
    assert abs(loss.item() - ref.item()) < 1e-5 * abs(ref.item())
    assert rel_err(dp.cpu(), tt_.grad) < 1e-5
    # a large batch runs on many blocks; the last-arriving block adds the partials in block order: repeated launches give
    # the same bits (and re-arm the ticket)
    big = 300000
    tb, pb_, nb, ub = [torch.randn(2 * big, device=dev) for _ in range(4)]
    lb = torch.zeros(3, device=dev)
    for i in range(3):
        ops.esd_loss(tb, pb_, nb, ub, 0.5, 2.0, 1.0, big, lb[i:i + 1], None).run()
    _sync(dev)

    def gd5(z):
        return (z[:big] + 0.5 * (z[big:] - z[:big])).double().cpu()
    refb = ((gd5(tb) - (gd5(nb) + 2.0 * (gd5(pb_) - gd5(ub)))) ** 2).mean().item()
    assert abs(lb[0].item() - refb) < 1e-5 * refb and lb[0].item() == lb[1].item() == lb[2].item()
    N = 1000
    p0 = torch.randn(N); g0 = torch.randn(N)
    pp = p0.clone().to(dev); m0 = torch.zeros(N, device=dev); v0 = torch.zeros(N, device=dev)
    sh = torch.zeros(N, dtype=bf, device=dev)
    pt = torch.nn.Parameter(p0.clone()); opt = torch.optim.AdamW([pt], lr=1e-2)
    for step in range(1, 4):
        pt.grad = g0 * step
        opt.step()
        hyper = torch.tensor([1e-2, 1 - 0.9 ** step, 1 - 0.999 ** step, 1.0], device=dev)
        gg = (g0 * step).to(dev)
        ops.adamw(pp, gg, m0, v0, sh, hyper, 0.9, 0.999, 1e-8, 0.01, N).run()
        _sync(dev)
    assert rel_err(pp.cpu(), pt.data) < 1e-5 and rel_err(sh.cpu(), pt.data) < TOLBF


def test_lora_pack_forward_wgrad(dev):
    torch.manual_seed(10)
    r, K, N_, groups, M = 4, 64, 192, 3, 70
    downs = [torch.randn(r, K).to(bf).to(dev) for _ in range(groups)]
    ups = [torch.randn(N_ // groups, r).to(bf).to(dev) for _ in range(groups)]
    R, R16, Rp = groups * r, 16, 32
    dn_s = torch.zeros(Rp, K, dtype=bf, device=dev); up_p = torch.zeros(N_, Rp, dtype=bf, device=dev)
    up_t = torch.zeros(R16, N_, dtype=bf, device=dev); dn_p = torch.zeros(K, Rp, dtype=bf, device=dev)
    site = hip.LoraSite()
    for g in range(groups):
        site.down[g] = downs[g].data_ptr(); site.up[g] = ups[g].data_ptr()
    site.groups, site.r, site.k, site.n, site.scale = groups, r, K, N_, 0.25
    site.dn_s, site.up_p, site.up_t, site.dn_p = dn_s.data_ptr(), up_p.data_ptr(), up_t.data_ptr(), dn_p.data_ptr()
    buf = torch.frombuffer(bytearray(bytes(site)), dtype=torch.uint8).clone().to(dev)
    ops.lora_pack(buf, 1).run()
    _sync(dev)
    dn_ref = torch.cat(downs, 0).cpu()
    upbd = torch.block_diag(*[u_.float().cpu() for u_ in ups])
    assert torch.equal(dn_s[:R].cpu(), dn_ref) and dn_s[R:].abs().max().item() == 0
    assert rel_err(up_p[:, :R].cpu(), 0.25 * upbd) < TOLBF and up_p[:, R:].abs().max().item() == 0
    assert torch.equal(up_t[:R].cpu().float(), upbd.T)
    assert rel_err(dn_p[:, :R].cpu(), 0.25 * dn_ref.float().T) < TOLBF
    # y = x W^T + s (x A^T) B^T via T = x dn_s^T then the K-extension tile
    x = torch.randn(M, K).to(bf).to(dev); Wm = (torch.randn(N_, K) / 8).to(bf).to(dev)
    T = torch.zeros(M, Rp, dtype=bf, device=dev)
    ops.gemm(hip.gemm_args(x, dn_s, T, m=M, n=Rp, k=K)).run()
    o32 = torch.zeros(M, N_, device=dev)
    ops.gemm(hip.gemm_args(x, Wm, None, m=M, n=N_, k=K, a_ext=T, w_ext=up_p, ext_k=Rp, out_f32=o32)).run()
    _sync(dev)
    ref = x.float().cpu() @ Wm.float().cpu().T + T.float().cpu()[:, :R] @ (0.25 * upbd).to(bf).float().T
    assert rel_err(o32.cpu(), ref) < TOL32
    ref_T = x.float().cpu() @ dn_ref.float().T
    assert rel_err(T[:, :R].cpu(), ref_T) < TOLBF
    dy = torch.randn(M, N_).to(bf).to(dev); G = torch.zeros(N_ // groups, r, device=dev)
    ops.lora_wgrad(T.data_ptr() + 2 * r, Rp, dy.data_ptr() + 2 * 64, N_, G.data_ptr(), 1, r, M, r, 64, 0.25).run()
    _sync(dev)
    assert rel_err(G.cpu(), 0.25 * dy[:, 64:128].float().cpu().T @ T[:, r:2 * r].float().cpu()) < 1e-5


def test_lora_wgrad_grouped_launch_equals_single_launches(dev):
    """leco_lora_wgrad_grouped: problems of different M / cols / rank / gather mode in one launch == one launch each."""
    torch.manual_seed(18)
    probs, singles, keep = [], [], []
    for (M, r, cols, conv) in [(300, 4, 320, None), (70, 4, 64, None), (128, 8, 600, None), (2 * 6 * 6, 4, 64, (2, 6, 6, 6, 6))]:
        P = torch.randn(M, 32).to(bf).to(dev)
        Q = torch.randn(M if conv is None else conv[0] * conv[3] * conv[4], cols).to(bf).to(dev)
        Ga = torch.zeros(r, cols, device=dev); Gb = torch.zeros(r, cols, device=dev)
        keep += [P, Q, Ga, Gb]
        pr = dict(p=P.data_ptr(), ldp=32, q=Q.data_ptr(), ldq=cols, g_sj=cols, g_sc=1, m=M, r=r, cols=cols, scale=0.5)
        if conv is not None:
            pr.update(a_mode=hip.A_CONV3_S1, h_out=conv[1], w_out=conv[2], h_in=conv[3], w_in=conv[4], kh=0, kw=2)
        probs.append(dict(pr, g=Ga.data_ptr()))
        if conv is None:
            singles.append(ops.lora_wgrad(P.data_ptr(), 32, Q.data_ptr(), cols, Gb.data_ptr(), cols, 1, M, r, cols, 0.5))
        else:
            singles.append(ops.Op("leco_lora_wgrad_conv", (P.data_ptr(), 32, Q.data_ptr(), cols, Gb.data_ptr(), cols, 1, M, r,
                                                           cols, 0.5, hip.A_CONV3_S1, conv[1], conv[2], conv[3], conv[4], 0, 2,
                                                           None, 0)))
    ops.lora_wgrad_grouped(probs, dev).run()
    for op in singles:
        op.run()
    _sync(dev)
    for i in range(4):
        assert rel_err(keep[4 * i + 2].cpu(), keep[4 * i + 3].cpu()) < 1e-6 and keep[4 * i + 3].abs().sum().item() > 0


def test_lora_wgrad_deterministic_mode(dev):
    """part != NULL: per-slab partials + an in-order reduce instead of fp32 atomics -- bitwise reproducible, and it
    ACCUMULATES into G like the atomic form."""
    torch.manual_seed(16)
    M, r, cols = 700, 4, 320           # 6 slabs of 128 rows
    P = torch.randn(M, 32).to(bf).to(dev); Q = torch.randn(M, cols).to(bf).to(dev)
    ref = 0.5 * P[:, :r].float().cpu().T @ Q.float().cpu()
    part = torch.empty(6 * r * cols + 8, device=dev)
    outs = []
    for _ in range(3):
        G = torch.full((r, cols), 1.0, device=dev)
        ops.lora_wgrad(P.data_ptr(), 32, Q.data_ptr(), cols, G.data_ptr(), cols, 1, M, r, cols, 0.5, part).run()
        _sync(dev)
        outs.append(G.cpu().clone())
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    assert rel_err(outs[0] - 1.0, ref) < 1e-5
    with pytest.raises(hip.LecoError):     # scratch too small
        ops.lora_wgrad(P.data_ptr(), 32, Q.data_ptr(), cols, G.data_ptr(), cols, 1, M, r, cols, 0.5, part[:100]).run()


@pytest.mark.parametrize("name", ["euler_a", "lms", "ddpm"])
def test_cfg_sched_step_matches_scheduler_objects(dev, name):
    """leco_cfg_sched_step driven by `rows()` reproduces predict_noise's CFG combine (train_util.py:163-166) followed by
    scheduler.step (train_util.py:190) and the next scale_model_input (train_util.py:153) for the non-DDIM schedulers."""
    from leco_amd import scheduler as S
    torch.manual_seed(21)
    n, bs, numel, g = 6, 2, 4 * 8 * 8, 3.0
    half = bs * numel
    sch = S.create_noise_scheduler(name)
    sch.set_timesteps(n)
    ref = S.create_noise_scheduler(name)
    ref.set_timesteps(n)
    coef = sch.rows().to(dev)
    x0 = torch.randn(half) * float(sch.init_noise_sigma)
    x = x0.clone().to(dev); xr = x0.clone()
    x2 = torch.zeros(2 * half, dtype=bf, device=dev)
    step = torch.zeros(1, dtype=torch.int32, device=dev)
    hist = torch.zeros(3 * half, device=dev)
    for i, t in enumerate(ref.timesteps):
        pred = torch.randn(2 * half)
        noise = torch.randn(half)
        ops.cfg_sched_step(pred.to(dev), x, x2, coef, step, g, half, noise.to(dev) if sch.needs_noise else None,
                           hist if sch.n_hist else None, sch.n_hist).run()
        ops.advance(step).run()
        out = pred[:half] + g * (pred[half:] - pred[:half])
        xr = (ref.step(out, t, xr).prev_sample if name == "lms" else ref.step(out, t, xr, noise=noise).prev_sample)
        _sync(dev)
        assert rel_err(x.cpu(), xr) < 1e-5, (name, i)
        nxt = ref.scale_model_input(xr, ref.timesteps[i + 1]) if i + 1 < n else xr
        assert rel_err(x2.cpu()[:half].float(), nxt) < TOLBF and torch.equal(x2[:half], x2[half:])


def test_lion_matches_the_published_update(dev):
    """leco_lion vs the Lion update rule (Chen et al. 2023, as in lion_pytorch): decay, sign(interp), momentum."""
    torch.manual_seed(22)
    n, lr, b1, b2, wd, gs = 1000, 3e-4, 0.9, 0.99, 0.1, 0.5
    p0 = torch.randn(n); g0 = torch.randn(n); m0 = torch.randn(n) * 0.1
    g0[:10] = 0.0; m0[:10] = 0.0          # sign(0) = 0
    p, g, m = p0.clone().to(dev), g0.clone().to(dev), m0.clone().to(dev)
    shadow = torch.zeros(n, dtype=bf, device=dev)
    hyper = torch.tensor([lr, 1.0, 1.0, gs], device=dev)
    ops.lion(p, g, m, shadow, hyper, b1, b2, wd, n).run()
    _sync(dev)
    gr = g0 * gs
    pr = p0 * (1 - lr * wd) - lr * torch.sign(b1 * m0 + (1 - b1) * gr)
    mr = b2 * m0 + (1 - b2) * gr
    assert torch.allclose(p.cpu(), pr, atol=1e-7) and torch.allclose(m.cpu(), mr, atol=1e-7)
    assert torch.equal(shadow.cpu(), pr.to(bf))
