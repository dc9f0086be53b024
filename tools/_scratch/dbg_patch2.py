import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
from leco_amd import hip, ops
bf = torch.bfloat16
dev = torch.device("cuda:0")
def case(tile, B, H, W_, Ci, Co, ks, split, what):
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
    if "b" in what: kw["bias"] = bias.to(dev)
    if "r" in what: kw.update(rowbias=rowb.to(dev), rows_per_group=H * W_)
    if "s" in what: kw["residual"] = resh
    if "a" in what: kw["act"] = hip.ACT_SILU
    g = hip.gemm_args(xh, wh, out if "o" in what else None, m=M, n=Co, k=9 * Ci, a_mode=hip.A_CONV3_S1, conv=(B, H, W_, H, W_), out_f32=o32, lda=Ci, **kw)
    ws = torch.zeros(split * M * Co, device=dev) if split > 1 else None
    hip.gemm(g, ops.default_stream(), tile=tile, split_k=split, ws=ws)
    torch.cuda.synchronize()
    ref = F.conv2d(x.float(), wt.float(), padding=1)
    if "b" in what: ref = ref + bias[None, :, None, None]
    if "r" in what: ref = ref + rowb[:, :, None, None]
    if "s" in what: ref = ref + res.float()
    if "a" in what: ref = F.silu(ref)
    ref = ref.permute(0, 2, 3, 1).reshape(M, Co)
    d = (o32.cpu() - ref)
    bad = (d.abs() > 1e-3) | torch.isnan(d)
    rows = bad.any(1).nonzero().flatten().tolist()
    print(tile, split, what, "rel", (d.norm() / ref.norm()).item(), "bad rows", len(rows), rows[:16], "maxabs", d.abs().max().item())
for what in ["", "b", "r", "s", "a", "o", "brsao"]:
    case(9, 2, 12, 20, 128, 64, 0, 2, what)
case(9, 2, 12, 20, 128, 64, 0, 1, "brsao")
case(7, 2, 12, 20, 128, 64, 0, 2, "brsao")
case(1, 2, 12, 20, 128, 64, 0, 2, "brsao")
