import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
from leco_amd import hip, ops
bf = torch.bfloat16
dev = torch.device("cuda:0")
def case(tile, B, H, W_, Ci, Co, ks, split, epi=True):
    torch.manual_seed(tile * 100 + H)
    x = torch.randn(B, Ci, H, W_).to(bf)
    wt = (torch.randn(Co, Ci, 3, 3) / (9 * Ci) ** 0.5).to(bf)
    xh = x.permute(0, 2, 3, 1).contiguous().to(dev)
    wh = wt.permute(0, 2, 3, 1).contiguous().reshape(Co, 9 * Ci).to(dev)
    M = B * H * W_
    o32 = torch.zeros(M, Co, device=dev)
    g = hip.gemm_args(xh, wh, None, m=M, n=Co, k=9 * Ci, a_mode=hip.A_CONV3_S1, conv=(B, H, W_, H, W_), out_f32=o32, lda=Ci)
    ws = torch.zeros(max(1, split) * M * Co, device=dev)
    for rep in range(3):
        o32.zero_(); ws.fill_(float("nan"))
        hip.gemm(g, ops.default_stream(), tile=tile, split_k=split, ws=ws)
        torch.cuda.synchronize()
        ref = F.conv2d(x.float(), wt.float(), padding=1).permute(0, 2, 3, 1).reshape(M, Co)
        d = (o32.cpu() - ref)
        bad = (d.abs() > 1e-3) | torch.isnan(d)
        rows = bad.any(1).nonzero().flatten().tolist()
        cols = bad.any(0).nonzero().flatten().tolist()
        print(tile, (B, H, W_, Ci, Co, ks, split), "rep", rep, "rel", (d.norm() / ref.norm()).item(), "bad rows", len(rows), rows[:24], "bad cols", len(cols), cols[:8],
              "ws nan", torch.isnan(ws[:max(1, split) * M * Co]).sum().item() if split > 1 else "-")
for c in [(9, 2, 12, 20, 128, 64, 0, 2), (9, 2, 12, 20, 128, 64, 0, 1), (9, 2, 12, 20, 128, 128, 0, 2), (9, 2, 16, 16, 128, 64, 0, 2), (7, 2, 12, 20, 128, 64, 0, 2),
          (10, 1, 20, 16, 256, 200, 0, 2), (7, 4, 5, 7, 64, 64, 0, 1), (9, 2, 12, 20, 256, 64, 0, 2)]:
    case(*c)
