"""Two graph replays of the full-size SD1.5 forward: bitwise equal without the producer-side GroupNorm statistics
(LECO_GN_FUSED=0: no atomics in the forward), noise-level apart with them (fp32 atomics)."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from leco_amd import model_util  # noqa: E402
from leco_amd.unet import UNet2DConditionModel  # noqa: E402

dev = torch.device("cuda:0")
m = model_util.init_synthetic_(UNet2DConditionModel(model_util.SYNTHETIC["sd15"]()), 1234).to(dev, torch.bfloat16)
m.requires_grad_(False)
m.use_graphs = True
g = torch.Generator().manual_seed(0)
x = torch.randn(4, 4, 64, 64, generator=g).to(torch.bfloat16).to(dev)
ctx = torch.randn(4, 77, 768, generator=g).to(torch.bfloat16).to(dev)
ys = [m(x, torch.tensor(500), encoder_hidden_states=ctx).sample.float().clone() for _ in range(4)]
torch.cuda.synchronize()
rel = lambda a, b: ((a - b).norm() / b.norm()).item()
print(f"LECO_GN_FUSED={os.environ.get('LECO_GN_FUSED', 'auto')}: replays vs the first: "
      + ", ".join(f"{rel(y, ys[0]):.3g}{' (bitwise)' if torch.equal(y, ys[0]) else ''}" for y in ys[1:]))
