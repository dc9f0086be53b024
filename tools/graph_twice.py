"""Reproducer hunt: a SECOND UNet object capturing hipGraphs in one process (ROCm 7.2 segfault in hipStreamBeginCapture)."""
import gc, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from leco_amd import model_util
from leco_amd.unet import UNet2DConditionModel
import faulthandler; faulthandler.enable()
dev = torch.device("cuda:0")
mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
x = torch.randn(2, 4, 16, 16, device=dev).to(torch.bfloat16); ctx = torch.randn(2, 77, 64, device=dev).to(torch.bfloat16)
shared = torch.cuda.Stream() if mode == "shared" else None
for i in range(3):
    m = model_util.init_synthetic_(UNet2DConditionModel(model_util.tiny_config()), 1 + i).to(dev, torch.bfloat16)
    m.requires_grad_(False)
    m.use_graphs = True
    if shared is not None:
        m._capture_stream = shared
    for _ in range(2):
        y = m(x, torch.tensor(10), encoder_hidden_states=ctx).sample
    torch.cuda.synchronize()
    print(mode, "model", i, "ok", float(y.float().abs().mean()), flush=True)
    if mode == "release":
        m.release()
    if mode == "keep":
        globals().setdefault("alive", []).append(m)
    del m
    gc.collect()
print(mode, "DONE", flush=True)
