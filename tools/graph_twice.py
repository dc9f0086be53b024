"""Reproducer hunt: a SECOND UNet object capturing hipGraphs in one process (ROCm 7.2 segfault in hipStreamBeginCapture)."""
import gc, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from leco_amd import model_util
from leco_amd.unet import UNet2DConditionModel
import faulthandler; faulthandler.enable()
dev = torch.device("cuda:0")
mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
arch = sys.argv[2] if len(sys.argv) > 2 else "tiny"       # "sd15": the multi-GB case that crashes
cfg_fn = model_util.tiny_config if arch == "tiny" else model_util.SYNTHETIC[arch]
hw, cdim = (16, 64) if arch == "tiny" else (64, cfg_fn().cross_attention_dim)
x = torch.randn(2, 4, hw, hw, device=dev).to(torch.bfloat16); ctx = torch.randn(2, 77, cdim, device=dev).to(torch.bfloat16)
shared = torch.cuda.Stream() if mode == "shared" else None
if mode == "eager_first":      # does an EAGER run of the new model before its first capture avoid the crash?
    pass
for i in range(3 if arch == "tiny" else 2):
    m = model_util.init_synthetic_(UNet2DConditionModel(cfg_fn()), 1 + i).to(dev, torch.bfloat16)
    m.requires_grad_(False)
    m.use_graphs = True
    if shared is not None:
        m._capture_stream = shared
    if mode == "eager_first":
        m.use_graphs = False
        y = m(x, torch.tensor(10), encoder_hidden_states=ctx).sample
        torch.cuda.synchronize()
        m.use_graphs = True
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
