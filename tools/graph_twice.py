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
    if mode.startswith("fused"):
        # the case DESIGN.md section 6 describes: a whole FusedStep (three plans, five graphs) on a SECOND multi-GB model
        import contextlib, io
        from leco_amd import prompt_util, train_util
        from leco_amd.lora import LoRANetwork
        from leco_amd.scheduler import create_noise_scheduler
        from leco_amd.train import FusedStep
        with contextlib.redirect_stdout(io.StringIO()):
            net = LoRANetwork(m, rank=4, multiplier=1.0, alpha=1.0).to(dev)
        sched = create_noise_scheduler("ddim")
        e = {p_: torch.randn(1, 77, cdim, generator=torch.Generator().manual_seed(len(p_))).to(dev, torch.bfloat16) for p_ in ("a", "")}
        st = prompt_util.PromptSettings(target="a", positive="a", unconditional="", neutral="", action="erase", guidance_scale=1.0,
                                        resolution=hw * 8, batch_size=1)
        pair = prompt_util.PromptEmbedsPair(torch.nn.MSELoss(), e["a"], e["a"], e[""], e[""], st)
        fs = FusedStep(m, net, sched, 50, lr=1e-4)
        for _ in range(2):
            loss = fs.step(pair, 2, train_util.get_initial_latents(sched, 1, hw * 8, hw * 8, 1))
        torch.cuda.synchronize()
        print(mode, "model", i, "fused step ok", float(loss.item()), flush=True)
        if mode in ("fused_mm", "fused_conv", "fused_bwd"):
            # candidate triggers between two capture phases: the vendor GEMM (hipBLASLt / rocBLAS), the vendor convolution
            # (MIOpen), an autograd backward (torch's autograd worker thread)
            a_ = torch.randn(2048, 2048, device=dev)
            if mode == "fused_mm":
                for dt in (torch.float32, torch.bfloat16):
                    (a_.to(dt) @ a_.to(dt)).sum().item()
            elif mode == "fused_conv":
                xi = torch.randn(2, 64, 64, 64, device=dev)
                wt = torch.randn(64, 64, 3, 3, device=dev)
                torch.nn.functional.conv2d(xi, wt, padding=1).sum().item()
            else:
                q = a_.clone().requires_grad_(True)
                (q * q).sum().backward()
                q.grad.sum().item()
            print(mode, "model", i, "torch work between the captures done", flush=True)
        if mode == "fused_release":
            m.release()
            del fs, net
            torch.cuda.empty_cache()
        if mode == "fused_keep":
            globals().setdefault("alive", []).append((m, fs, net))
        del m
        gc.collect()
        continue
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
