"""ORACLE (test infrastructure only) -- plain-PyTorch restatement of the reference's LoRA adapter
semantics (lora.py:44-106, :158-199, :231-237): module discovery by class name, key naming,
``down``/``up`` shapes, kaiming-uniform(a=sqrt 5) / zeros init, and the patched forward
``org(x) + up(down(x)) * multiplier * scale``.

Pinned against the reference itself: tests/test_reference_crosscheck.py imports
/root/reference/lora.py (through oracle/stub_diffusers) and checks names, shapes, init stream and
forward values are identical on the oracle UNet; tests/golden/ holds fixtures produced by the
reference's own code.  Only tests/, smoke() and bench.py's cpu_baseline may import this."""
import math

import torch
import torch.nn as nn

TRANSFORMER_TARGETS = ["Transformer2DModel"]
CONV_TARGETS = ["ResnetBlock2D", "Downsample2D", "Upsample2D"]


class LoRAModuleRef(nn.Module):
    def __init__(self, lora_name, org_module, multiplier=1.0, lora_dim=4, alpha=1.0):
        super().__init__()
        self.lora_name = lora_name
        self.lora_dim = lora_dim
        if org_module.__class__.__name__ == "Linear":
            self.lora_down = nn.Linear(org_module.in_features, lora_dim, bias=False)
            self.lora_up = nn.Linear(lora_dim, org_module.out_features, bias=False)
        else:
            cin, cout = org_module.in_channels, org_module.out_channels
            self.lora_dim = min(lora_dim, cin, cout)
            self.lora_down = nn.Conv2d(cin, self.lora_dim, org_module.kernel_size, org_module.stride,
                                       org_module.padding, bias=False)
            self.lora_up = nn.Conv2d(self.lora_dim, cout, (1, 1), (1, 1), bias=False)
        alpha = lora_dim if alpha is None or alpha == 0 else alpha
        self.scale = alpha / self.lora_dim
        self.register_buffer("alpha", torch.tensor(alpha))
        nn.init.kaiming_uniform_(self.lora_down.weight, a=math.sqrt(5))
        nn.init.zeros_(self.lora_up.weight)
        self.multiplier = multiplier
        self.org_forward = org_module.forward
        org_module.forward = self.forward

    def forward(self, x):
        return self.org_forward(x) + self.lora_up(self.lora_down(x)) * self.multiplier * self.scale


class LoRANetworkRef(nn.Module):
    def __init__(self, unet, rank=4, multiplier=1.0, alpha=1.0, targets=None):
        super().__init__()
        targets = TRANSFORMER_TARGETS if targets is None else targets
        self.unet_loras = []
        for name, module in unet.named_modules():
            if module.__class__.__name__ in targets:
                for child_name, child in module.named_modules():
                    if child.__class__.__name__ in ["Linear", "Conv2d"]:
                        lora_name = ("lora_unet." + name + "." + child_name).replace(".", "_")
                        self.unet_loras.append(LoRAModuleRef(lora_name, child, multiplier, rank, alpha))
        for lora in self.unet_loras:
            self.add_module(lora.lora_name, lora)

    def __enter__(self):
        for lora in self.unet_loras:
            lora.multiplier = 1.0

    def __exit__(self, *a):
        for lora in self.unet_loras:
            lora.multiplier = 0
