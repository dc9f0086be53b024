"""LoRA adapter network for the MI355X UNet engine.

Mirrors the reference's ``lora.py`` interface -- ``LoRANetwork(unet, rank, multiplier, alpha,
train_method)``, ``prepare_optimizer_params()``, ``save_weights(file, dtype, metadata)``, the
``with network:`` on/off switch (lora.py:110-237) -- and reproduces its module discovery, key
names, tensor shapes and init (lora.py:49-95,158-199) so the emitted ``.safetensors`` drop into
the reference tooling / webui unchanged.

What is different is the storage: instead of monkey-patching ``org_module.forward``
(lora.py:97-106) with five tiny kernels per module, all LoRA matrices live in ONE flat fp32
slab (plus a bf16 shadow the MFMA kernels read and a flat fp32 gradient slab), and the UNet
engine fuses ``up(down(x)) * multiplier * scale`` into the base GEMM as an extra K tile.  The
per-module ``lora_down.weight`` / ``lora_up.weight`` Parameters are views into the slab, so
``torch.optim`` optimizers, ``state_dict()`` and the fused AdamW kernel all see the same memory,
and the data-parallel gradient exchange is a single all-reduce of the gradient slab.
"""
from __future__ import annotations

import math
import os
from typing import List, Literal, Optional

import torch
import torch.nn as nn
from safetensors.torch import save_file

from . import ops

UNET_TARGET_REPLACE_MODULE_TRANSFORMER = ["Transformer2DModel"]
UNET_TARGET_REPLACE_MODULE_CONV = ["ResnetBlock2D", "Downsample2D", "Upsample2D"]  # locon / c3lier
LORA_PREFIX_UNET = "lora_unet"
DEFAULT_TARGET_REPLACE = UNET_TARGET_REPLACE_MODULE_TRANSFORMER

TRAINING_METHODS = Literal["noxattn", "innoxattn", "selfattn", "xattn", "full"]


class LoRAModule(nn.Module):
    """Holds one (lora_down, lora_up, alpha) triple.  Shapes follow lora.py:62-88."""

    def __init__(self, lora_name: str, leaf_name: str, org_module: nn.Module, multiplier=1.0, lora_dim=4, alpha=1):
        super().__init__()
        self.lora_name = lora_name
        self.leaf_name = leaf_name
        self.lora_dim = lora_dim
        cls = org_module.__class__.__name__
        if cls == "Linear":
            in_dim, out_dim = org_module.in_features, org_module.out_features
            self.lora_down = nn.Linear(in_dim, lora_dim, bias=False)
            self.lora_up = nn.Linear(lora_dim, out_dim, bias=False)
        elif cls == "Conv2d":
            in_dim, out_dim = org_module.in_channels, org_module.out_channels
            self.lora_dim = min(self.lora_dim, in_dim, out_dim)
            if self.lora_dim != lora_dim:
                print(f"{lora_name} dim (rank) is changed to: {self.lora_dim}")
            self.lora_down = nn.Conv2d(in_dim, self.lora_dim, org_module.kernel_size, org_module.stride,
                                       org_module.padding, bias=False)
            self.lora_up = nn.Conv2d(self.lora_dim, out_dim, (1, 1), (1, 1), bias=False)
        else:
            raise ValueError(f"unsupported LoRA target class {cls}")
        if isinstance(alpha, torch.Tensor):
            alpha = alpha.detach().float().item()
        alpha = lora_dim if alpha is None or alpha == 0 else alpha
        self.scale = alpha / self.lora_dim
        self.register_buffer("alpha", torch.tensor(alpha))
        nn.init.kaiming_uniform_(self.lora_down.weight, a=math.sqrt(5))
        nn.init.zeros_(self.lora_up.weight)
        self.multiplier = multiplier
        self.down_off = self.up_off = -1  # element offsets into the network slab


class LoRANetwork(nn.Module):
    def __init__(self, unet, rank: int = 4, multiplier: float = 1.0, alpha: float = 1.0,
                 train_method: TRAINING_METHODS = "full", target_replace_modules: Optional[List[str]] = None,
                 strict_reference: bool = False, strict_dtype: torch.dtype = torch.bfloat16) -> None:
        super().__init__()
        self.multiplier = multiplier
        self.lora_dim = rank
        self.alpha = alpha
        self.strict_reference = strict_reference
        self.strict_dtype = strict_dtype          # the precision `--strict_reference` rounds the parameters to (train.precision)
        targets = list(DEFAULT_TARGET_REPLACE if target_replace_modules is None else target_replace_modules)
        self.unet_loras: List[LoRAModule] = self.create_modules(LORA_PREFIX_UNET, unet, targets, rank, multiplier,
                                                               train_method)
        print(f"create LoRA for U-Net: {len(self.unet_loras)} modules.")
        names = set()
        for lora in self.unet_loras:
            assert lora.lora_name not in names, f"duplicated lora name: {lora.lora_name}. {names}"
            names.add(lora.lora_name)
        for lora in self.unet_loras:
            self.add_module(lora.lora_name, lora)
        self.version = 0
        self._packed_version = -1
        self._unet = [unet]  # not registered as a submodule
        self._build_slab(torch.device("cpu"))
        if hasattr(unet, "engine") and next(unet.parameters()).device.type != "meta":
            if unet.device != self.slab.device:
                self._adopt(self.slab.detach().to(unet.device))
            unet.engine().attach_lora(self)

    # ---- discovery (lora.py:158-199, including its name-filter semantics) ---------------------------
    def create_modules(self, prefix, root_module, target_replace_modules, rank, multiplier, train_method) -> list:
        loras = []
        for name, module in root_module.named_modules():
            if train_method == "noxattn":
                if "attn2" in name or "time_embed" in name:
                    continue
            elif train_method == "innoxattn":
                if "attn2" in name:
                    continue
            elif train_method == "selfattn":
                if "attn1" not in name:
                    continue
            elif train_method == "xattn":
                if "attn2" not in name:
                    continue
            elif train_method == "full":
                pass
            else:
                raise NotImplementedError(f"train_method: {train_method} is not implemented.")
            if module.__class__.__name__ in target_replace_modules:
                for child_name, child_module in module.named_modules():
                    if child_module.__class__.__name__ in ["Linear", "Conv2d"]:
                        leaf = name + "." + child_name
                        lora_name = (prefix + "." + leaf).replace(".", "_")
                        loras.append(LoRAModule(lora_name, leaf, child_module, multiplier, rank, self.alpha))
        return loras

    # ---- flat storage ---------------------------------------------------------------------------------
    def _build_slab(self, device) -> None:
        off = 0
        for lora in self.unet_loras:
            lora.down_off = off
            off += lora.lora_down.weight.numel()
            lora.up_off = off
            off += lora.lora_up.weight.numel()
        self.numel = off
        if self.unet_loras and self.unet_loras[0].lora_down.weight.device.type == "meta":
            self.slab = None   # shape-only instance (built under torch.device("meta")): names / census only
            return
        pad = (-off) % 64
        slab = torch.zeros(off + pad, dtype=torch.float32, device=device)
        for lora in self.unet_loras:
            for w, o in ((lora.lora_down.weight, lora.down_off), (lora.lora_up.weight, lora.up_off)):
                slab[o:o + w.numel()].copy_(w.detach().reshape(-1).float())
        self._adopt(slab)

    def _adopt(self, slab: torch.Tensor) -> None:
        dev = slab.device
        if self.strict_reference:  # reference keeps the parameters and the AdamW state in train.precision (train_lora.py:78,89)
            slab = slab.to(getattr(self, "strict_dtype", torch.bfloat16)).float()
        self.slab = slab.requires_grad_(True)         # autograd anchor of the engine's Function
        self.grad = torch.zeros_like(slab)
        self.shadow = slab.detach().to(torch.bfloat16)
        self.exp_avg = torch.zeros_like(slab)
        self.exp_avg_sq = torch.zeros_like(slab)
        self.hyper = torch.zeros(4, dtype=torch.float32, device=dev)
        data = slab.detach()
        for lora in self.unet_loras:
            for mod, o in ((lora.lora_down, lora.down_off), (lora.lora_up, lora.up_off)):
                shape = mod.weight.shape
                mod.weight.data = data[o:o + mod.weight.numel()].view(shape)
                mod.weight.grad = None
            lora.alpha = lora.alpha.to(dev)
        self.version += 1

    def to(self, *args, **kwargs):
        """Moves the slab; the fp32 master stays fp32 (the requested dtype only selects the
        compute/saved precision, which is bf16 on the MFMA path).  The reference's
        ``network.to(DEVICE, dtype=weight_dtype)`` call (train_lora.py:72-78) therefore works."""
        device = kwargs.get("device")
        for a in args:
            if isinstance(a, (str, torch.device)):
                device = a
        if device is not None and torch.device(device) != self.slab.device:
            self._adopt(self.slab.detach().to(device))
            unet = self._unet[0]
            if hasattr(unet, "engine") and torch.device(device).type != "meta":
                unet.engine().attach_lora(self)
        return self

    def attach_grads(self) -> None:
        """Expose the flat gradient slab as the per-parameter ``.grad`` views torch optimizers expect."""
        g = self.grad
        for lora in self.unet_loras:
            for mod, o in ((lora.lora_down, lora.down_off), (lora.lora_up, lora.up_off)):
                if mod.weight.grad is None:
                    mod.weight.grad = g[o:o + mod.weight.numel()].view(mod.weight.shape)

    def grads_cleared(self) -> bool:
        return self.unet_loras[0].lora_down.weight.grad is None

    def mark_updated(self) -> None:
        """Call after the LoRA parameters changed (optimizer step): the bf16 shadow and the packed
        MFMA operands are refreshed lazily before the next LoRA-on pass."""
        self.version += 1

    def sync_shadow(self) -> None:
        ops.cast_f32_bf16(self.slab.detach(), self.shadow, self.slab.numel()).run()

    # ---- reference API ----------------------------------------------------------------------------------
    def prepare_optimizer_params(self):
        all_params = []
        if self.unet_loras:
            params = []
            [params.extend(lora.parameters()) for lora in self.unet_loras]
            all_params.append({"params": params})
        return all_params

    def save_weights(self, file, dtype=None, metadata: Optional[dict] = None):
        state_dict = self.state_dict()
        if dtype is not None:
            for key in list(state_dict.keys()):
                state_dict[key] = state_dict[key].detach().clone().to("cpu").to(dtype)
        for key in list(state_dict.keys()):
            if not key.startswith("lora"):
                del state_dict[key]
        if os.path.splitext(str(file))[1] == ".safetensors":
            save_file({k: v.contiguous() for k, v in state_dict.items()}, str(file), metadata)
        else:
            torch.save(state_dict, file)

    def load_weights(self, file, strict: bool = True) -> None:
        """Inverse of `save_weights` (the reference has none): fills `lora_down` / `lora_up` from a file written by
        this package or by the reference (same key names, Appendix E); `alpha` must agree with the construction."""
        if os.path.splitext(str(file))[1] == ".safetensors":
            from safetensors.torch import load_file
            sd = load_file(str(file))
        else:
            sd = torch.load(file, map_location="cpu", weights_only=True)
        own = self.state_dict()
        missing = [k for k in own if k.startswith("lora") and k not in sd]
        unexpected = [k for k in sd if k not in own]
        if strict and (missing or unexpected):
            raise KeyError(f"{file}: LoRA keys differ (missing {missing[:3]}, unexpected {unexpected[:3]})")
        with torch.no_grad():
            for k, v in sd.items():
                if k not in own:
                    continue
                if k.endswith(".alpha"):
                    if strict and abs(float(v) - float(own[k])) > 1e-6:
                        raise ValueError(f"{file}: {k} = {float(v)} but the network was built with {float(own[k])}")
                    continue
                own[k].copy_(v.to(own[k].dtype).reshape(own[k].shape))
        self.mark_updated()
        self.sync_shadow()

    def needs_repack(self) -> bool:
        """True when the packed MFMA operand images are older than the parameters."""
        return self._packed_version != self.version

    def __enter__(self):
        self.multiplier = 1.0
        for lora in self.unet_loras:
            lora.multiplier = 1.0

    def __exit__(self, exc_type, exc_value, tb):
        self.multiplier = 0
        for lora in self.unet_loras:
            lora.multiplier = 0


class _ForeignModuleView:
    """One forward-patching LoRA module of another implementation, seen through the fields the engine needs."""

    def __init__(self, leaf_name: str, fm):
        self.fm = fm
        self.leaf_name = leaf_name
        self.lora_name = getattr(fm, "lora_name", LORA_PREFIX_UNET + "_" + leaf_name.replace(".", "_"))
        self.lora_down, self.lora_up = fm.lora_down, fm.lora_up
        self.lora_dim = int(fm.lora_down.weight.shape[0])
        self.scale = float(fm.scale)
        self.alpha = fm.alpha if hasattr(fm, "alpha") else torch.tensor(self.scale * self.lora_dim)
        self.down_off = self.up_off = -1

    @property
    def multiplier(self):
        return self.fm.multiplier

    @multiplier.setter
    def multiplier(self, v):
        self.fm.multiplier = v

    def parameters(self):
        return [self.lora_down.weight, self.lora_up.weight]


class ForeignLoRANetwork(LoRANetwork):
    """The reference's LoRA-injection seam (lora.py:97-106): its own ``lora.LoRANetwork`` works by
    re-assigning ``org_module.forward`` of every target leaf.  This UNet never calls leaf modules, so instead of
    ignoring (or refusing) such a network the engine ADOPTS it: the foreign ``lora_down`` / ``lora_up`` Parameters become
    views into a flat fp32 slab exactly like this package's own modules (`_adopt`), the low-rank products run fused in
    the GEMMs, gradients land in ``param.grad`` views of the gradient slab -- the foreign optimizer, ``with network:``
    switch (its ``multiplier`` fields are read at every forward), ``state_dict()`` and ``save_weights`` keep working on
    the objects the caller created.  The parameters can change behind the engine's back (a torch optimizer stepping the
    views), so the operand images are re-packed before every LoRA-on pass."""

    def __init__(self, unet, patched):      # patched: [(leaf qualified name, foreign LoRA module)]
        nn.Module.__init__(self)
        self.strict_reference = False
        self.unet_loras = [_ForeignModuleView(n, fm) for n, fm in patched]
        self.lora_dim = self.unet_loras[0].lora_dim
        self.alpha = float(self.unet_loras[0].scale * self.lora_dim)
        # the packed operand images carry ONE rank and ONE scale per GEMM site (fused q|k|v: three modules): modules that
        # share a site must agree -- which the reference's own network guarantees (one rank / alpha for all, lora.py:118-127)
        for l in self.unet_loras:
            if l.lora_dim != self.lora_dim or abs(l.scale - self.unet_loras[0].scale) > 1e-12:
                raise ValueError(f"adopted LoRA modules must share rank and alpha: {l.lora_name} has rank {l.lora_dim} / scale "
                                 f"{l.scale}, the first module rank {self.lora_dim} / scale {self.unet_loras[0].scale}")
        self.version = 0
        self._packed_version = -1
        self._unet = [unet]
        self._build_slab(unet.device)
        unet.engine().attach_lora(self)

    @property
    def multiplier(self):
        return self.unet_loras[0].multiplier

    @multiplier.setter
    def multiplier(self, v):
        for lora in self.unet_loras:
            lora.multiplier = v

    def needs_repack(self) -> bool:
        return True


def adopt_forward_patches(unet) -> Optional[ForeignLoRANetwork]:
    """Finds leaves whose ``forward`` was re-assigned to a bound method of an object carrying ``lora_down`` / ``lora_up`` /
    ``multiplier`` / ``scale`` (the reference's LoRAModule.apply_to, lora.py:97-100) and adopts them (None if there are
    none).  Leaves patched by anything else raise: the launch plans would silently ignore such a patch."""
    patched = []
    for name, m in unet.named_modules():
        if isinstance(m, (nn.Linear, nn.Conv2d)) and "forward" in m.__dict__:
            fm = getattr(m.__dict__["forward"], "__self__", None)
            if fm is None or not all(hasattr(fm, a) for a in ("lora_down", "lora_up", "multiplier", "scale")):
                raise RuntimeError(
                    f"{name}.forward has been re-assigned by something that is not a LoRA module (lora_down / lora_up / "
                    "multiplier / scale): this UNet executes static launch plans and never calls leaf modules.")
            patched.append((name, fm))
    if not patched:
        return None
    return ForeignLoRANetwork(unet, patched)
