#!/usr/bin/env python
"""CLI with the reference's flag (train_lora.py:333-343):  python train_lora.py --config_file X.yaml
Same config / prompts YAML schema, same output files ({save.path}/{save.name}_{i}steps.safetensors,
{save.name}_last.safetensors).  `pretrained_model.name_or_path` is a local diffusers-format folder or
`synthetic:<sd15|sd21|sdxl|tiny>`.  Multi-GPU: `python -m torch.distributed.run --nproc-per-node N
train_lora.py --config_file X.yaml` (one process per GPU, one all-reduce of the LoRA gradients per step)."""
import argparse

from leco_amd.train import main

if __name__ == "__main__":
    parser = argparse.ArgumentParser()
    parser.add_argument("--config_file", required=True, help="Config file for training.")
    parser.add_argument("--save_state", action="store_true", help="also write {save.name}_state.pt (resumable state)")
    parser.add_argument("--resume", default=None, help="continue from a {save.name}_state.pt of the same config")
    parser.add_argument("--strict_reference", action="store_true",
                        help="LoRA parameters and optimizer state in train.precision, like the reference (default: fp32 masters)")
    main(parser.parse_args())
