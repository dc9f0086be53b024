#!/usr/bin/env python
"""SDXL entry point with the reference's flag (train_lora_xl.py:388-407):
    python train_lora_xl.py --config_file X.yaml
(`pretrained_model.name_or_path`: a local SDXL diffusers folder or `synthetic:sdxl` / `synthetic:tiny_xl`)."""
import argparse

from leco_amd.train import main

if __name__ == "__main__":
    parser = argparse.ArgumentParser()
    parser.add_argument("--config_file", required=True, help="Config file for training.")
    parser.add_argument("--save_state", action="store_true", help="also write {save.name}_state.pt (resumable state)")
    parser.add_argument("--resume", default=None, help="continue from a {save.name}_state.pt of the same config")
    parser.add_argument("--strict_reference", action="store_true",
                        help="LoRA parameters and optimizer state in train.precision, like the reference (default: fp32 masters)")
    main(parser.parse_args(), xl=True)
