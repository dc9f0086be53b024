#!/usr/bin/env python
"""SDXL entry point with the reference's flag (train_lora_xl.py:388-407):
    python train_lora_xl.py --config_file X.yaml
(`pretrained_model.name_or_path`: a local SDXL diffusers folder or `synthetic:sdxl` / `synthetic:tiny_xl`)."""
import argparse

from leco_amd.train import main

if __name__ == "__main__":
    parser = argparse.ArgumentParser()
    parser.add_argument("--config_file", required=True, help="Config file for training.")
    main(parser.parse_args(), xl=True)
