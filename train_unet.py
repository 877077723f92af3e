#!/usr/bin/env python3
"""`train_unet.py` (the name the reference's README and sbatch script use; README.md:75, UNet/sbatch_train.sh:83)."""
import importlib
importlib.import_module("semantic-segmentation-unet_amd.train").main()
