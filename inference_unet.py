#!/usr/bin/env python3
"""`inference_unet.py` (README.md:193 name for UNet/inference.py)."""
import importlib
importlib.import_module("semantic-segmentation-unet_amd.inference").main()
