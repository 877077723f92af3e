"""MI355X-native U-Net train / inference hot path (drop-in for UNet/model.py of the reference)."""
