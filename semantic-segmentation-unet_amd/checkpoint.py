"""Checkpoint write / restore under the reference's path stem `<out>/checkpoint/ckpt` (reference UNet/train.py:96,184;
UNet/model.py:81-83).  The reference stores a TensorFlow object-graph bundle (ckpt.index + ckpt.data-*); reading that
format is a "next" row.  This build writes one `<stem>.npz` holding the same variables under
Keras-style names in the Keras layouts: model weights, BN moving statistics, Adam slots and the iteration counter."""
import os

import numpy as np
import torch


def _stem(path):
    return path[:-4] if path.endswith(".npz") else path


def save_checkpoint(unet, checkpoint_filepath):
    e = unet.engine
    out = {"model/" + k: v for k, v in e.export_parameters().items()}
    for k, (o, n, shape) in e.slices.items():
        out["optimizer/m/" + k] = e.adam_m[o:o + n].view(shape).cpu().numpy()
        out["optimizer/v/" + k] = e.adam_v[o:o + n].view(shape).cpu().numpy()
    out["optimizer/iterations"] = np.int64(e.iterations)
    out["optimizer/learning_rate"] = np.float64(unet.optimizer.learning_rate)
    d = os.path.dirname(_stem(checkpoint_filepath))
    if d:
        os.makedirs(d, exist_ok=True)
    np.savez(_stem(checkpoint_filepath) + ".npz", **out)


def load_checkpoint(unet, checkpoint_filepath):
    """restore(...).expect_partial(): model variables are required, optimizer slots are optional."""
    e = unet.engine
    z = np.load(_stem(checkpoint_filepath) + ".npz")
    e.load_parameters({k[len("model/"):]: z[k] for k in z.files if k.startswith("model/")})
    for k, (o, n, shape) in e.slices.items():
        if "optimizer/m/" + k in z.files:
            e.adam_m[o:o + n].view(shape).copy_(torch.as_tensor(z["optimizer/m/" + k]))
            e.adam_v[o:o + n].view(shape).copy_(torch.as_tensor(z["optimizer/v/" + k]))
    if "optimizer/iterations" in z.files:
        e.iterations = int(z["optimizer/iterations"])
