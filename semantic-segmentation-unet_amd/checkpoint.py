"""Checkpoint write / restore under the reference's path stem `<out>/checkpoint/ckpt` (reference UNet/train.py:96,184
`tf.train.Checkpoint(optimizer=..., model=...).write(stem)`; UNet/model.py:81-83 `.restore(stem).expect_partial()`).

The files are the reference's: a TensorFlow TensorBundle `<stem>.index` + `<stem>.data-00000-of-00001` with object-graph keys
(tf_checkpoint.py) holding the model variables in the Keras layouts, the BatchNorm moving statistics, Adam's hyper-parameter
variables (`iter`, `beta_1`, `beta_2`, `decay`, `learning_rate`) and its `m` / `v` slots -- so a checkpoint trained by the reference
loads here and one written here has the structure the reference's restore walks.  TensorFlow is not available in this environment:
the format is restated from its specification and verified by round trip + checksums only.
Restoring follows `expect_partial()`: model variables are required, everything of the optimizer is optional.  Round-1 checkpoints
(`<stem>.npz`) still load."""
import os

import numpy as np
import torch

from . import engine as _engine
from . import tf_checkpoint as tfc


def _stem(path):
    for ext in (".npz", ".index"):
        if path.endswith(ext):
            return path[:-len(ext)]
    return path


def save_checkpoint(unet, checkpoint_filepath):
    e = unet.engine
    prm = e.export_parameters()
    t = {}
    for stem, eng_name, _ in tfc.variable_keys(e.layers):
        t[stem + tfc.ATTR] = prm[eng_name]
        if eng_name in e.slices:                                   # trainable: Adam slots
            o, n, shape = e.slices[eng_name]
            t[stem + "/.OPTIMIZER_SLOT/optimizer/m" + tfc.ATTR] = e.adam_m[o:o + n].view(shape).cpu().numpy()
            t[stem + "/.OPTIMIZER_SLOT/optimizer/v" + tfc.ATTR] = e.adam_v[o:o + n].view(shape).cpu().numpy()
    t["optimizer/iter" + tfc.ATTR] = np.array(e.iterations, np.int64)
    t["optimizer/beta_1" + tfc.ATTR] = np.array(_engine.ADAM_BETA1, np.float32)
    t["optimizer/beta_2" + tfc.ATTR] = np.array(_engine.ADAM_BETA2, np.float32)
    t["optimizer/decay" + tfc.ATTR] = np.array(0.0, np.float32)
    t["optimizer/learning_rate" + tfc.ATTR] = np.array(float(unet.optimizer.learning_rate), np.float32)
    t[tfc.OBJECT_GRAPH_KEY.decode()] = tfc.encode_object_graph(e.layers)
    tfc.write_bundle(_stem(checkpoint_filepath), t)


def _load_npz(unet, path):
    e = unet.engine
    z = np.load(path)
    e.load_parameters({k[len("model/"):]: z[k] for k in z.files if k.startswith("model/")})
    for k, (o, n, shape) in e.slices.items():
        if "optimizer/m/" + k in z.files:
            e.adam_m[o:o + n].view(shape).copy_(torch.as_tensor(z["optimizer/m/" + k]))
            e.adam_v[o:o + n].view(shape).copy_(torch.as_tensor(z["optimizer/v/" + k]))
    if "optimizer/iterations" in z.files:
        e.iterations = int(z["optimizer/iterations"])


def load_checkpoint(unet, checkpoint_filepath):
    """restore(...).expect_partial(): model variables are required, optimizer state is optional."""
    e = unet.engine
    stem = _stem(checkpoint_filepath)
    if not os.path.exists(stem + ".index"):
        if os.path.exists(stem + ".npz"):
            return _load_npz(unet, stem + ".npz")
        raise IOError("no checkpoint at %s (expected %s.index + %s.data-00000-of-00001, the tf.train.Checkpoint files)"
                      % (checkpoint_filepath, stem, stem))
    b = tfc.read_bundle(stem)
    # TensorFlow matches variables by walking the saved object graph, not by key strings: do the same when the graph is there
    keys = {}
    graph = b.get(tfc.OBJECT_GRAPH_KEY.decode())
    if isinstance(graph, (bytes, bytearray)):
        keys = tfc.resolve_keys_through_object_graph(graph, e.layers)
    for stem_k, eng_name, _ in tfc.variable_keys(e.layers):        # canonical key strings as the fallback
        keys.setdefault(eng_name, stem_k + tfc.ATTR)
        keys.setdefault("optimizer/m/" + eng_name, stem_k + "/.OPTIMIZER_SLOT/optimizer/m" + tfc.ATTR)
        keys.setdefault("optimizer/v/" + eng_name, stem_k + "/.OPTIMIZER_SLOT/optimizer/v" + tfc.ATTR)
    for h in tfc.OPT_HYPER:
        keys.setdefault("optimizer/" + h, "optimizer/%s%s" % (h, tfc.ATTR))
    values = {}
    for _, eng_name, _ in tfc.variable_keys(e.layers):
        k = keys[eng_name]
        if k not in b:
            raise KeyError("checkpoint %s has no variable %s (needed for %s)" % (stem, k, eng_name))
        want = tuple(e.p[eng_name].shape) if eng_name in e.p else tuple(e.moving[eng_name].shape)
        if tuple(b[k].shape) != want:
            raise ValueError("checkpoint variable %s has shape %s, the model needs %s (number_classes / number_channels differ?)"
                             % (k, tuple(b[k].shape), want))
        values[eng_name] = b[k]
    e.load_parameters(values)
    for eng_name, (o, n, shape) in e.slices.items():
        km, kv = keys["optimizer/m/" + eng_name], keys["optimizer/v/" + eng_name]
        if km in b and kv in b:
            e.adam_m[o:o + n].view(shape).copy_(torch.as_tensor(b[km].astype(np.float32)))
            e.adam_v[o:o + n].view(shape).copy_(torch.as_tensor(b[kv].astype(np.float32)))
    if keys["optimizer/iter"] in b:
        e.iterations = int(b[keys["optimizer/iter"]])
    if keys["optimizer/learning_rate"] in b:
        unet.optimizer.learning_rate = float(b[keys["optimizer/learning_rate"]])
