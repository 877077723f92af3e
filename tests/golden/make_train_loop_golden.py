"""Writes tests/golden/train_loop_trace.json by running the REFERENCE's own `train_model` (UNet/train.py:33-206) on scripted inputs.
Run in the build container only (needs /root/reference):

    python tests/golden/make_train_loop_golden.py

`train_model` is host-side control flow around four collaborators -- TensorFlow's strategy / metrics / summary / checkpoint objects,
`model.UNet` and `imagereader.ImageReader`.  All four are replaced by recording stand-ins (this file), none of the reference's arithmetic
is involved: the network's test losses come from a script, and what is recorded is everything the loop DECIDES -- learning rate of every
optimizer step, optimizer steps per epoch, test steps per epoch, which epochs write a checkpoint and where, the text of test_loss.csv after
the last epoch, the tensorboard scalar tags and step numbers, the constructor arguments of the model and the readers, the dataset batch /
prefetch sizes, and the order of reader start-up and shut-down.  The fixture holds that trace (data only); tests/test_train_loop.py
requires this repository's `train.train_model` to reproduce it."""
import contextlib
import io
import json
import os
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/UNet"

# (script of per-epoch test losses, test_every_n_steps, test image count, per-replica batch, early_stopping_count, replicas, reader_count)
CASES = [
    dict(script=[0.9, 0.7, 0.70005, 0.8, 0.75, 0.9, 0.9], n=3, count=10, batch=4, stop=2, replicas=1, readers=1),   # 0.70005: within 1e-4 of the best
    dict(script=[0.5, 0.6, 0.7, 0.8], n=5, count=8, batch=2, stop=1, replicas=1, readers=1),                        # never improves after epoch 0
    dict(script=[0.9, 0.8, 0.7, 0.6, 0.65, 0.66, 0.67], n=1200, count=3, batch=4, stop=2, replicas=1, readers=1),   # warm-up capped at 1000 steps
    dict(script=[0.8, 0.6, 0.61, 0.59995, 0.7, 0.7, 0.7], n=4, count=12, batch=2, stop=2, replicas=2, readers=3),   # two replicas, three readers each
]

T = {}          # the trace of the running case


class _Strategy:
    def __init__(self):
        self.num_replicas_in_sync = T["case"]["replicas"]

    @contextlib.contextmanager
    def scope(self):
        yield

    def experimental_distribute_dataset(self, ds):
        T["distributed_datasets"] += 1
        return ds


class _Metric:
    def __init__(self, name, dtype=None):
        self.name, self.values = name, []

    def update_state(self, *a):
        self.values.append(a)

    def result(self):
        return np.float32(len(self.values))

    def reset_states(self):
        T["metric_resets"][self.name] = T["metric_resets"].get(self.name, 0) + 1
        self.values = []


class _Writer:
    current = None

    def __init__(self, log_dir):
        self.kind = os.path.basename(log_dir)
        assert os.path.basename(os.path.dirname(log_dir)).startswith("tensorboard-") and os.path.isdir(log_dir)

    @contextlib.contextmanager
    def as_default(self):
        _Writer.current = self
        yield
        _Writer.current = None


def _scalar(tag, value, step=None):
    T["scalars"][_Writer.current.kind].append([tag, int(step)])


class _Checkpoint:
    def __init__(self, optimizer=None, model=None):
        T["checkpoint_tracks"] = sorted(k for k, v in (("optimizer", optimizer), ("model", model)) if v is not None)

    def write(self, path):
        T["ckpt_epochs"].append(T["epoch"])
        T["ckpt_relpath"] = os.path.relpath(path, T["out"])
        os.makedirs(os.path.dirname(path), exist_ok=True)           # (TensorFlow creates the folder and ckpt.index / ckpt.data-* in it)
        open(path + ".marker", "w").write(str(T["epoch"]))


class _Loss:
    def __init__(self, v):
        self.v = v

    def numpy(self):
        return np.float32(self.v)          # TensorFlow hands back an fp32 scalar


class _UNet:
    def __init__(self, *a, **kw):
        T["unet_args"], T["unet_kwargs"] = list(a), kw
        self.lr = None

    def get_optimizer(self):
        return "optimizer"

    def get_keras_model(self):
        return "model"

    def set_learning_rate(self, lr):
        if T["test_n"]:
            _close_epoch()
        self.lr = lr

    def dist_train_step(self, strategy, inputs):
        assert isinstance(strategy, _Strategy) and len(inputs) == 4
        T["lr"].append(float(self.lr)); T["train_n"] += 1
        inputs[2].update_state(0.5); inputs[3].update_state(1, 2)
        return _Loss(0.5)

    def dist_test_step(self, strategy, inputs):
        T["test_n"] += 1
        inputs[2].update_state(0.5)
        return _Loss(T["case"]["script"][T["epoch"]])


def _close_epoch():
    T["train_steps"].append(T["train_n"]); T["test_steps"].append(T["test_n"])
    T["train_n"] = T["test_n"] = 0
    T["epoch"] += 1


class _Dataset:
    def __init__(self, who):
        self.who = who

    def batch(self, n):
        T["dataset"][self.who]["batch"] = int(n)
        return self

    def prefetch(self, n):
        T["dataset"][self.who]["prefetch"] = int(n)
        return self

    def __iter__(self):
        while True:
            yield ("images", "labels")


class _ImageReader:
    def __init__(self, path, **kw):
        self.who = path
        T["reader_kwargs"][path] = {k: (bool(v) if isinstance(v, (bool, np.bool_)) else int(v)) for k, v in kw.items()}

    def get_image_count(self):
        return T["case"]["count"] if self.who == "test_db" else 64

    def get_image_size(self):
        return [16, 16, 3]

    def startup(self):
        T["events"].append("startup " + self.who)

    def shutdown(self):
        T["events"].append("shutdown " + self.who)

    def get_tf_dataset(self):
        T["dataset"][self.who] = {}
        return _Dataset(self.who)


def install_stand_ins():
    tf = types.ModuleType("tensorflow")
    tf.__version__ = "2.0.0"
    tf.float32 = "float32"
    tf.distribute = types.SimpleNamespace(MirroredStrategy=_Strategy)
    tf.train = types.SimpleNamespace(Checkpoint=_Checkpoint)
    tf.summary = types.SimpleNamespace(create_file_writer=_Writer, scalar=_scalar)
    keras = types.ModuleType("tensorflow.keras")
    keras.metrics = types.SimpleNamespace(Mean=_Metric, CategoricalAccuracy=_Metric)
    mp = types.ModuleType("tensorflow.keras.mixed_precision")
    mp.experimental = types.ModuleType("tensorflow.keras.mixed_precision.experimental")
    keras.mixed_precision = mp
    tf.keras = keras
    sys.modules.update({"tensorflow": tf, "tensorflow.keras": keras, "tensorflow.keras.mixed_precision": mp,
                        "tensorflow.keras.mixed_precision.experimental": mp.experimental})
    m = types.ModuleType("model"); m.UNet = _UNet
    r = types.ModuleType("imagereader"); r.ImageReader = _ImageReader
    sys.modules.update({"model": m, "imagereader": r})


def main():
    install_stand_ins()
    sys.path.insert(0, REF)
    import train as ref_train                                        # the reference's module (UNet/train.py)
    traces = []
    for case in CASES:
        T.clear()
        T.update(case=case, lr=[], train_steps=[], test_steps=[], ckpt_epochs=[], scalars={"train": [], "test": []}, events=[],
                 reader_kwargs={}, dataset={}, metric_resets={}, distributed_datasets=0, train_n=0, test_n=0, epoch=0)
        with tempfile.TemporaryDirectory() as out:
            T["out"] = out
            with contextlib.redirect_stdout(io.StringIO()):
                # positional, as UNet/launch_train.py:42 calls it
                ref_train.train_model(out, case["batch"], case["readers"], "train_db", "test_db", 1, 2, 0, 3e-4, case["n"], case["stop"])
            _close_epoch()
            T["csv_text"] = open(os.path.join(out, "test_loss.csv")).read()
            T["output_entries"] = sorted("tensorboard-*" if e.startswith("tensorboard-") else e for e in os.listdir(out))
        keep = {k: T[k] for k in ("case", "unet_args", "unet_kwargs", "reader_kwargs", "dataset", "distributed_datasets", "lr", "train_steps",
                                  "test_steps", "ckpt_epochs", "ckpt_relpath", "checkpoint_tracks", "csv_text", "scalars", "events",
                                  "metric_resets", "output_entries")}
        # the learning rates are piecewise constant: run-length encode them (the 1200-step case would be 8000 numbers)
        rle = []
        for v in keep["lr"]:
            if rle and rle[-1][0] == v:
                rle[-1][1] += 1
            else:
                rle.append([v, 1])
        keep["lr"] = rle
        for kind in ("train", "test"):                               # tags alternate loss / accuracy at one step number: store the step once
            s = keep["scalars"][kind]
            assert all(s[i][0] == "loss" and s[i + 1][0] == "accuracy" and s[i][1] == s[i + 1][1] for i in range(0, len(s), 2))
            steps = [s[i][1] for i in range(0, len(s), 2)]
            if kind == "train":                                      # consecutive runs: [first, count]
                runs = []
                for v in steps:
                    if runs and runs[-1][0] + runs[-1][1] == v:
                        runs[-1][1] += 1
                    else:
                        runs.append([v, 1])
                steps = runs
            keep["scalars"][kind] = {"tags": ["loss", "accuracy"], "steps": steps}
        traces.append(keep)
        print(case, "-> epochs", len(keep["train_steps"]), "ckpt", keep["ckpt_epochs"], "csv", keep["csv_text"].split())
    path = os.path.join(HERE, "train_loop_trace.json")
    json.dump({"generator": "tests/golden/make_train_loop_golden.py", "source": "reference UNet/train.py:33-206 train_model, run with recording stand-ins",
               "traces": traces}, open(path, "w"), indent=1)
    print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1e3))


if __name__ == "__main__":
    main()
