"""Loop semantics of `train_model` (reference UNet/train.py:123-200), checked on CPU with a scripted stand-in for the model:
lr/10 warm-up over min(1000, N) steps in epoch 0, N+1 optimizer steps per epoch (`if step > N: break`), test-epoch length from
the PER-REPLICA batch size (floor(count / batch_size) + 1 steps), test_loss.csv rewritten every epoch, checkpoint only on a new
best test loss, early stopping counted from the FIRST epoch within 1e-4 of the best.  The expected trace comes from a direct
restatement of those reference lines below; also covers the reader-side flags (--balance_classes, --reader_count) and the
rank-strided test readers."""
import os

import numpy as np
import pytest
import torch

from conftest import pkg


def reference_loop(test_losses, learning_rate, test_every_n_steps, test_count, batch_size, early_stopping_count):
    """UNet/train.py:123-200 with the model calls replaced by a trace.  test_losses[e] = mean test loss of epoch e."""
    trace = {"lr": [], "train_steps": [], "test_steps": [], "ckpt_epochs": [], "csv": None}
    train_epoch_size = test_every_n_steps                       # :99
    test_epoch_size = test_count / batch_size                   # :100
    test_loss = []
    epoch = 0
    while True:
        if epoch == 0:                                          # :126-132
            cur = min(1000, train_epoch_size); lr = learning_rate / 10
        else:
            cur = train_epoch_size; lr = learning_rate
        n = 0
        step = 0
        while True:                                             # :136-141  for step, batch in enumerate(ds): if step > cur: break
            if step > cur:
                break
            trace["lr"].append(lr); n += 1; step += 1
        trace["train_steps"].append(n)
        n = 0
        step = 0
        while True:                                             # :153-156
            if step > test_epoch_size:
                break
            n += 1; step += 1
        trace["test_steps"].append(n)
        test_loss.append(test_losses[epoch])                    # :162
        trace["csv"] = list(test_loss)                          # :173-176
        if (len(test_loss) - 1) == np.argmin(test_loss):        # :181-184
            trace["ckpt_epochs"].append(epoch)
        error_from_best = np.abs(np.asarray(test_loss) - np.min(test_loss))      # :187-196
        error_from_best[error_from_best < 1e-4] = 0
        best_epoch = np.where(error_from_best == 0)[0][0]
        if len(test_loss) - best_epoch > early_stopping_count:  # :198-199
            break
        epoch += 1
    return trace


class ScriptedUNet:
    """What train_model touches of model.UNet; test losses come from a script, everything is recorded."""
    script = None
    last = None

    def __init__(self, number_classes, global_batch_size, number_channels, learning_rate, device=None, compute_dtype=None):
        ScriptedUNet.last = self
        self.args = (number_classes, global_batch_size, number_channels, learning_rate)
        self.lr = learning_rate
        self.trace = {"lr": [], "train_steps": [], "test_steps": [], "ckpt_epochs": []}
        self._train_n = self._test_n = 0
        self.epoch = 0
        self.parallel = None

    def set_learning_rate(self, lr):
        if self._test_n:                       # a new epoch begins: close the previous one
            self.trace["train_steps"].append(self._train_n); self.trace["test_steps"].append(self._test_n)
            self._train_n = self._test_n = 0
            self.epoch += 1
        self.lr = lr

    def dist_train_step(self, strategy, inputs):
        images, labels, lm, am = inputs
        assert images.dtype == torch.float32 and images.dim() == 4 and labels.dtype == torch.int32 and labels.dim() == 4
        self.trace["lr"].append(self.lr); self._train_n += 1
        lm.update_state(0.5); am.update_state(1.0, 2.0)
        return pkg("model")._Loss(torch.tensor([0.5]))

    def dist_test_step(self, strategy, inputs):
        self._test_n += 1
        inputs[2].update_state(ScriptedUNet.script[self.epoch]); inputs[3].update_state(1.0, 2.0)
        return pkg("model")._Loss(torch.tensor([ScriptedUNet.script[self.epoch]], dtype=torch.float64))

    def save_checkpoint(self, path):
        self.trace["ckpt_epochs"].append(self.epoch)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        open(path + ".marker", "w").write(str(self.epoch))

    def finish(self):
        self.trace["train_steps"].append(self._train_n); self.trace["test_steps"].append(self._test_n)
        return self.trace


@pytest.mark.parametrize("case", [
    dict(script=[0.9, 0.7, 0.70005, 0.8, 0.75, 0.9, 0.9], n=3, count=10, batch=4, stop=2),      # 0.70005 is within 1e-4 of the best: not a new best epoch
    dict(script=[0.5, 0.6, 0.7, 0.8], n=5, count=8, batch=2, stop=1),                            # never improves after epoch 0
    dict(script=[0.9, 0.8, 0.7, 0.6, 0.65, 0.66, 0.67], n=1200, count=3, batch=4, stop=2),       # warm-up capped at 1000 steps
])
def test_train_model_loop_matches_reference_semantics(tmp_path, case):
    train, readers = pkg("train"), pkg("readers")
    ScriptedUNet.script = case["script"]
    tr = readers.SyntheticReader(64, 16, 16, 1, 2, seed=1)
    te = readers.SyntheticReader(case["count"], 16, 16, 1, 2, seed=2)
    lr = 3e-4
    out = train.train_model(str(tmp_path), case["batch"], 1, None, None, 0, 2, 0, lr, case["n"], case["stop"],
                            train_reader=tr, test_reader=te, quiet=True, unet_factory=ScriptedUNet)
    got = ScriptedUNet.last.finish()
    exp = reference_loop(case["script"], lr, case["n"], case["count"], case["batch"], case["stop"])
    assert got["train_steps"] == exp["train_steps"]
    assert got["test_steps"] == exp["test_steps"]
    assert got["lr"] == pytest.approx(exp["lr"], rel=1e-12)
    assert got["ckpt_epochs"] == exp["ckpt_epochs"]
    assert out == pytest.approx(exp["csv"], rel=1e-6)
    csv = [float(v) for v in open(os.path.join(str(tmp_path), "test_loss.csv")).read().split()]
    assert csv == pytest.approx(exp["csv"], rel=1e-6)
    # properties spelled out (so a wrong restatement above cannot hide a wrong loop): N+1 steps, warm-up epoch, test length
    assert got["train_steps"][0] == min(1000, case["n"]) + 1 and all(v == case["n"] + 1 for v in got["train_steps"][1:])
    assert all(v == case["count"] // case["batch"] + 1 for v in got["test_steps"])
    assert got["lr"][0] == pytest.approx(lr / 10) and got["lr"][-1] == pytest.approx(lr)
    assert ScriptedUNet.last.args[1] == case["batch"]             # global batch = batch_size x replicas (1 here)


def _write_tiles(folder, masks):
    os.makedirs(folder, exist_ok=True)
    for i, m in enumerate(masks):
        np.save(os.path.join(folder, "t%02d.npy" % i), np.full((16, 16), float(i), np.float32))
        np.save(os.path.join(folder, "t%02d_mask.npy" % i), m)


def test_balance_classes_draws_classes_uniformly(tmp_path):
    # 9 background-only tiles, 1 tile containing class 1: unbalanced sampling sees it ~10 % of the time, balanced sampling
    # picks class 1 half of the time and class 0 (all ten tiles contain it) the other half -> 0.5 + 0.5/10 = 55 %
    # (reference UNet/imagereader.py:146-157,211-233)
    readers = pkg("readers")
    masks = [np.zeros((16, 16), np.uint8) for _ in range(10)]
    masks[7][:4, :4] = 1
    _write_tiles(str(tmp_path), masks)
    for balance, lo, hi in ((False, 0.06, 0.14), (True, 0.50, 0.60)):
        rd = readers.TileFolderReader(str(tmp_path), 2, shuffle=True, seed=3, balance_classes=balance)
        keys = rd.key_sequence()
        frac = np.mean([next(keys) == 7 for _ in range(4000)])
        assert lo < frac < hi, (balance, frac)
    # a class id the data never shows is re-drawn (:218-226); a class id beyond the key table raises like the reference
    rd = readers.TileFolderReader(str(tmp_path), 2, shuffle=True, seed=0, balance_classes=True)
    rd.keys = [rd.keys[0], []]
    keys = rd.key_sequence()
    assert all(0 <= next(keys) < 10 for _ in range(50))
    rd.keys = [rd.keys[0]]
    with pytest.raises(IndexError):
        next(rd.key_sequence())


def test_test_readers_stride_by_global_worker_id(tmp_path):
    # non-shuffled readers: worker w of W walks keys w, w+W, ... (reference UNet/imagereader.py:239-241,247); with
    # W = reader_count x replicas and global worker ids, the replicas' test slices are disjoint and cover the set
    readers = pkg("readers")
    _write_tiles(str(tmp_path), [np.zeros((16, 16), np.uint8) for _ in range(12)])
    rd = readers.TileFolderReader(str(tmp_path), 2, shuffle=False)
    world, reader_count = 2, 2
    seen = {}
    for rank in range(world):
        for w in range(reader_count):
            gid = rank * reader_count + w
            it = rd.key_sequence(gid, world * reader_count)
            seen[gid] = [next(it) for _ in range(3)]
    assert seen == {0: [0, 4, 8], 1: [1, 5, 9], 2: [2, 6, 10], 3: [3, 7, 11]}
    flat = sorted(v for vs in seen.values() for v in vs)
    assert flat == list(range(12))
    # and the batches carry those tiles (the tile value encodes its index; z-score of a constant tile is 0, so read raw)
    img, _ = next(rd.batches(3, classmap=True, pin=False, raw=True, worker=2, num_workers=4))
    assert [int(v) for v in img[:, 0, 0, 0]] == [2, 6, 10]


def test_reader_count_feeds_several_worker_streams(tmp_path):
    # --reader_count N -> N seeded worker iterators behind one DeviceFeed (CPU mode here): every batch comes from one of them
    train, readers = pkg("train"), pkg("readers")
    ScriptedUNet.script = [0.5, 0.6]
    tr = readers.SyntheticReader(64, 16, 16, 1, 2, seed=1)
    te = readers.SyntheticReader(4, 16, 16, 1, 2, seed=2)
    train.train_model(str(tmp_path), 2, 3, None, None, 0, 2, 0, 3e-4, 4, 0, train_reader=tr, test_reader=te, quiet=True,
                      unet_factory=ScriptedUNet)
    assert ScriptedUNet.last.finish()["train_steps"] == [5]
    with pytest.raises(ValueError):            # a reader without class balancing must not silently ignore --balance_classes 1
        train.train_model(str(tmp_path), 2, 1, None, None, 0, 2, 1, 3e-4, 4, 0, train_reader=tr, test_reader=te, quiet=True,
                          unet_factory=ScriptedUNet)


def test_train_model_takes_a_reference_style_reader_object(tmp_path):
    # train_model(train_reader=<object with the reference ImageReader's surface>) (UNet/train.py:66-90): batched by the adapter, fed
    # through the DeviceFeed, started and shut down by the loop, NOT augmented a second time on the device
    from test_feed import FakeReferenceReader
    train = pkg("train")
    ScriptedUNet.script = [0.5, 0.6]
    tr = FakeReferenceReader(64, 16, 16, 1, 2, seed=1)
    te = FakeReferenceReader(4, 16, 16, 1, 2, seed=2)
    out = train.train_model(str(tmp_path), 2, 1, None, None, 1, 2, 0, 3e-4, 3, 1,
                            train_reader=tr, test_reader=te, quiet=True, unet_factory=ScriptedUNet)
    trace = ScriptedUNet.last.finish()
    assert np.allclose(out, [0.5, 0.6]) and trace["train_steps"] == [4, 4] and trace["test_steps"] == [3, 3]
    assert tr.started and tr.stopped and te.started and te.stopped
    assert tr.handed_out >= 2 * 8
