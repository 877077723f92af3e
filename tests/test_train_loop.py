"""Loop semantics of `train_model` (reference UNet/train.py:123-200), checked on CPU with a scripted stand-in for the model:
lr/10 warm-up over min(1000, N) steps in epoch 0, N+1 optimizer steps per epoch (`if step > N: break`), test-epoch length from
the PER-REPLICA batch size (floor(count / batch_size) + 1 steps), test_loss.csv rewritten every epoch, checkpoint only on a new
best test loss, early stopping counted from the FIRST epoch within 1e-4 of the best.  The expected trace was RECORDED FROM THE
REFERENCE'S OWN `train_model` (tests/golden/make_train_loop_golden.py imports UNet/train.py in the build container with recording
stand-ins for TensorFlow, model.UNet and imagereader.ImageReader; tests/golden/train_loop_trace.json), one case with two replicas.
Also covers the reader-side flags (--balance_classes, --reader_count) and the rank-strided test readers."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import pkg


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


GOLDEN = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "train_loop_trace.json")))["traces"]


class RecordingReader:
    """SyntheticReader that records how the loop drives it (worker streams, start-up / shut-down)."""

    def __init__(self, who, events, count, seed):
        self.who, self.events, self.calls = who, events, []
        self.r = pkg("readers").SyntheticReader(count, 16, 16, 3, 2, seed=seed)

    def startup(self):
        self.events.append("startup " + self.who)

    def shutdown(self):
        self.events.append("shutdown " + self.who)

    def get_image_count(self):
        return self.r.get_image_count()

    def get_image_size(self):
        return self.r.get_image_size()

    def batches(self, batch_size, classmap=False, pin=True, raw=False, worker=0, num_workers=1):
        self.calls.append((batch_size, worker, num_workers))
        return self.r.batches(batch_size, classmap=classmap, pin=pin, raw=raw, worker=worker, num_workers=num_workers)


def _expand_lr(rle):
    return [v for v, n in rle for _ in range(n)]


def _run_product_loop(case, out_dir):
    """this repository's train_model on the golden case, with the scripted model -> a trace in the golden file's terms"""
    train = pkg("train")
    ScriptedUNet.script = case["script"]
    events = []
    tr = RecordingReader("train_db", events, 64, 1)
    te = RecordingReader("test_db", events, case["count"], 2)
    out = train.train_model(out_dir, case["batch"], case["readers"], "train_db", "test_db", 0, 2, 0, 3e-4, case["n"], case["stop"],
                            train_reader=tr, test_reader=te, quiet=True, unet_factory=ScriptedUNet)
    got = ScriptedUNet.last.finish()
    got.update(unet_args=list(ScriptedUNet.last.args), events=events, reader_calls={"train_db": tr.calls, "test_db": te.calls}, returned=[str(v) for v in out])
    return got


def _check_against_reference_trace(got, exp, out_dir, rank=0):
    case = exp["case"]
    R = case["replicas"]
    assert got["train_steps"] == exp["train_steps"]
    assert got["test_steps"] == exp["test_steps"]
    assert got["lr"] == pytest.approx(_expand_lr(exp["lr"]), rel=1e-12)
    assert got["ckpt_epochs"] == (exp["ckpt_epochs"] if rank == 0 else [])          # rank 0 writes the files
    # model constructor: (number_classes, GLOBAL batch, channels of the training reader, learning rate)  (UNet/train.py:61,93-94)
    assert got["unet_args"] == exp["unet_args"] and exp["unet_args"][1] == case["batch"] * R
    # readers: the reference hands reader_count x replicas workers to each reader (UNet/train.py:63-76); here every rank opens
    # `reader_count` worker streams with GLOBAL ids out of reader_count x replicas, at the per-replica batch size
    for who in ("train_db", "test_db"):
        total = exp["reader_kwargs"][who]["num_workers"]
        assert total == case["readers"] * R
        assert got["reader_calls"][who] == [(case["batch"], rank * case["readers"] + w, total) for w in range(case["readers"])]
        assert exp["dataset"][who]["batch"] == case["batch"] * R            # the reference's global batch = replicas x this rank's batch
    # readers are started before the first step and shut down at the end, train first (UNet/train.py:78-83,201-206)
    assert got["events"] == exp["events"]
    if rank != 0:
        return
    # files: test_loss.csv holds the reference's TEXT (fp32 means printed by numpy), the checkpoint lands where the reference writes it,
    # and the scalar log carries the reference's tags and step numbers
    assert open(os.path.join(out_dir, "test_loss.csv")).read() == exp["csv_text"]
    assert "\n".join(got["returned"]) + "\n" == exp["csv_text"]
    assert os.path.exists(os.path.join(out_dir, exp["ckpt_relpath"] + ".marker"))
    entries = sorted("tensorboard-*" if e.startswith("tensorboard-") else e for e in os.listdir(out_dir))
    assert entries == exp["output_entries"]
    tb = [e for e in os.listdir(out_dir) if e.startswith("tensorboard-")][0]
    for kind in ("train", "test"):
        rows = [json.loads(l) for l in open(os.path.join(out_dir, tb, kind, "scalars.jsonl"))]
        steps = exp["scalars"][kind]["steps"]
        if kind == "train":
            steps = [a + i for a, n in steps for i in range(n)]
        assert [(r["tag"], r["step"]) for r in rows] == [(t, s) for s in steps for t in exp["scalars"][kind]["tags"]]


@pytest.mark.parametrize("index", [i for i, t in enumerate(GOLDEN) if t["case"]["replicas"] == 1])
def test_train_model_loop_reproduces_the_reference_trace(tmp_path, index):
    """tests/golden/train_loop_trace.json was recorded from the reference's own train_model (tests/golden/make_train_loop_golden.py)."""
    exp = GOLDEN[index]
    got = _run_product_loop(exp["case"], str(tmp_path))
    _check_against_reference_trace(got, exp, str(tmp_path))
    # properties spelled out as well: N+1 steps, warm-up epoch of min(1000, N) + 1 steps at lr / 10, test length from the per-replica batch
    case = exp["case"]
    assert got["train_steps"][0] == min(1000, case["n"]) + 1 and all(v == case["n"] + 1 for v in got["train_steps"][1:])
    assert all(v == case["count"] // case["batch"] + 1 for v in got["test_steps"])
    assert got["lr"][0] == pytest.approx(3e-4 / 10) and got["lr"][-1] == pytest.approx(3e-4)


def _rank_worker(rank, world, port, index, out_dir, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank))
    try:
        import torch.distributed as dist
        mine = os.path.join(out_dir, "rank%d" % rank)
        got = _run_product_loop(GOLDEN[index]["case"], mine)
        _check_against_reference_trace(got, GOLDEN[index], mine, rank)
        q.put((rank, "ok"))
        if dist.is_initialized():
            dist.destroy_process_group()
    except Exception:                           # surface the failure in the parent
        import traceback
        q.put((rank, traceback.format_exc()))


@pytest.mark.parametrize("index", [i for i, t in enumerate(GOLDEN) if t["case"]["replicas"] > 1])
def test_train_model_loop_two_replicas_reproduces_the_reference_trace(tmp_path, index):
    """The reference's MirroredStrategy run with R replicas in one process == R processes here (gloo on the CPU): every rank takes the
    reference's decisions (steps, learning rates, test length from the PER-REPLICA batch, stopping epoch); rank 0 writes its files."""
    import socket
    import torch.multiprocessing as mp
    world = GOLDEN[index]["case"]["replicas"]
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank_worker, args=(r, world, port, index, str(tmp_path), q)) for r in range(world)]
    [p.start() for p in procs]
    res = dict(q.get(timeout=180) for _ in procs)
    [p.join(30) for p in procs]
    assert res == {r: "ok" for r in range(world)}, res


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
