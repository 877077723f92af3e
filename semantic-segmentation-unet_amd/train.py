#!/usr/bin/env python3
"""`train_unet` -- the caller of the hot path (reference UNet/train.py), re-hosted on the HIP engine + RCCL.

Flags are the reference's (UNet/train.py:213-233) plus opt-in extras that default to reference behaviour.  Behaviour kept:
global batch = batch_size x replicas (:61); epoch = `test_every_n_steps` optimizer steps, and because the reference breaks
on `step > N` an epoch runs N+1 steps (:137-138); epoch 0 is an Adam warm-up at lr/10 over min(1000, N) steps (:126-132);
the test pass runs while step <= image_count / batch_size (:100,154-156); `test_loss.csv` is rewritten every epoch
(:173-176); the checkpoint `<out>/checkpoint/ckpt` is written only on a new best test loss (:181-184); early stopping
counts epochs since the first loss within 1e-4 of the best (:187-199).  Differences: one process per GPU (launch with
torchrun) instead of MirroredStrategy; scalars go to `<out>/tensorboard-<time>/{train,test}/scalars.jsonl` (TensorBoard
event files only if `tensorboard` is importable); data comes from readers.py unless a compatible reader object is passed.
"""
import argparse
import datetime
import json
import os
import time

LABEL_CHECK_EVERY = 16          # train steps between polls of the device-side out-of-range-label counter (one 4-byte read)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # before HIP initialises: RCCL needs dmabuf IPC on this platform

import numpy as np
import torch
import torch.distributed as dist

from . import model as unet_model_module
from . import readers

CONVERGENCE_TOLERANCE = 1e-4


class _Scalars:
    def __init__(self, log_dir):
        os.makedirs(log_dir, exist_ok=True)
        self.f = open(os.path.join(log_dir, "scalars.jsonl"), "a")
        self.tb = None
        try:
            from torch.utils.tensorboard import SummaryWriter
            self.tb = SummaryWriter(log_dir)
        except Exception:
            pass

    def scalar(self, tag, value, step):
        self.f.write(json.dumps({"tag": tag, "value": float(value), "step": int(step)}) + "\n")
        self.f.flush()
        if self.tb is not None:
            self.tb.add_scalar(tag, float(value), int(step))


def best_epoch_index(test_loss):
    """First epoch whose loss is within the tolerance of the minimum (reference UNet/train.py:187-196)."""
    err = np.abs(np.asarray(test_loss, dtype=np.float64) - np.min(test_loss))
    err[err < CONVERGENCE_TOLERANCE] = 0
    return int(np.where(err == 0)[0][0])


def train_model(output_folder, batch_size, reader_count, train_lmdb_filepath, test_lmdb_filepath, use_augmentation,
                number_classes, balance_classes, learning_rate, test_every_n_steps, early_stopping_count,
                train_reader=None, test_reader=None, max_epochs=None, quiet=False, compute_dtype=None, unet_factory=None):
    """unet_factory (tests): callable(number_classes, global_batch_size, number_channels, learning_rate, device=..., compute_dtype=...)
    standing in for model.UNet, so the loop semantics can be checked without a GPU; the default is the HIP-backed class, which
    refuses to run anywhere but on an MI355X."""
    say = (lambda *a: None) if quiet else print
    for k, v in (("batch_size", batch_size), ("number_classes", number_classes), ("learning_rate", learning_rate),
                 ("test_every_n_steps", test_every_n_steps), ("balance_classes", balance_classes),
                 ("use_augmentation", use_augmentation), ("train_database", train_lmdb_filepath),
                 ("test_database", test_lmdb_filepath), ("output folder", output_folder),
                 ("early_stopping count", early_stopping_count), ("reader_count", reader_count)):
        say("{} = {}".format(k, v))
    os.makedirs(output_folder, exist_ok=True)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    have_gpu = torch.cuda.is_available()
    dev_ = torch.device("cuda", local) if have_gpu else torch.device("cpu")
    if have_gpu:
        torch.cuda.set_device(local)
    if world > 1 and not dist.is_initialized():
        dist.init_process_group("nccl", device_id=dev_) if have_gpu else dist.init_process_group("gloo")
    global_batch_size = batch_size * world

    # reader workers: `reader_count` per replica (the reference scales it by the replica count, UNet/train.py:63); worker ids
    # are global, so the non-shuffled test readers of different ranks walk DISJOINT strides of the key list -- each global
    # test batch is split over the replicas like experimental_distribute_dataset does (UNet/train.py:85-90) -- and every
    # training worker draws its own random key stream (UNet/imagereader.py:209-233)
    reader_count = max(1, int(reader_count))
    total_workers = reader_count * world
    worker_ids = [rank * reader_count + w for w in range(reader_count)]
    # a reader with the reference's own surface (per-sample generator(), UNet/imagereader.py:338-355) is batched by an adapter
    train_reader = readers.as_batch_reader(train_reader) if train_reader is not None else None
    test_reader = readers.as_batch_reader(test_reader) if test_reader is not None else None
    if train_reader is None:
        train_reader = readers.TileFolderReader(train_lmdb_filepath, number_classes, shuffle=True, seed=0,
                                                balance_classes=bool(balance_classes))
    elif balance_classes and not getattr(train_reader, "balance_classes", False):
        raise ValueError("--balance_classes 1 needs a reader that implements class-balanced sampling (readers.TileFolderReader); "
                         "the reader passed in does not")
    if test_reader is None:
        test_reader = readers.TileFolderReader(test_lmdb_filepath, number_classes, shuffle=False)

    def worker_batches(reader, **kw):
        import inspect
        if "worker" not in inspect.signature(reader.batches).parameters:       # a caller-supplied reader without worker streams
            if total_workers > 1:
                raise ValueError("reader_count x replicas > 1 needs a reader whose batches() takes worker/num_workers")
            return [reader.batches(batch_size, **kw)]
        return [reader.batches(batch_size, worker=w, num_workers=total_workers, **kw) for w in worker_ids]
    say("Test Reader has {} images".format(test_reader.get_image_count()))
    say("Train Reader has {} images".format(train_reader.get_image_count()))
    feeds = []
    try:
        train_reader.startup()
        test_reader.startup()
        # device feed (feed.py): reader batches are staged and copied on a copy stream while the previous step runs, labels
        # travel as uint8 class maps and become the one-hot on the device; UNET_FEED=0 restores the synchronous hand-over
        use_feed = os.environ.get("UNET_FEED", "1") != "0"
        if use_feed:
            from .feed import DeviceFeed
            if use_augmentation and not getattr(train_reader, "augments_itself", False):
                # the reference augments inside its reader processes (UNet/imagereader.py:283-301, settings :79-85), ~31 images/s
                # per host core; here the raw tiles go to the device and the same sequence runs as HIP kernels (augment.py)
                from .augment import AugmentingFeed, DeviceAugmenter
                raw_feed = DeviceFeed(worker_batches(train_reader, classmap=True, pin=False, raw=True), dev_, classmap=True,
                                      number_classes=number_classes, onehot=False)
                train_batches = AugmentingFeed(raw_feed, DeviceAugmenter(
                    rotation_flag=True, reflection_flag=True, jitter_augmentation_severity=0.1, noise_augmentation_severity=0.02,
                    scale_augmentation_severity=0.1, blur_augmentation_max_sigma=2, intensity_augmentation_severity=None,
                    seed=rank, device=dev_), number_classes)
            else:
                train_batches = DeviceFeed(worker_batches(train_reader, classmap=True, pin=False), dev_, classmap=True, number_classes=number_classes)
            test_batches = DeviceFeed(worker_batches(test_reader, classmap=True, pin=False), dev_, classmap=True, number_classes=number_classes)
            feeds = [train_batches, test_batches]
        else:
            train_batches = readers.round_robin(worker_batches(train_reader))
            test_batches = readers.round_robin(worker_batches(test_reader))
        number_channels = train_reader.get_image_size()[2]
        net = (unet_factory or unet_model_module.UNet)(number_classes, global_batch_size, number_channels, learning_rate,
                                                       device=dev_, compute_dtype=compute_dtype)
        strategy = None
        if world > 1 and unet_factory is None:
            from .parallel import DataParallel
            strategy = net.parallel = DataParallel(net.engine)

        train_epoch_size = test_every_n_steps
        test_epoch_size = test_reader.get_image_count() / batch_size
        test_loss = []
        train_loss_metric, train_acc_metric = unet_model_module.Mean("train_loss"), unet_model_module.CategoricalAccuracy("train_accuracy")
        test_loss_metric, test_acc_metric = unet_model_module.Mean("test_loss"), unet_model_module.CategoricalAccuracy("test_accuracy")
        stamp = datetime.datetime.now().strftime("%Y%m%dT%H%M%S")
        writers = None
        if rank == 0:
            writers = (_Scalars(os.path.join(output_folder, "tensorboard-" + stamp, "train")),
                       _Scalars(os.path.join(output_folder, "tensorboard-" + stamp, "test")))

        def check_labels():
            """The reference's reader raises IndexError on the first tile with a class id >= number_classes (UNet/imagereader.py:302-312).
            The device feed only COUNTS such pixels (no host sync per batch), so the counter is polled every LABEL_CHECK_EVERY steps and
            before anything is written; with several replicas the verdict is all-reduced (MAX) so that every rank raises on the same
            step instead of one raising and the others waiting in the next collective."""
            bad = sum(f.out_of_range_labels() for f in feeds if hasattr(f, "out_of_range_labels"))
            if world > 1:
                t = torch.tensor([float(bad)], device=dev_)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                bad = int(t.item())
            if bad > 0:
                raise IndexError("Number of classes specified differs from number of observed classes in data")

        epoch = 0
        while True:
            say("---- Epoch: {} ----".format(epoch))
            if epoch == 0:
                steps_this_epoch = min(1000, train_epoch_size)
                say("Performing Adam Optimizer learning rate warmup for {} steps".format(steps_this_epoch))
                net.set_learning_rate(learning_rate / 10)
            else:
                steps_this_epoch = train_epoch_size
                net.set_learning_rate(learning_rate)
            t0 = time.time()
            step = 0
            while step <= steps_this_epoch:                      # N+1 steps, as the reference's `step > N: break`
                images, labels = next(train_batches)
                net.dist_train_step(strategy, (images.to(dev_, non_blocking=True), labels.to(dev_, non_blocking=True),
                                               train_loss_metric, train_acc_metric))
                say("Train Epoch {}: Batch {}/{}: Loss {} Accuracy = {}".format(
                    epoch, step, train_epoch_size, train_loss_metric.result(), train_acc_metric.result()))
                if writers:
                    writers[0].scalar("loss", train_loss_metric.result(), epoch * train_epoch_size + step)
                    writers[0].scalar("accuracy", train_acc_metric.result(), epoch * train_epoch_size + step)
                train_loss_metric.reset_states(); train_acc_metric.reset_states()
                step += 1
                if step % LABEL_CHECK_EVERY == 0:
                    check_labels()
            epoch_test_loss = []
            step = 0
            while step <= test_epoch_size:
                images, labels = next(test_batches)
                loss_value = net.dist_test_step(strategy, (images.to(dev_, non_blocking=True), labels.to(dev_, non_blocking=True),
                                                           test_loss_metric, test_acc_metric))
                epoch_test_loss.append(loss_value.numpy())
                step += 1
            test_loss.append(np.mean(epoch_test_loss))             # (kept as numpy's fp32 scalar: test_loss.csv then carries the reference's text, UNet/train.py:162,173-176)
            check_labels()                                         # before the csv / checkpoint of this epoch are written
            say("Test Epoch: {}: Loss = {} Accuracy = {}".format(epoch, test_loss_metric.result(), test_acc_metric.result()))
            if writers:
                writers[1].scalar("loss", test_loss_metric.result(), (epoch + 1) * train_epoch_size)
                writers[1].scalar("accuracy", test_acc_metric.result(), (epoch + 1) * train_epoch_size)
            test_loss_metric.reset_states(); test_acc_metric.reset_states()
            if rank == 0:
                with open(os.path.join(output_folder, "test_loss.csv"), "w") as f:
                    f.writelines(str(v) + "\n" for v in test_loss)
            say("Epoch took: {} s".format(time.time() - t0))
            if (len(test_loss) - 1) == int(np.argmin(test_loss)):
                say("Test loss improved: {}, saving checkpoint".format(np.min(test_loss)))
                if strategy is not None:
                    strategy.average_moving_stats()
                if rank == 0:
                    net.save_checkpoint(os.path.join(output_folder, "checkpoint", "ckpt"))
            best = best_epoch_index(test_loss)
            say("Best epoch: {}".format(best))
            if len(test_loss) - best > early_stopping_count:
                break
            epoch += 1
            if max_epochs is not None and epoch >= max_epochs:
                break
        return test_loss
    finally:
        for f in feeds:
            f.close()
        train_reader.shutdown()
        test_reader.shutdown()


def main(argv=None):
    ap = argparse.ArgumentParser(prog="train_unet", description="Script which trains a unet model")
    ap.add_argument("--batch_size", dest="batch_size", type=int, default=4, help="training batch size")
    ap.add_argument("--number_classes", dest="number_classes", type=int, default=2)
    ap.add_argument("--learning_rate", dest="learning_rate", type=float, default=3e-4)
    ap.add_argument("--output_dir", dest="output_folder", type=str, required=True, help="Folder where outputs will be saved (Required)")
    ap.add_argument("--test_every_n_steps", dest="test_every_n_steps", type=int, default=1000)
    ap.add_argument("--balance_classes", dest="balance_classes", type=int, default=0)
    ap.add_argument("--use_augmentation", dest="use_augmentation", type=int, default=1)
    ap.add_argument("--train_database", dest="train_database_filepath", type=str, required=False, default=None,
                    help="training data store (folder of .npy tiles; see readers.py)")
    ap.add_argument("--test_database", dest="test_database_filepath", type=str, required=False, default=None)
    ap.add_argument("--early_stopping", dest="early_stopping_count", type=int, default=10)
    ap.add_argument("--reader_count", dest="reader_count", type=int, default=1)
    # opt-in extras (not in the reference)
    ap.add_argument("--synthetic", type=str, default=None, help="HxWxC[xCOUNT] synthetic tiles instead of databases")
    ap.add_argument("--max_epochs", type=int, default=None)
    ap.add_argument("--compute_dtype", choices=["fp32", "bf16"], default=None,
                    help="bf16: mixed precision (the policy the reference keeps commented out, UNet/train.py:52-54): bf16 "
                         "contractions in the wide 3x3 layers, fp32 accumulation and master weights")
    a = ap.parse_args(argv)
    tr = te = None
    if a.synthetic:
        parts = [int(v) for v in a.synthetic.lower().split("x")]
        h, w, c = parts[:3]
        count = parts[3] if len(parts) > 3 else 64
        tr = readers.SyntheticReader(count, h, w, c, a.number_classes, seed=1)
        te = readers.SyntheticReader(max(count // 4, a.batch_size), h, w, c, a.number_classes, seed=2)
    elif not (a.train_database_filepath and a.test_database_filepath):
        ap.error("--train_database and --test_database are required unless --synthetic is given")
    train_model(a.output_folder, a.batch_size, a.reader_count, a.train_database_filepath, a.test_database_filepath,
                a.use_augmentation, a.number_classes, a.balance_classes, a.learning_rate, a.test_every_n_steps,
                a.early_stopping_count, train_reader=tr, test_reader=te, max_epochs=a.max_epochs, compute_dtype=a.compute_dtype)


if __name__ == "__main__":
    main()
