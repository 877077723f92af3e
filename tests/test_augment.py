"""Augmentation stage.  The golden file was produced by RUNNING THE REFERENCE's UNet/augment.py
(tests/golden/make_augment_golden.py), so this is the one stage whose parity is pinned to the reference itself:
  * not-gpu: oracle/augment_numpy.py reproduces every golden case (images to 1 ulp-level, masks exactly);
  * gpu:     the HIP kernels (package augment.DeviceAugmenter), fed the same random draws, reproduce the reference's outputs.
"""
import os

import numpy as np
import pytest
import torch

from conftest import pkg
from oracle import augment_numpy as A

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "augment_ref.npz"))
NAMES = [str(n) for n in GOLD["names"]]


def _case(name):
    img, mask = GOLD[name + "/img"], GOLD[name + "/mask"]
    kw = {k.split("/kw/")[1]: float(GOLD[k]) for k in GOLD.files if k.startswith(name + "/kw/")}
    for k in ("rotation_flag", "reflection_flag"):
        if k in kw:
            kw[k] = bool(kw[k])
    np.random.seed(int(GOLD[name + "/seed"]))              # the reference draws from numpy's legacy global RNG
    h, w, c = img.shape
    return img, mask, A.draw(h, w, c, **kw), GOLD[name + "/out_img"], GOLD[name + "/out_mask"], kw


@pytest.mark.parametrize("name", NAMES)
def test_oracle_reproduces_reference_outputs(name):
    img, mask, p, ref_img, ref_mask, _ = _case(name)
    out_img, out_mask = A.augment(img, mask, p)
    assert out_img.dtype == np.float32 and out_img.shape == ref_img.shape
    assert np.abs(out_img - ref_img).max() <= 5e-7 * np.abs(ref_img).max()
    assert np.array_equal(out_mask, ref_mask)


def test_oracle_boundary_rules():
    # skimage 'reflect' does not repeat the edge sample, scipy 'reflect' does
    assert A._mirror(np.array([-2, -1, 0, 4, 5, 6, 9]), 5).tolist() == [2, 1, 0, 4, 3, 2, 1]
    assert A._sym(np.array([-2, -1, 0, 4, 5, 6]), 5).tolist() == [1, 0, 0, 4, 4, 3]
    k = A.gaussian_kernel1d(1.3)
    assert len(k) == 2 * int(4 * 1.3 + 0.5) + 1 and abs(k.sum() - 1) < 1e-15


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_device_augmenter_reproduces_reference_outputs(name):
    img, mask, p, ref_img, ref_mask, kw = _case(name)
    aug = pkg("augment").DeviceAugmenter(**kw)
    x = torch.as_tensor(img[None]).cuda(); m = torch.as_tensor(mask[None].astype(np.float32)).cuda()
    out_img, out_mask = aug(x, m, params=[p])
    oi, om = out_img[0].cpu().numpy(), out_mask[0].cpu().numpy()
    assert np.abs(oi - ref_img).max() <= 2e-6 * np.abs(ref_img).max()
    assert (om != ref_mask).mean() <= 1e-3                 # an interpolated value within 1 ulp of .5 may round the other way


@pytest.mark.gpu
def test_device_augmenter_batches_and_is_deterministic():
    # a batch with different per-image parameters == the oracle image by image; same seed -> same result
    rng = np.random.RandomState(0)
    imgs = (rng.rand(3, 40, 56, 1) * 1000).astype(np.float32)
    masks = (rng.rand(3, 40, 56) > 0.6).astype(np.float32)
    kw = dict(rotation_flag=True, reflection_flag=True, jitter_augmentation_severity=0.1, noise_augmentation_severity=0.02,
              scale_augmentation_severity=0.1, blur_augmentation_max_sigma=2, intensity_augmentation_severity=0.05)
    np.random.seed(11)
    ps = [A.draw(40, 56, 1, **kw) for _ in range(3)]
    aug = pkg("augment").DeviceAugmenter(**kw)
    oi, om = aug(torch.as_tensor(imgs).cuda(), torch.as_tensor(masks).cuda(), params=ps)
    for i in range(3):
        ri, rm = A.augment(imgs[i], masks[i], ps[i])
        assert np.abs(oi[i].cpu().numpy() - ri).max() <= 2e-6 * np.abs(ri).max()
        assert (om[i].cpu().numpy() != rm).mean() <= 1e-3
    a1 = pkg("augment").DeviceAugmenter(seed=3, **kw); a2 = pkg("augment").DeviceAugmenter(seed=3, **kw)
    x = torch.as_tensor(imgs).cuda(); m = torch.as_tensor(masks).cuda()
    r1, r2 = a1(x, m), a2(x, m)
    assert torch.equal(r1[0], r2[0]) and torch.equal(r1[1], r2[1]) and r1[0].shape == x.shape
    assert set(np.unique(r1[1].cpu().numpy()).tolist()) <= {0.0, 1.0}


@pytest.mark.gpu
def test_zscore_kernel_and_augmenting_feed(tmp_path):
    # z-score + NHWC->NCHW kernel == the reader's zscore_normalize; the whole device pipeline (raw tiles -> augment -> z-score ->
    # one-hot) yields what the train step takes and equals the oracle when fed the same draws
    import ctypes
    readers, feed, aug = pkg("readers"), pkg("feed"), pkg("augment")
    L = pkg("_lib").lib()
    rng = np.random.RandomState(2)
    x = (rng.rand(2, 24, 32, 3) * np.array([0.5, 300.0, 4000.0])).astype(np.float32)       # channel 0 has std <= 1
    xd = torch.as_tensor(x).cuda(); out = torch.empty(2, 3, 24, 32, device="cuda")
    nb = L.unet_zscore_workspace(2, 3); ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    L.unet_zscore_nhwc_to_nchw(ctypes.c_void_p(xd.data_ptr()), ctypes.c_void_p(out.data_ptr()), 2, 24, 32, 3,
                               ctypes.c_void_p(ws.data_ptr()), nb, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    ref = np.stack([readers.zscore_normalize(x[i].transpose(2, 0, 1)) for i in range(2)])
    assert np.abs(out.cpu().numpy() - ref).max() < 2e-5

    for k in range(4):
        np.save(tmp_path / ("t%d.npy" % k), (rng.rand(32, 32) * 4000).astype(np.float32))
        np.save(tmp_path / ("t%d_mask.npy" % k), (rng.rand(32, 32) > 0.5).astype(np.uint8))
    rd = readers.TileFolderReader(str(tmp_path), 2)
    kw = dict(rotation_flag=True, reflection_flag=True, jitter_augmentation_severity=0.1, scale_augmentation_severity=0.1)
    a = aug.DeviceAugmenter(seed=5, **kw)
    pipe = aug.AugmentingFeed(feed.DeviceFeed(rd.batches(2, classmap=True, pin=False, raw=True), "cuda:0", classmap=True,
                                              number_classes=2, onehot=False), a, 2)
    img, lab = next(pipe)
    assert tuple(img.shape) == (2, 1, 32, 32) and lab.dtype == torch.int32 and tuple(lab.shape) == (2, 32, 32, 2)
    assert torch.equal(lab.sum(-1), torch.ones(2, 32, 32, dtype=torch.int32, device=lab.device))
    # replay: same seed -> same draws; oracle on the raw tiles, then the reader's z-score
    rs = np.random.RandomState(5)
    raw = [np.load(tmp_path / ("t%d.npy" % k)) for k in range(2)]; msk = [np.load(tmp_path / ("t%d_mask.npy" % k)) for k in range(2)]
    for i in range(2):
        p = A.draw(32, 32, 1, rand=rs.rand, randn=rs.randn, **kw)
        oi, om = A.augment(raw[i][:, :, None], msk[i], p)
        ref_i = readers.zscore_normalize(oi.transpose(2, 0, 1))
        assert np.abs(img[i].cpu().numpy() - ref_i).max() < 1e-4
        assert (lab[i].argmax(-1).cpu().numpy() != om).mean() <= 1e-3
    pipe.close()
