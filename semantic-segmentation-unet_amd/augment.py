"""Device-side augmentation: the reference's `augment.augment_image` (UNet/augment.py:19-157) batched on the GPU.

Same arguments and the same sequence of operations as the reference -- rotate (bilinear, mirror boundary), scale/translate
warp, flips, additive gaussian noise scaled by the image range, gaussian blur, additive intensity shift, mask rounded to
integers -- executed by the HIP kernels of csrc/augment.hip on [N,H,W,C] fp32 batches resident in HBM.  The per-image random
parameters are drawn on the host in the reference's order (:64-106; a private numpy RandomState instead of the global one)
and shipped as a few small arrays; the noise field comes from the device RNG.  `params=` injects explicit draws (as produced
by the oracle's `draw`) so that tests can pin the kernels to the reference's outputs.
"""
import ctypes

import numpy as np
import torch

from . import _lib


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _rotation_matrix(h, w, angle_deg):
    cx, cy = w / 2.0 - 0.5, h / 2.0 - 0.5                      # skimage.transform.rotate: centre (cols/2 - .5, rows/2 - .5)
    a = np.deg2rad(angle_deg)
    t1 = np.array([[1, 0, cx], [0, 1, cy], [0, 0, 1.0]])
    r = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1.0]])
    t3 = np.array([[1, 0, -cx], [0, 1, -cy], [0, 0, 1.0]])
    return (t1 @ r @ t3)[:2].reshape(6)


def _affine_inverse(jx, jy, sx, sy):                            # AffineTransform(translation, scale)._inv_matrix
    return np.linalg.inv(np.array([[sx, 0.0, jx], [0.0, sy, jy], [0.0, 0.0, 1.0]]))[:2].reshape(6)


class DeviceAugmenter:
    def __init__(self, rotation_flag=False, reflection_flag=False, jitter_augmentation_severity=0,
                 noise_augmentation_severity=0, scale_augmentation_severity=0, blur_augmentation_max_sigma=0,
                 intensity_augmentation_severity=0, seed=0, device="cuda"):
        for v in (jitter_augmentation_severity, noise_augmentation_severity, scale_augmentation_severity,
                  intensity_augmentation_severity):
            assert 0 <= (v or 0) < 1                            # UNet/augment.py:49-52
        self.kw = dict(rotation_flag=bool(rotation_flag), reflection_flag=bool(reflection_flag),
                       jitter=float(jitter_augmentation_severity or 0), noise=float(noise_augmentation_severity or 0),
                       scale=float(scale_augmentation_severity or 0), blur=float(blur_augmentation_max_sigma or 0),
                       intensity=float(intensity_augmentation_severity or 0))
        self.rs = np.random.RandomState(seed)
        self.dev = torch.device(device)
        self.gen = torch.Generator(device=self.dev); self.gen.manual_seed(seed)
        self.L = _lib.lib()                                     # fails loudly without the HIP library

    def _draw(self, h, w):
        """One image's draws in the reference's order (UNet/augment.py:64-150)."""
        k, rand = self.kw, self.rs.rand
        p = dict(orientation=None, reflect_x=False, reflect_y=False, jitter_x=0, jitter_y=0, scale_x=1.0, scale_y=1.0)
        if k["rotation_flag"]:
            p["orientation"] = 360 * rand()
        if k["reflection_flag"]:
            p["reflect_x"] = bool(rand() > 0.5); p["reflect_y"] = bool(rand() > 0.5)
        if k["jitter"] > 0:
            jx = int(k["jitter"] * (w * rand())); jx = -jx if rand() > 0.5 else jx
            jy = int(k["jitter"] * (h * rand())); jy = -jy if rand() > 0.5 else jy
            p["jitter_x"], p["jitter_y"] = jx, jy
        if k["scale"] > 0:
            p["scale_x"] = (1 - k["scale"]) + 2 * k["scale"] * rand()
            p["scale_y"] = (1 - k["scale"]) + 2 * k["scale"] * rand()
        if k["noise"] > 0:
            p["noise_u"] = rand()
        if k["blur"] > 0:
            p["blur_u"] = rand()
        if k["intensity"] > 0:
            p["intensity_u"] = rand(); p["intensity_sign"] = 1.0 if rand() > 0.5 else -1.0
        return p

    def __call__(self, images, masks=None, params=None):
        """images [N,H,W,C] fp32 on the device, masks [N,H,W] (any real dtype) or None -> (images', masks' fp32 rounded)."""
        L, dev = self.L, self.dev
        st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        x = images.to(dev, torch.float32).contiguous()
        n, h, w, c = x.shape
        ps = params if params is not None else [self._draw(h, w) for _ in range(n)]
        assert len(ps) == n
        k = self.kw
        f32 = lambda a: torch.as_tensor(np.asarray(a, dtype=np.float32), device=dev)
        rot = all(p["orientation"] is not None for p in ps)
        assert rot or all(p["orientation"] is None for p in ps)
        aff = f32([_affine_inverse(p["jitter_x"], p["jitter_y"], p["scale_x"], p["scale_y"]) for p in ps])
        flips = torch.as_tensor(np.asarray([int(p["reflect_x"]) | (int(p["reflect_y"]) << 1) for p in ps], dtype=np.int32), device=dev)
        rmat = f32([_rotation_matrix(h, w, p["orientation"]) for p in ps]) if rot else None

        def geometry(t, cc, do_round):
            a = t
            if rot:
                b = torch.empty_like(a)
                L.unet_augment_warp(_p(a), _p(b), n, h, w, cc, _p(rmat), None, 0, st)
                a = b
            b = torch.empty_like(a)
            L.unet_augment_warp(_p(a), _p(b), n, h, w, cc, _p(aff), _p(flips), do_round, st)
            return b

        x = geometry(x, c, 0)
        m = None
        if masks is not None:
            m = geometry(masks.to(dev, torch.float32).contiguous().view(n, h, w, 1), 1, 1).view(n, h, w)   # :108-111,152-155
        per = h * w * c
        need_noise = any("noise_u" in p for p in ps)
        need_blur = any("blur_u" in p for p in ps)
        need_int = any("intensity_u" in p for p in ps)
        if need_noise or need_int:
            mm = torch.empty(n, 2, dtype=torch.float32, device=dev)
            nb = L.unet_augment_minmax_workspace(n)
            ws = torch.empty(nb, dtype=torch.uint8, device=dev)
        if need_noise:
            cn = f32([p.get("noise_severity", k["noise"]) * (2 * p["noise_u"] - 1) if "noise_u" in p else 0.0 for p in ps])
            if params is not None and "noise_field" in ps[0]:
                field = f32(np.stack([p["noise_field"] for p in ps]))
            else:
                field = torch.randn(n, h, w, c, device=dev, generator=self.gen)
            L.unet_augment_minmax(_p(x), n, per, _p(mm), _p(ws), nb, st)
            L.unet_augment_noise_intensity(_p(x), _p(field), n, per, _p(mm), _p(cn), None, st)
        if need_blur:
            sg = f32([max(0.0, p.get("blur_max_sigma", k["blur"]) * (2 * p["blur_u"] - 1)) if "blur_u" in p else 0.0 for p in ps])
            tmp = torch.empty_like(x)
            L.unet_augment_gaussian_blur(_p(x), _p(tmp), n, h, w, c, _p(sg), st)
        if need_int:
            ca = f32([p["intensity_sign"] * p["intensity_u"] * p.get("intensity_severity", k["intensity"]) if "intensity_u" in p else 0.0 for p in ps])
            L.unet_augment_minmax(_p(x), n, per, _p(mm), _p(ws), nb, st)
            L.unet_augment_noise_intensity(_p(x), None, n, per, _p(mm), None, _p(ca), st)
        return x, m


class AugmentingFeed:
    """reader (RAW images [B,C,H,W] fp32 + uint8 class maps [B,H,W]) -> feed.DeviceFeed -> DeviceAugmenter -> z-score ->
    one-hot: the reference's reader order (augment, then zscore_normalize, then one-hot: UNet/imagereader.py:283-312), all on
    the device.  Yields what the train step takes: images [B,C,H,W] fp32 z-scored, labels int32 one-hot [B,H,W,K]."""

    def __init__(self, feed, augmenter, number_classes):
        self.feed, self.aug, self.k = feed, augmenter, number_classes
        self.L = augmenter.L

    def __iter__(self):
        return self

    def __next__(self):
        img, cls = next(self.feed)                                   # [B,C,H,W] fp32, [B,H,W] uint8 on the device
        dev = img.device
        st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        b, c, h, w = img.shape
        x = img.permute(0, 2, 3, 1).contiguous() if c > 1 else img.reshape(b, h, w, 1)
        x, m = self.aug(x, cls)
        out = torch.empty(b, c, h, w, dtype=torch.float32, device=dev)
        nb = self.L.unet_zscore_workspace(b, c)
        ws = torch.empty(nb, dtype=torch.uint8, device=dev)
        self.L.unet_zscore_nhwc_to_nchw(_p(x), _p(out), b, h, w, c, _p(ws), nb, st)
        cm = m.to(torch.uint8)
        onehot = torch.empty(b, h, w, self.k, dtype=torch.int32, device=dev)
        # class ids >= number_classes are counted on the device (the reference's reader raises IndexError for them,
        # UNet/imagereader.py:302-312); train.py reads the counter at its per-epoch sync point
        bad = getattr(self.feed, "_bad", None)
        self.L.unet_labels_onehot(_p(cm), _p(onehot), b * h * w, self.k, _p(bad) if bad is not None else None, st)
        return out, onehot

    def out_of_range_labels(self):
        return self.feed.out_of_range_labels() if hasattr(self.feed, "out_of_range_labels") else 0

    def close(self):
        self.feed.close()
