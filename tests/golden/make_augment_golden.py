"""Golden vectors for the augmentation stage, produced BY THE REFERENCE ITSELF.

Run with the interpreter that has scikit-image / scipy of the reference's era (here: /opt/conda/bin/python3.9, scikit-image
0.18.3, scipy 1.7.1):

    /opt/conda/bin/python3.9 tests/golden/make_augment_golden.py /root/reference/UNet tests/golden/augment_ref.npz

It imports the reference's UNet/augment.py unchanged, seeds numpy's legacy global RNG (the reference draws all its random
parameters from it, UNet/augment.py:64-150) and records inputs, the seed and the keyword arguments of every case next to the
reference's outputs.  Nothing of the reference's source is stored - only data.  tests/test_augment.py replays the same
draws and checks oracle/augment_numpy.py (and, on the GPU, the HIP kernels) against these outputs.
"""
import sys
import numpy as np

ref_dir, out_path = sys.argv[1], sys.argv[2]
sys.path.insert(0, ref_dir)
import augment                      # the reference module (UNet/augment.py)

CASES = [
    # name, H, W, C, seed, kwargs of augment_image
    ("identity", 24, 32, 1, 1, {}),
    ("reflect_only", 24, 32, 1, 2, dict(reflection_flag=True)),
    ("rotate_only", 32, 32, 1, 3, dict(rotation_flag=True)),
    ("jitter_scale", 32, 48, 1, 4, dict(jitter_augmentation_severity=0.1, scale_augmentation_severity=0.1)),
    ("noise_blur_intensity", 32, 32, 1, 5, dict(noise_augmentation_severity=0.02, blur_augmentation_max_sigma=2, intensity_augmentation_severity=0.05)),
    ("defaults_1ch", 64, 64, 1, 6, dict(rotation_flag=True, reflection_flag=True, jitter_augmentation_severity=0.1,
                                         noise_augmentation_severity=0.02, scale_augmentation_severity=0.1, blur_augmentation_max_sigma=2)),
    ("defaults_1ch_b", 64, 48, 1, 7, dict(rotation_flag=True, reflection_flag=True, jitter_augmentation_severity=0.1,
                                           noise_augmentation_severity=0.02, scale_augmentation_severity=0.1, blur_augmentation_max_sigma=2)),
    ("defaults_3ch", 48, 64, 3, 8, dict(rotation_flag=True, reflection_flag=True, jitter_augmentation_severity=0.1,
                                         noise_augmentation_severity=0.02, scale_augmentation_severity=0.1, blur_augmentation_max_sigma=2)),
    ("defaults_1ch_c", 256, 256, 1, 9, dict(rotation_flag=True, reflection_flag=True, jitter_augmentation_severity=0.1,
                                             noise_augmentation_severity=0.02, scale_augmentation_severity=0.1, blur_augmentation_max_sigma=2)),
]

out = {}
rng = np.random.RandomState(1234)
for name, h, w, c, seed, kw in CASES:
    img = (rng.rand(h, w, c) * 4000).astype(np.float32)
    # piecewise-constant class map like the reference's masks (two classes, blobs)
    yy, xx = np.mgrid[0:h, 0:w]
    mask = (((yy // 5 + xx // 7) % 3) == 0).astype(np.uint8)
    np.random.seed(seed)
    aimg, amask = augment.augment_image(img, mask, **kw)
    out[name + "/img"] = img; out[name + "/mask"] = mask
    out[name + "/out_img"] = np.asarray(aimg); out[name + "/out_mask"] = np.asarray(amask)
    out[name + "/seed"] = np.int64(seed)
    for k, v in kw.items():
        out[name + "/kw/" + k] = np.float64(v)
out["names"] = np.array([c[0] for c in CASES])
np.savez_compressed(out_path, **out)
print("wrote", out_path, "with", len(CASES), "cases")
