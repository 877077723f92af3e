"""CPU restatement of the reference's augmentation stage -- TEST INFRASTRUCTURE ONLY (never imported by the product path).

Follows UNet/augment.py of the reference:
  augment_image            UNet/augment.py:19-157   (order of the random draws :64-106, noise :113-122, blur :124-136,
                                                      intensity :138-150, mask rounding :152-155)
  apply_affine_transformation UNet/augment.py:160-174 (rotate, then scale/translate warp, then flips)
and the third-party routines it calls, restated from their published behaviour and PINNED against the reference run in this
container (tests/golden/augment_ref.npz, made by tests/golden/make_augment_golden.py with scikit-image 0.18.3 / scipy 1.7.1):
  skimage.transform.rotate / warp, order 1, mode='reflect', float32 images: output pixel (r, c) samples the input at
      (x, y) = M . (c, r, 1) computed in float32, bilinear weights in float32, out-of-range taps mirrored WITHOUT repeating the
      edge sample (skimage's 'reflect' = numpy.pad 'reflect');
  scipy.ndimage.gaussian_filter(img, sigma, mode='reflect') on the [H,W,C] array: separable, radius int(4*sigma + 0.5),
      along ALL THREE axes (the channel axis too -- a quirk of the reference for multi-channel images), boundary d c b a | a b c d
      (scipy's 'reflect' repeats the edge sample).

Parity status: PINNED for this module (unlike the network oracle): every case of the golden file is reproduced, see
tests/test_augment.py for the tolerances.
"""
import numpy as np

F = np.float32


def draw(h, w, c, rotation_flag=False, reflection_flag=False, jitter_augmentation_severity=0, noise_augmentation_severity=0,
         scale_augmentation_severity=0, blur_augmentation_max_sigma=0, intensity_augmentation_severity=0, rand=None, randn=None):
    """The reference's random draws, in its order, from numpy's legacy global RNG (or the supplied callables).  Returns the
    geometry parameters and the raw uniforms / normal field of the photometric stages (their scales depend on the image)."""
    rand = rand or np.random.rand
    randn = randn or np.random.randn
    p = dict(orientation=None, reflect_x=False, reflect_y=False, jitter_x=0, jitter_y=0, scale_x=1.0, scale_y=1.0,
             noise_severity=float(noise_augmentation_severity or 0), blur_max_sigma=float(blur_augmentation_max_sigma or 0),
             intensity_severity=float(intensity_augmentation_severity or 0))
    if rotation_flag:
        p["orientation"] = 360 * rand()
    if reflection_flag:
        p["reflect_x"] = bool(rand() > 0.5)
        p["reflect_y"] = bool(rand() > 0.5)
    js = jitter_augmentation_severity or 0
    if js > 0:
        jx = int(js * (w * rand()))
        if rand() > 0.5:
            jx = -jx
        jy = int(js * (h * rand()))
        if rand() > 0.5:
            jy = -jy
        p["jitter_x"], p["jitter_y"] = jx, jy
    ss = scale_augmentation_severity or 0
    if ss > 0:
        p["scale_x"] = (1 - ss) + 2 * ss * rand()
        p["scale_y"] = (1 - ss) + 2 * ss * rand()
    if p["noise_severity"] > 0:
        p["noise_u"] = rand()
        p["noise_field"] = randn(h, w, c)
    if p["blur_max_sigma"] > 0:
        p["blur_u"] = rand()
    if p["intensity_severity"] > 0:
        p["intensity_u"] = rand()
        p["intensity_sign"] = 1.0 if rand() > 0.5 else -1.0
    return p


def rotation_matrix(h, w, angle_deg):
    """Output->input map of skimage.transform.rotate(resize=False): T(center) R(angle) T(-center), center = (cols/2-.5, rows/2-.5)."""
    cx, cy = w / 2.0 - 0.5, h / 2.0 - 0.5
    a = np.deg2rad(angle_deg)
    t1 = np.array([[1, 0, cx], [0, 1, cy], [0, 0, 1.0]])
    r = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1.0]])
    t3 = np.array([[1, 0, -cx], [0, 1, -cy], [0, 0, 1.0]])
    m = t1 @ r @ t3
    m[2] = (0, 0, 1)
    return m


def affine_inverse_matrix(jitter_x, jitter_y, scale_x, scale_y):
    """skimage AffineTransform(translation, scale)._inv_matrix, which the reference passes to warp as the output->input map."""
    m = np.array([[scale_x, 0.0, jitter_x], [0.0, scale_y, jitter_y], [0.0, 0.0, 1.0]])
    return np.linalg.inv(m)


def _mirror(idx, n):
    """skimage 'reflect': period 2(n-1), no repeated edge sample."""
    if n == 1:
        return np.zeros_like(idx)
    cmax = n - 1
    a = np.abs(idx)
    q, r = a // cmax, a % cmax
    return np.where(q % 2 != 0, cmax - r, r)


def warp_bilinear_reflect(img2d, m):
    """One channel, float32 throughout, as skimage's _warp_fast: c = m00*tfc + m01*tfr + m02, r = m10*tfc + m11*tfr + m12."""
    img2d = np.asarray(img2d, dtype=F)
    h, w = img2d.shape
    m = np.asarray(m, dtype=F)
    tfr, tfc = np.meshgrid(np.arange(h, dtype=F), np.arange(w, dtype=F), indexing="ij")
    c = (m[0, 0] * tfc + m[0, 1] * tfr) + m[0, 2]
    r = (m[1, 0] * tfc + m[1, 1] * tfr) + m[1, 2]
    minr, minc = np.floor(r), np.floor(c)
    maxr, maxc = np.ceil(r), np.ceil(c)
    dr, dc = (r - minr).astype(F), (c - minc).astype(F)
    i0, i1 = _mirror(minr.astype(np.int64), h), _mirror(maxr.astype(np.int64), h)
    j0, j1 = _mirror(minc.astype(np.int64), w), _mirror(maxc.astype(np.int64), w)
    tl, tr, bl, br = img2d[i0, j0], img2d[i0, j1], img2d[i1, j0], img2d[i1, j1]
    one = F(1)
    top = (one - dc) * tl + dc * tr
    bot = (one - dc) * bl + dc * br
    return ((one - dr) * top + dr * bot).astype(F)


def apply_affine(I, p):
    """UNet/augment.py:160-174 on an [H,W] or [H,W,C] float32 array."""
    I = np.asarray(I, dtype=F)
    chans = [I] if I.ndim == 2 else [I[..., k] for k in range(I.shape[2])]
    h, w = chans[0].shape
    if p["orientation"] is not None:
        mr = rotation_matrix(h, w, p["orientation"])
        chans = [warp_bilinear_reflect(ch, mr) for ch in chans]
    ma = affine_inverse_matrix(p["jitter_x"], p["jitter_y"], p["scale_x"], p["scale_y"])
    chans = [warp_bilinear_reflect(ch, ma) for ch in chans]
    out = chans[0] if I.ndim == 2 else np.dstack(chans)
    if p["reflect_x"]:
        out = np.fliplr(out)
    if p["reflect_y"]:
        out = np.flipud(out)
    return out


def gaussian_kernel1d(sigma):
    radius = int(4.0 * float(sigma) + 0.5)
    x = np.arange(-radius, radius + 1)
    k = np.exp(-0.5 / (sigma * sigma) * x ** 2)
    return k / k.sum()


def _sym(idx, n):
    """scipy.ndimage 'reflect': d c b a | a b c d | d c b a (edge sample repeated), period 2n."""
    idx = np.mod(idx, 2 * n)
    return np.where(idx >= n, 2 * n - 1 - idx, idx)


def gaussian_filter_reflect(img, sigma):
    """scipy.ndimage.gaussian_filter(img, sigma, mode='reflect') on every axis of `img` (float32 in, float32 out, double sums)."""
    out = np.asarray(img, dtype=F)
    k = gaussian_kernel1d(sigma)
    radius = (len(k) - 1) // 2
    for axis in range(out.ndim):
        n = out.shape[axis]
        src = np.moveaxis(out, axis, 0).astype(np.float64)
        acc = np.zeros_like(src)
        base = np.arange(n)
        for t in range(-radius, radius + 1):
            acc += k[t + radius] * src[_sym(base + t, n)]
        out = np.moveaxis(acc.astype(F), 0, axis)
    return out


def augment(img, mask, p):
    """UNet/augment.py:19-157 with the draws of `draw` -> (img float32 [H,W,C], mask float32 [H,W] rounded)."""
    img = apply_affine(np.asarray(img, dtype=F), p)
    if mask is not None:
        mask = apply_affine(np.asarray(mask, dtype=F), p)
    if p["noise_severity"] > 0:
        sigma_max = p["noise_severity"] * (np.max(img) - np.min(img))
        sigma = -sigma_max + 2 * sigma_max * p["noise_u"]
        img = img + p["noise_field"] * sigma                      # float64 from here on, like the reference
    if p["blur_max_sigma"] > 0:
        sigma = -p["blur_max_sigma"] + 2 * p["blur_max_sigma"] * p["blur_u"]
        if sigma > 0:
            k = gaussian_kernel1d(sigma)
            radius = (len(k) - 1) // 2
            out = np.asarray(img)
            for axis in range(out.ndim):                          # same dtype in and out per pass, double sums
                n = out.shape[axis]
                src = np.moveaxis(out, axis, 0).astype(np.float64)
                acc = np.zeros_like(src)
                base = np.arange(n)
                for t in range(-radius, radius + 1):
                    acc += k[t + radius] * src[_sym(base + t, n)]
                out = np.moveaxis(acc.astype(out.dtype), 0, axis)
            img = out
    if p["intensity_severity"] > 0:
        rng_ = np.max(img) - np.min(img)
        img = img + p["intensity_sign"] * (p["intensity_u"] * p["intensity_severity"] * rng_)
    img = np.asarray(img, dtype=F)
    if mask is not None:
        mask = np.round(np.asarray(mask, dtype=F))
    return img, mask
