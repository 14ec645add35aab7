"""CPU spec (NumPy) of the `scan_resize != 1` input resize (reference utils/dataset.py:180-181: `image.resize((tile_w, tile_h))`
on an RGB PIL image, default filter).

Test infrastructure only (see oracle/__init__.py).  The arithmetic lives in Pillow (un-vendored dependency of the reference, no
version pinned there; this image carries Pillow 12.2.0, whose default `Image.resize` filter for RGB images is BICUBIC - it has
been since Pillow 7.0).  Restated from Pillow's published algorithm (src/libImaging/Resample.c: precompute_coeffs,
normalize_coeffs_8bpc, ImagingResampleHorizontal_8bpc / Vertical_8bpc):
  * per output index: centre = (i + 0.5) * scale, support = 2 * max(scale, 1), taps [xmin, xmin + n) clipped to the image,
    weights bicubic(a = -0.5)((x + xmin - centre + 0.5) / max(scale, 1)) normalised to sum 1 in float64;
  * weights -> fixed point with 22 fractional bits, rounded half away from zero;
  * horizontal pass first, then vertical, each: (2^21 + sum(pixel * weight)) >> 22 clipped to [0, 255] (uint8 between passes).
PINNED: tests/test_resize_oracle.py compares this bit-for-bit with the installed Pillow on seeded images (CPU suite), and the
GPU tests compare the HIP kernel with both."""
import numpy as np

PRECISION_BITS = 32 - 8 - 2


def _bicubic(x):
    a = -0.5
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def coefficients(in_size, out_size):
    """-> (ksize, bounds (out,2) int [xmin, n], kk (out, ksize) int32 fixed-point weights)."""
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 2.0 * filterscale
    ksize = int(np.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x in range(xmax):
            v = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return ksize, bounds, kk


def _pass(img, out_size, axis):
    img = np.moveaxis(np.asarray(img, np.int64), axis, 0)
    _, bounds, kk = coefficients(img.shape[0], out_size)
    out = np.empty((out_size,) + img.shape[1:], np.int64)
    for i in range(out_size):
        x0, n = bounds[i]
        acc = np.full(img.shape[1:], 1 << (PRECISION_BITS - 1), np.int64)
        for t in range(n):
            acc += img[x0 + t] * int(kk[i, t])
        out[i] = np.clip(acc >> PRECISION_BITS, 0, 255)
    return np.moveaxis(out, 0, axis).astype(np.uint8)


def resize_bicubic_u8(img, out_hw):
    """(H,W,C) uint8 -> (out_h, out_w, C) uint8, Pillow's two-pass BICUBIC (horizontal, then vertical)."""
    img = np.asarray(img, np.uint8)
    if img.shape[1] != out_hw[1]:
        img = _pass(img, out_hw[1], 1)
    if img.shape[0] != out_hw[0]:
        img = _pass(img, out_hw[0], 0)
    return img
