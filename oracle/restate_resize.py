"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the host-side image path that precedes the matchers.

Restates (a) Pillow's 8-bit LANCZOS resampling, which the reference calls through
``resize_image(image, size, "pil_LANCZOS")`` (src/dataset/utils.py:160-177), and (b) the reference's readers
``read_grayscale`` / ``read_rgb`` after the decode (src/dataset/utils.py:80-160: process_resize :14-28,
pad_bottom_right :30-52, grayscale2tensor / rgb2tensor :55-59, mask2tensor :60-61).

Pillow is a third-party dependency of the reference (imported at src/dataset/utils.py:10, not pinned in
requirements.txt).  Algorithm restated from its published source, src/libImaging/Resample.c: ``precompute_coeffs``
(double coefficients of the filter over ``support * max(scale, 1)``, normalised to sum 1), ``normalize_coeffs_8bpc``
(rounded to 22 fractional bits), ``ImagingResampleHorizontal_8bpc`` then ``ImagingResampleVertical_8bpc``
(``clip8((2^21 + sum src * k) >> 22)`` in int32, the horizontal result rounded to bytes in between; a pass whose
size does not change is skipped).  PINNED: tests/test_images_cpu.py compares this file byte for byte with the installed
Pillow itself (12.2.0 in this image) on seeded frames, and with the reference's own reader functions through
tests/golden/read_image.npz (oracle/make_golden.py::read_image_golden).
"""
import math

import numpy as np

PRECISION_BITS = 22


def lanczos(x):
    def sinc(t):
        return 1.0 if t == 0.0 else math.sin(t * math.pi) / (t * math.pi)
    return sinc(x) * sinc(x / 3) if -3.0 <= x < 3.0 else 0.0


def coeffs(in_size, out_size):
    """precompute_coeffs + normalize_coeffs_8bpc for the box (0, in_size): list of (xmin, int32 taps)."""
    scale = in_size / out_size
    fscale = max(scale, 1.0)
    support = 3.0 * fscale
    out = []
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size)
        w = np.array([lanczos((x - center + 0.5) * (1.0 / fscale)) for x in range(xmin, xmax)], dtype=np.float64)
        ww = 0.0
        for v in w:
            ww += float(v)
        if ww != 0.0:
            w = w / ww
        k = np.where(w < 0, (-0.5 + w * (1 << PRECISION_BITS)), (0.5 + w * (1 << PRECISION_BITS)))
        out.append((xmin, np.trunc(k).astype(np.int32)))
    return out


def _pass(img, out_size, axis):
    """One 8bpc pass along ``axis`` (0 vertical, 1 horizontal) of img [H,W,C] uint8."""
    src = np.moveaxis(img, axis, 0).astype(np.int32)                     # resampled axis first
    dst = np.empty((out_size,) + src.shape[1:], dtype=np.uint8)
    for i, (lo, k) in enumerate(coeffs(src.shape[0], out_size)):
        ss = (1 << (PRECISION_BITS - 1)) + np.tensordot(k, src[lo:lo + len(k)], axes=(0, 0))
        dst[i] = np.clip(ss >> PRECISION_BITS, 0, 255)
    return np.moveaxis(dst, 0, axis)


def pil_resize_lanczos(img, size):
    """PIL.Image.fromarray(img).resize(size, LANCZOS) for uint8 [H,W] / [H,W,3]; size = (w, h)."""
    a = img[:, :, None] if img.ndim == 2 else img
    if a.shape[1] != size[0]:
        a = _pass(a, size[0], 1)
    if a.shape[0] != size[1]:
        a = _pass(a, size[1], 0)
    return a[:, :, 0] if img.ndim == 2 else a


def process_resize(w, h, resize, df=None, resize_no_larger_than=False):
    if resize_no_larger_than and max(h, w) <= max(resize):
        w_new, h_new = w, h
    elif len(resize) == 1 and resize[0] > -1:
        s = resize[0] / max(h, w)
        w_new, h_new = int(round(w * s)), int(round(h * s))
    elif len(resize) == 1:
        w_new, h_new = w, h
    else:
        w_new, h_new = resize
    if df is not None:
        w_new, h_new = int(w_new // df * df), int(h_new // df * df)
    return w_new, h_new


def read_image(img, resize=None, resize_no_larger_than=False, df=None, pad_to=None):
    """The reference's reader after the decode: (tensor [C,h,w] float32, scales [2], original_hw [2], mask or None)."""
    h, w = img.shape[:2]
    w_new, h_new = process_resize(w, h, tuple(resize) if resize is not None else (w, h), df, resize_no_larger_than)
    x = pil_resize_lanczos(img, (w_new, h_new)).astype(np.float32)
    mask = None
    if pad_to is not None:
        p = max(w_new, h_new) if pad_to == -1 else pad_to
        padded = np.zeros((p, p) + x.shape[2:], dtype=np.float32)
        padded[:h_new, :w_new] = x
        mask = np.zeros((p, p), dtype=np.float32)
        mask[:h_new, :w_new] = 1
        x = padded
    t = (x / 255.).astype(np.float32)
    t = t[None] if t.ndim == 2 else np.ascontiguousarray(t.transpose(2, 0, 1))
    return t, np.array([h / h_new, w / w_new], dtype=np.float32), np.array([h, w]), mask
