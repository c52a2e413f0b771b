"""Image feeding on the device -- the last part of SURVEY.md 8(f) rank 1.

The reference's readers (src/dataset/utils.py:80-121 ``read_rgb``, :123-160 ``read_grayscale``) decode a file with
cv2, resize it with PIL's LANCZOS filter (``resize_image(..., "pil_LANCZOS")``, :160-177), pad it bottom / right
(``pad_bottom_right``, :30-52) and turn it into a float tensor in [0, 1] (``grayscale2tensor`` / ``rgb2tensor``,
:55-59), all on one host core per image.  Here everything after the decode runs on the GPU: the decoded uint8 frame is
copied once (1 byte per pixel) and ``dfsfm_resample_u8`` returns PIL's bytes exactly (fixed-point arithmetic, see
csrc/image_resize.hip) already converted, padded and masked.  Since r05 the decode itself runs on the GPU too for baseline
JPEG files (``jpeg.decode`` -> csrc/jpeg_decode.hip, bytes identical to libjpeg-turbo, the library behind cv2.imread); other
files keep the reference's host decode (``decode="auto"``).

``read_grayscale`` / ``read_rgb`` keep the reference's signature and return values; ``path`` may also be an already
decoded uint8 array / tensor ([H,W] or [H,W,3] RGB), which is what a maintainer passes after ``cv2.imread``.
"""
import math
import warnings
from functools import lru_cache

import numpy as np
import torch

from . import ops

PRECISION_BITS = 32 - 8 - 2          # Pillow src/libImaging/Resample.c
LANCZOS_SUPPORT = 3.0


def _sinc(x: float) -> float:
    if x == 0.0:
        return 1.0
    x = x * math.pi
    return math.sin(x) / x


def _lanczos(x: float) -> float:
    """lanczos_filter of Resample.c: truncated sinc, a = 3."""
    if -3.0 <= x < 3.0:
        return _sinc(x) * _sinc(x / 3)
    return 0.0


@lru_cache(maxsize=256)
def lanczos_tables(in_size: int, out_size: int):
    """(bounds int32 [out,2], kk int32 [out,ksize]) of PIL for a whole-axis resize in_size -> out_size:
    precompute_coeffs (double arithmetic, libm sin through ``math``) followed by normalize_coeffs_8bpc.
    in_size == out_size gives the identity tables (PIL skips that pass; 1 << 22 reproduces the byte)."""
    if in_size == out_size:
        b = np.stack([np.arange(out_size, dtype=np.int32), np.ones(out_size, dtype=np.int32)], 1)
        return b, np.full((out_size, 1), 1 << PRECISION_BITS, dtype=np.int32)
    scale = float(np.float32(in_size) - np.float32(0.0)) / out_size      # box is a float[4] in C
    filterscale = max(scale, 1.0)
    support = LANCZOS_SUPPORT * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [_lanczos((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x in range(xmax):
            k = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + k * (1 << PRECISION_BITS)) if k < 0 else int(0.5 + k * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


@lru_cache(maxsize=64)
def _device_tables(in_size, out_size, device):
    b, k = lanczos_tables(in_size, out_size)
    return torch.from_numpy(b).to(device), torch.from_numpy(k).to(device)


@lru_cache(maxsize=8)
def _lut255(device):
    return torch.from_numpy(np.arange(256, dtype=np.float32) / 255.).float().to(device)     # image / 255. of utils.py:55-59


def process_resize(w, h, resize, df=None, resize_no_larger_than=False):
    """src/dataset/utils.py:14-28."""
    assert 0 < len(resize) <= 2
    if resize_no_larger_than and (max(h, w) <= max(resize)):
        w_new, h_new = w, h
    elif len(resize) == 1 and resize[0] > -1:
        scale = resize[0] / max(h, w)
        w_new, h_new = int(round(w * scale)), int(round(h * scale))
    elif len(resize) == 1 and resize[0] == -1:
        w_new, h_new = w, h
    else:
        w_new, h_new = resize[0], resize[1]
    if df is not None:
        w_new, h_new = (int(x // df * df) for x in (w_new, h_new))
    return w_new, h_new


def _as_device_u8(image, device):
    if isinstance(image, np.ndarray):
        image = torch.from_numpy(np.array(image, copy=True, order="C") if not image.flags.writeable else np.ascontiguousarray(image))
    if not isinstance(image, torch.Tensor) or image.dtype != torch.uint8 or image.dim() not in (2, 3):
        raise TypeError("a decoded frame is a uint8 [H,W] or [H,W,3] array")
    return image.to(device, non_blocking=True).contiguous()


def resize_lanczos(image_u8, size, device=None):
    """``PIL.Image.fromarray(image).resize(size, LANCZOS)`` (size = (w, h)) on the device, bytes identical."""
    device = torch.device(device if device is not None else (image_u8.device if isinstance(image_u8, torch.Tensor) else "cuda"))
    img = _as_device_u8(image_u8, device)
    bx, kx = _device_tables(img.shape[1], int(size[0]), device)
    by, ky = _device_tables(img.shape[0], int(size[1]), device)
    return ops.resample_u8(img, bx, kx, by, ky, out_u8=True)[0]


_warned_pil_decode = False
_warned_device_fallback = False


def _decode_host(path, color: bool):
    """Host decode, the reference's own: ``cv2.imread(path, IMREAD_GRAYSCALE)`` / ``IMREAD_COLOR`` + BGR->RGB
    (src/dataset/utils.py:86-92, 127).  Without cv2 (this image) Pillow decodes -- the same libjpeg-turbo underneath: for a JPEG
    ``draft('L')`` selects the library's grey output (the luma plane, what OpenCV asks for too) instead of an RGB -> L conversion,
    and the EXIF orientation is applied as cv2.imread does.  Other formats (PNG ...) go through Pillow's own converters and are
    NOT parity-pinned to OpenCV; that case warns once."""
    try:
        import cv2
    except ImportError:
        cv2 = None
    if cv2 is not None:
        if color:
            im = cv2.imread(str(path), cv2.IMREAD_COLOR)
            if im is None:
                raise FileNotFoundError(str(path))
            return np.ascontiguousarray(im[:, :, ::-1])
        im = cv2.imread(str(path), cv2.IMREAD_GRAYSCALE)
        if im is None:
            raise FileNotFoundError(str(path))
        return im
    from PIL import Image, ImageOps
    with Image.open(str(path)) as im:
        if im.format == "JPEG":
            if not color:
                im.draft("L", im.size)
        else:
            global _warned_pil_decode
            if not _warned_pil_decode:
                import warnings
                warnings.warn("detectorfreesfm_amd.images: cv2 is not installed, decoding a non-JPEG file with Pillow -- this "
                              "decode is not parity-pinned to the reference's cv2.imread (pass decoded frames, or install "
                              "OpenCV)", RuntimeWarning)
                _warned_pil_decode = True
        im = ImageOps.exif_transpose(im)                     # cv2.imread honours the EXIF orientation
        return np.asarray(im.convert("RGB" if color else "L"))


def _decode(path, color: bool, device=None, decode: str = "auto"):
    """A file name -> decoded uint8 frame.  ``decode``: "device" = ``jpeg.decode`` (csrc/jpeg_decode.hip: baseline JPEGs, bytes
    identical to libjpeg-turbo / cv2.imread; anything else raises ``jpeg.UnsupportedJpeg``), "host" = the reference's host decode
    (``_decode_host``), "auto" (default) = the device for the files it takes, the host for the rest (progressive JPEGs, PNG ...)."""
    if decode not in ("auto", "device", "host"):
        raise ValueError("decode is 'auto', 'device' or 'host'")
    if decode != "host":
        from . import jpeg
        with open(str(path), "rb") as f:
            buf = f.read()
        if decode == "device" and not jpeg.is_jpeg(buf):
            raise jpeg.UnsupportedJpeg("not a JPEG file")
        if jpeg.is_jpeg(buf):
            from ._lib import DfsfmError
            try:
                return jpeg.decode(buf, color, device if device is not None else "cuda")
            except (jpeg.UnsupportedJpeg, jpeg.CorruptJpeg, DfsfmError) as e:
                # "auto" returns what the reference's reader returns: libjpeg (cv2.imread) takes progressive / multi-scan files and
                # decodes slightly damaged ones with a warning (a wrong restart count, a truncated scan); the device path raises
                if decode == "device":
                    raise
                global _warned_device_fallback
                if not isinstance(e, jpeg.UnsupportedJpeg) and not _warned_device_fallback:
                    warnings.warn(f"{path}: device JPEG decode failed ({type(e).__name__}: {e}); decoding on the host like the "
                                  "reference (further files: silently)", RuntimeWarning)
                    _warned_device_fallback = True
    return _decode_host(path, color)


def _read(image, color, resize, resize_no_larger_than, df, pad_to, ret_scales, ret_pad_mask, device, decode="auto"):
    resize = tuple(resize) if resize is not None else None
    if isinstance(image, (str, bytes)) or hasattr(image, "__fspath__"):
        image = _decode(image, color, device, decode)
    device = torch.device(device if device is not None else (image.device if isinstance(image, torch.Tensor) and image.is_cuda else "cuda"))
    img = _as_device_u8(image, device)
    if (img.dim() == 3) != color:
        raise ValueError("read_rgb takes [H,W,3] frames, read_grayscale [H,W] ones")
    h, w = img.shape[:2]
    w_new, h_new = process_resize(w, h, resize if resize is not None else (w, h), df, resize_no_larger_than)
    scales = torch.tensor([float(h) / float(h_new), float(w) / float(w_new)])
    original_hw = torch.tensor([h, w])
    pad = None
    if pad_to is not None:
        if pad_to == -1:
            pad_to = max(w_new, h_new)
        assert isinstance(pad_to, int) and pad_to >= max(h_new, w_new)          # pad_bottom_right's own assert
        pad = (pad_to, pad_to)
    bx, kx = _device_tables(w, w_new, device)
    by, ky = _device_tables(h, h_new, device)
    _, ts_image, mask = ops.resample_u8(img, bx, kx, by, ky, lut=_lut255(device), pad_hw=pad,
                                        want_mask=bool(ret_pad_mask and pad_to))
    ret_val = [ts_image]
    if ret_scales:
        ret_val += [scales, original_hw]
    if ret_pad_mask:
        ret_val.append(mask if pad_to else None)
    return ret_val[0] if len(ret_val) == 1 else ret_val


def read_grayscale(path, resize=None, resize_no_larger_than=False, resize_float=False, df=None, client=None, pad_to=None,
                   ret_scales=False, ret_pad_mask=False, augmentor=None, device=None, decode="auto"):
    """src/dataset/utils.py:123-160 with the frame resized / padded / converted on the GPU: returns ts_image [1,h,w]
    fp32 on the device (+ scales, original_hw on the host, + the padding mask on the device) like the reference."""
    if client is not None or augmentor is not None:
        raise NotImplementedError("petrel clients and augmentors are training-time options of the reference")
    return _read(path, False, resize, resize_no_larger_than, df, pad_to, ret_scales, ret_pad_mask, device, decode)


def read_rgb(path, resize=None, resize_no_larger_than=False, resize_float=False, df=None, client=None, pad_to=None,
             ret_scales=False, ret_pad_mask=False, augmentor=None, device=None, decode="auto"):
    """src/dataset/utils.py:80-121 likewise: ts_image [3,h,w] fp32 on the device."""
    if client is not None or augmentor is not None:
        raise NotImplementedError("petrel clients and augmentors are training-time options of the reference")
    return _read(path, True, resize, resize_no_larger_than, df, pad_to, ret_scales, ret_pad_mask, device, decode)
