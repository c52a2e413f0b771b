"""TEST INFRASTRUCTURE ONLY -- loader of ``oracle/jpeg_baseline.c``, the plain-C restatement of the JPEG decode behind the
reference's ``cv2.imread`` calls (src/dataset/utils.py:86-92, 127, 183; see the header of the C file for what it restates and
how it is pinned).  ``__graft_entry__.build()`` compiles it with gcc into ``oracle/_build/``; this module compiles it on
first use when that file is missing (a fresh GPU box).  Only tests/, smoke() and bench.py's cpu_baseline may import this.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(_HERE, "jpeg_baseline.c")
LIB = os.path.join(_HERE, "_build", "libjpegref.so")
_lib = None


class JpegRefError(Exception):
    pass


def build(force: bool = False) -> str:
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        os.makedirs(os.path.dirname(LIB), exist_ok=True)
        subprocess.check_call(["gcc", "-O2", "-std=c99", "-fPIC", "-shared", "-Wall", SRC, "-o", LIB])
    return LIB


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        _lib.jpegref_info.argtypes = [ctypes.c_char_p, ctypes.c_long, ctypes.POINTER(ctypes.c_int)]
        _lib.jpegref_decode.argtypes = [ctypes.c_char_p, ctypes.c_long, ctypes.c_int, ctypes.c_void_p]
    return _lib


def info(buf: bytes) -> dict:
    a = (ctypes.c_int * 12)()
    rc = lib().jpegref_info(buf, len(buf), a)
    if rc == -1:
        raise JpegRefError("not a JPEG / corrupt")
    v = list(a)
    return dict(width=v[0], height=v[1], ncomp=v[2], progressive=bool(v[3]), restart=v[4],
                sampling=[(v[5 + 2 * c], v[6 + 2 * c]) for c in range(v[2])], orientation=v[11], supported=rc == 0)


def decode(buf: bytes, color: bool) -> np.ndarray:
    """The bytes of cv2.imread(IMREAD_GRAYSCALE) ([H,W]: the luma plane) / of IMREAD_COLOR after BGR2RGB ([H,W,3]), before any
    EXIF rotation (``apply_orientation`` does that)."""
    i = info(buf)
    if not i["supported"]:
        raise JpegRefError("unsupported JPEG (progressive / arithmetic / 12-bit / CMYK / sampling)")
    out = np.empty((i["height"], i["width"], 3) if color else (i["height"], i["width"]), dtype=np.uint8)
    rc = lib().jpegref_decode(buf, len(buf), int(color), out.ctypes.data)
    if rc:
        raise JpegRefError(f"decode failed ({rc})")
    return out


def apply_orientation(img: np.ndarray, orientation: int) -> np.ndarray:
    """EXIF orientation as cv2.imread applies it (modules/imgcodecs/src/loadsave.cpp ExifTransform): 1 identity, 2 mirror
    horizontally, 3 rotate 180, 4 mirror vertically, 5 transpose, 6 rotate 90 clockwise, 7 transverse, 8 rotate 270."""
    if orientation == 2:
        img = img[:, ::-1]
    elif orientation == 3:
        img = img[::-1, ::-1]
    elif orientation == 4:
        img = img[::-1]
    elif orientation == 5:
        img = img.swapaxes(0, 1)
    elif orientation == 6:
        img = img.swapaxes(0, 1)[:, ::-1]
    elif orientation == 7:
        img = img.swapaxes(0, 1)[::-1, ::-1]
    elif orientation == 8:
        img = img.swapaxes(0, 1)[::-1]
    return np.ascontiguousarray(img)
