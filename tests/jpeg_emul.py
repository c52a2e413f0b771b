"""Loader of the CPU lane model of the device JPEG decoder (tests/jpeg_emul.cpp: the thread functions of
csrc/jpeg_core.h compiled with g++, kernels as loops).  Test infrastructure."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
SRC = os.path.join(_HERE, "jpeg_emul.cpp")
LIB = os.path.join(_HERE, "_build", "libjpeg_emul.so")
_DEPS = [SRC, os.path.join(_ROOT, "detectorfreesfm_amd", "csrc", "jpeg_core.h"),
         os.path.join(_ROOT, "detectorfreesfm_amd", "csrc", "jpeg_host.h"), os.path.join(_ROOT, "include", "dfsfm_hip.h")]
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB) or any(os.path.getmtime(LIB) < os.path.getmtime(d) for d in _DEPS):
            os.makedirs(os.path.dirname(LIB), exist_ok=True)
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-I" + os.path.join(_ROOT, "include"),
                                   "-I" + os.path.join(_ROOT, "detectorfreesfm_amd", "csrc"), SRC, "-o", LIB])
        _lib = ctypes.CDLL(LIB)
        _lib.jd_emul_workspace.restype = ctypes.c_size_t
        _lib.jd_emul_workspace.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int]
        _lib.jd_emul_decode.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p] + [ctypes.c_void_p] * 7 + \
            [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
             ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
    return _lib


def decode(pl, color: bool, sweeps: int = 64, order: int = 0, max_calls: int = 64):
    """Runs the lane model on a ``detectorfreesfm_amd.jpeg.Plan``; returns (image, dict(status, sweeps used, calls))."""
    L = lib()
    ch = 3 if color else 1
    nbytes = L.jd_emul_workspace(ctypes.byref(pl.frame), pl.scan.size, ch)
    assert nbytes > 0
    ws = np.zeros(nbytes, dtype=np.uint8)
    out = np.zeros((pl.height, pl.width, 3) if color else (pl.height, pl.width), dtype=np.uint8)
    status = np.zeros(4, dtype=np.int32)
    work = np.zeros(64, dtype=np.int32)
    p = lambda a: a.ctypes.data
    used, calls = 0, 0
    while True:
        rc = L.jd_emul_decode(p(pl.scan), pl.scan.size, ctypes.byref(pl.frame), p(pl.tab), p(pl.qt), p(pl.block_base), p(pl.seg_beg), p(pl.seg_end),
                              p(pl.seg_chunk0), p(pl.chunk_seg), p(out), out.strides[0], ch, sweeps, int(calls > 0), p(status),
                              p(ws), nbytes, order, p(work))
        assert rc == 0, rc
        calls += 1
        nz = np.flatnonzero(work[:sweeps])
        assert status[3] == (int(nz[-1]) + 1 if nz.size else 0)
        used = (calls - 1) * sweeps + int(status[3]) if status[3] else used
        if status[0] == 0 or calls >= max_calls:
            break
    return out, dict(status=status.copy(), sweeps=used, calls=calls)


def decode_components(cp, color: bool, sweeps: int = 64, order: int = 0):
    """A ``jpeg.ComponentPlans`` (multi-scan sequential file) through the lane model: one grey decode per component, then the colour
    stage on the three planes; grey output = the luma plane."""
    planes = [decode(pl, False, sweeps, order)[0] for pl in (cp.plans if color else cp.plans[:1])]
    if not color:
        return planes[0]
    y, cb, cr = [np.ascontiguousarray(a) for a in planes]
    assert cb.shape == cr.shape and y.shape == (cp.height, cp.width)
    out = np.zeros((cp.height, cp.width, 3), dtype=np.uint8)
    L = lib()
    L.jd_emul_planes_to_rgb.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64] + [ctypes.c_int] * 4 + \
        [ctypes.c_void_p, ctypes.c_int64]
    rc = L.jd_emul_planes_to_rgb(y.ctypes.data, y.strides[0], cb.ctypes.data, cr.ctypes.data, cb.strides[0], cp.width, cp.height,
                                 cp.sampling[0][0], cp.sampling[0][1], out.ctypes.data, out.strides[0])
    assert rc == 0, rc
    return out
