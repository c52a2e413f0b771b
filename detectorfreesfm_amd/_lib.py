"""ctypes binding of ``libdfsfm_hip.so`` (C ABI declared in ``include/dfsfm_hip.h``).

The library is the product: there is NO fallback.  If the shared object is missing or an entry
point is absent, importing this module's ``lib()`` raises, and every op built on it fails loudly.
"""
import ctypes
import os
from ctypes import c_float, c_int, c_int64, c_size_t, c_void_p, c_char_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# DFSFM_LIB_PATH lets kernel A/B experiments point at an alternative build of the same ABI.
LIB_PATH = os.environ.get("DFSFM_LIB_PATH") or os.path.join(_HERE, "csrc", "libdfsfm_hip.so")

DFSFM_OK = 0
_ERRORS = {-1: "DFSFM_E_BADARG", -2: "DFSFM_E_UNSUPPORTED", -3: "DFSFM_E_WORKSPACE", -4: "DFSFM_E_LAUNCH"}

# (name, restype, argtypes) -- must list every symbol declared in include/dfsfm_hip.h
SIGNATURES = [
    ("dfsfm_version", c_int, []),
    ("dfsfm_last_error_string", c_char_p, []),
    ("dfsfm_linear_attention_workspace", c_size_t, [c_int, c_int, c_int, c_int]),
    ("dfsfm_linear_attention_f32", c_int,
     [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p,
      c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float,
      c_void_p, c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    ("dfsfm_coarse_match_workspace", c_size_t, [c_int, c_int, c_int]),
    ("dfsfm_coarse_match_f32", c_int,
     [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_float, c_int,
      c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_float,
      c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
      c_void_p, c_size_t, c_void_p]),
    ("dfsfm_coarse_match_split", c_int,
     [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_float, c_int,
      c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_float,
      c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
      c_void_p, c_size_t, c_void_p]),
    ("dfsfm_coarse_match_split_masked", c_int,
     [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_float, c_int,
      c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_float,
      c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
      c_void_p, c_size_t, c_void_p]),
    ("dfsfm_coarse_conf_matrix_f32", c_int,
     [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_size_t, c_void_p]),
    ("dfsfm_roi_align_f32", c_int,
     [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
      c_float, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    ("dfsfm_fine_match_f32", c_int,
     [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
      c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64,
      c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    ("dfsfm_fine_match_split", c_int,
     [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
      c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64,
      c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    ("dfsfm_encoder_kv_f32", c_int,
     [c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    ("dfsfm_encoder_apply_f32", c_int,
     [c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_float,
      c_void_p, c_void_p, c_float, c_float, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int, c_void_p]),
    ("dfsfm_encoder256_state_workspace", c_size_t, [c_int, c_int]),
    ("dfsfm_encoder256_state_f32", c_int,
     [c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    ("dfsfm_encoder256_kv_workspace", c_size_t, [c_int, c_int]),
    ("dfsfm_encoder256_kv_f32", c_int,
     [c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    ("dfsfm_encoder256_apply_f32", c_int,
     [c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_float,
      c_void_p, c_void_p, c_float, c_float, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int, c_void_p]),
    ("dfsfm_layernorm_f32", c_int,
     [c_void_p, c_int64, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64,
      c_void_p, c_void_p, c_int64, c_int64, c_int, c_void_p]),
    ("dfsfm_split_rows_f32", c_int,
     [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int64, c_int,
      c_void_p]),
    ("dfsfm_split_rows_blocked_f32", c_int,
     [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_int64, c_int64, c_int, c_void_p]),
    ("dfsfm_dwconv3x3_nhwc_f32", c_int,
     [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    ("dfsfm_bilinear_up_nhwc_f32", c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    ("dfsfm_avgpool_nhwc_f32", c_int, [c_void_p, c_int64, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int64, c_void_p]),
    ("dfsfm_full_attention_f32", c_int,
     [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64,
      c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    ("dfsfm_span_attention_f32", c_int,
     [c_void_p, c_int64, c_int64, c_int, c_int, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_int, c_int, c_void_p,
      c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_float, c_float, c_void_p, c_int64, c_int, c_int, c_void_p]),
    ("dfsfm_layernorm2d_f32", c_int,
     [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int64,
      c_int64, c_int, c_void_p]),
    ("dfsfm_upsample_nhwc_f32", c_int,
     [c_void_p, c_int64, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p]),
    ("dfsfm_resize_bilinear_f32", c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    ("dfsfm_flow_decode_f32", c_int, [c_void_p, c_int64, c_int64, c_float, c_float, c_void_p, c_void_p]),
    ("dfsfm_resample_u8", c_int,
     [c_void_p, c_int64, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_int,
      c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    ("dfsfm_jpeg_decode_workspace", c_size_t, [c_void_p, c_int64, c_int]),
    ("dfsfm_jpeg_decode_u8", c_int,
     [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
      c_int, c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    ("dfsfm_jpeg_decode_batch_u8", c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    ("dfsfm_jpeg_ycc_planes_to_rgb_u8", c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_void_p, c_int64, c_void_p]),
    ("dfsfm_jpeg_scan_index", c_int64, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p]),
    ("dfsfm_resample_separable_f32", c_int,
     [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    ("dfsfm_add_scatter_tokens_f32", c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    ("dfsfm_conv2d_nhwc_f32", c_int,
     [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
      c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int,
      c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p, c_float, c_void_p]),
    ("dfsfm_conv2d_direct_f32", c_int,
     [c_void_p, c_int64, c_int64, c_int64, c_int, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int,
      c_void_p, c_int, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p]),
    ("dfsfm_merge_keypoints_workspace", c_size_t, [c_int64]),
    ("dfsfm_merge_keypoints", c_int,
     [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
      c_void_p, c_size_t, c_void_p]),
    ("dfsfm_maxpool3x3s2_nhwc_f32", c_int,
     [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    ("dfsfm_s2d_front_f32", c_int,
     [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int, c_int,
      c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
]

_lib = None


class DfsfmError(RuntimeError):
    pass


def lib() -> ctypes.CDLL:
    """Load (once) and return the HIP library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DfsfmError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C detectorfreesfm_amd/csrc`). There is no CPU/PyTorch fallback.")
        handle = ctypes.CDLL(LIB_PATH)
        for name, restype, argtypes in SIGNATURES:
            fn = getattr(handle, name)          # AttributeError if the symbol is missing
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = handle
    return _lib


def check(rc: int, what: str):
    if rc != DFSFM_OK:
        extra = ""
        if rc == -4:
            extra = ": " + (lib().dfsfm_last_error_string() or b"").decode()
        raise DfsfmError(f"{what} failed with {_ERRORS.get(rc, rc)}{extra}")


def source_sha256() -> str:
    """sha256 over the library's SOURCES (csrc/*.hip, *.h, Makefile, exports.map, include/dfsfm_hip.h; names and contents, sorted): the
    identity of a build that survives a rebuild -- the .so itself is not bit-reproducible across build directories.  Measurement files
    that cannot be produced by the timed process itself (the rocprofv3 PMC passes) carry it, and bench.py compares."""
    import glob
    import hashlib
    files = sorted(glob.glob(os.path.join(_HERE, "csrc", "*.hip")) + glob.glob(os.path.join(_HERE, "csrc", "*.h")) +
                   [os.path.join(_HERE, "csrc", "Makefile"), os.path.join(_HERE, "csrc", "exports.map"),
                    os.path.join(os.path.dirname(_HERE), "include", "dfsfm_hip.h")])
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
        h.update(b"\0")
    return h.hexdigest()
