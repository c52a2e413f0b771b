"""Torch-tensor front ends of the HIP kernels (raw pointers + current stream -> C ABI).

PyTorch is plumbing here: it owns the device memory and the stream.  Every function launches
hand-written gfx950 kernels from ``libdfsfm_hip.so``; none has a PyTorch/CPU fallback.
"""
import ctypes
import functools
import os
from typing import Optional

import numpy as np
import torch

from . import _lib

_workspaces = {}
FP16_MAX = 65504.0
# Range guards of the split-plane representation (|v| < 65504).
# * default: the FIRST call of every model entry point after ``load_state_dict`` runs inside ``range_sweep`` -- every
#   producer of split planes queues the abs-max of what it wrote (device scalars, no host sync per launch) and the sweep
#   reads them back ONCE at the end of the call; a saturated activation raises there instead of silently clamping.  Later
#   calls pay nothing.  DFSFM_RANGE_SWEEP=0 switches it off.
# * DFSFM_DEBUG_RANGE=1 / set_debug_range(True): every producer checks every launch at once (one host sync per launch).
# * Coverage.  The fused encoder layers (encoder_fused.hip, encoder256.hip) split q, k, v, the message, norm1(merge) and the MLP's
#   hidden layer in registers: no producer sees those.  While either guard is active (``range_check_active``) the layer therefore
#   ALSO runs on its five-GEMM form, where each of them is a checked producer (coarse.encoder_layer_split) -- the first call after a
#   weight load pays one extra transformer pass, later calls nothing.  What the default guard does NOT cover: range depends on the
#   INPUT as well as on the weights, and only the first call's input is swept; a deployment that wants every call checked sets
#   DFSFM_DEBUG_RANGE=1 (or wraps calls in ``range_sweep``) and pays for it.
_debug_range = os.environ.get("DFSFM_DEBUG_RANGE", "0") == "1"
RANGE_SWEEP = os.environ.get("DFSFM_RANGE_SWEEP", "1") != "0"
_sweep = None            # list of (producer, abs-max device scalar) while a sweep is open


def set_debug_range(on: bool):
    global _debug_range
    _debug_range = bool(on)


def check_split_range(sa, what: str):
    """Raises if a split-plane tensor holds a saturated element: split_f32 clamps hi to +-65504, the largest finite
    fp16, so |hi| == 65504 means the fp32 value was out of the representable range (or exactly on its edge)."""
    if sa.hi.numel() and not (float(sa.hi.float().abs().max()) < FP16_MAX):          # NaN fails the test as well
        raise _lib.DfsfmError(f"{what}: activation outside the split-plane range |v| < {FP16_MAX:.0f} "
                              "(the fp16x2 representation would saturate; rescale the layer)")


def _range(sa, what: str):
    """Called by every producer of split planes on its output."""
    if _debug_range:
        check_split_range(sa, what)
    elif _sweep is not None and sa.hi.numel():
        _sweep.append((what, sa.hi.abs().max()))


def range_note(what: str, value):
    """Queue (or, in debug mode, check at once) a device scalar that must stay below 65504: bounds on quantities that only the
    fused kernels ever hold as split planes (q, k, v, the per-head KV state), computed from the fp32 tensors of the five-GEMM form."""
    if _debug_range:
        if not (float(value) < FP16_MAX):
            raise _lib.DfsfmError(f"{what}: outside the split-plane range |v| < {FP16_MAX:.0f} (the fused encoder layer would saturate)")
    elif _sweep is not None:
        _sweep.append((what, value))


def range_check_active() -> bool:
    """True while split-plane producers are being checked (an open sweep or DFSFM_DEBUG_RANGE): callers that normally keep
    intermediates in registers (the fused encoder layers) then ALSO run their memory-going form so the guard sees them."""
    return _debug_range or _sweep is not None


class range_sweep:
    """Context manager: collect the abs-max of every split-plane tensor produced inside, check them with one host read."""

    def __init__(self, label: str, report=None):
        """report: optional list that receives (producer, abs-max) of everything the sweep saw (tools/verify_checkpoint.py)."""
        self.label, self.mine, self.report = label, False, report

    def __enter__(self):
        global _sweep
        if _sweep is None and RANGE_SWEEP and not _debug_range:
            _sweep, self.mine = [], True
        return self

    def __exit__(self, et, ev, tb):
        global _sweep
        if not self.mine:
            return False
        items, _sweep = _sweep, None
        if et is None and items:
            mx = torch.stack([v.float() for _, v in items]).cpu()
            if self.report is not None:
                self.report.extend((n, m) for (n, _), m in zip(items, mx.tolist()))
            bad = sorted({n for (n, _), m in zip(items, mx.tolist()) if not (m < FP16_MAX)})       # NaN / inf count as out of range
            if bad:
                raise _lib.DfsfmError(f"{self.label}: activations outside the split-plane range |v| < {FP16_MAX:.0f} after "
                                      f"{', '.join(bad)} (the fp16x2 representation saturates there; these weights need a "
                                      "rescaled layer -- DFSFM_DEBUG_RANGE=1 names the first launch)")
        return False


def first_call_range_sweep(fn):
    """Method decorator for the models' entry points: the first call after ``load_state_dict`` (``_range_done`` is cleared
    there, params.ParamModule) runs under ``range_sweep``."""
    @functools.wraps(fn)
    def wrapper(self, *a, **kw):
        done = self.__dict__.setdefault("_range_done", set())
        if fn.__name__ in done or not RANGE_SWEEP:
            return fn(self, *a, **kw)
        with range_sweep(f"{type(self).__name__}.{fn.__name__}"):
            out = fn(self, *a, **kw)
        done.add(fn.__name__)
        return out
    return wrapper


def _device_of(a):
    if isinstance(a, torch.Tensor):
        return a.device if a.is_cuda else None
    if isinstance(a, (SplitAct, PackedDense)):
        return a.hi.device if a.hi.is_cuda else None
    return None


def _on_device(fn):
    """Runs an op with its operands' device current, so that the launch stream (``_stream``), the workspace and the
    kernels all belong to that device even when the caller never called torch.cuda.set_device (a process may drive
    several GPUs); operands spread over different devices are refused."""
    @functools.wraps(fn)
    def wrapper(*args, **kw):
        devs = {d for d in map(_device_of, list(args) + list(kw.values())) if d is not None}
        if len(devs) > 1:
            raise _lib.DfsfmError(f"{fn.__name__}: operands live on different devices {sorted(map(str, devs))}")
        if not devs:
            return fn(*args, **kw)
        with torch.cuda.device(next(iter(devs))):
            return fn(*args, **kw)
    return wrapper


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _require_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.DfsfmError("HIP ops need device tensors (there is no CPU path)")


_GUARD = 65536 if os.environ.get("DFSFM_GUARD", "0") == "1" else 0       # debugging aid: canary bands around the workspaces
_GUARD_BYTE = 0xA5


def _workspace(nbytes: int, device) -> torch.Tensor:
    """Grow-only scratch buffer per (device, stream); kernels on one stream are ordered.  With DFSFM_GUARD=1 every workspace
    sits between two 64-KB bands of a known byte that ``check_workspace_guards`` verifies (tools/gpu_harden.sh)."""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    buf = _workspaces.get(key)
    if buf is None or buf.numel() - 2 * _GUARD < nbytes:
        buf = torch.empty(max(nbytes, 1 << 20) + 2 * _GUARD, dtype=torch.uint8, device=device)
        if _GUARD:
            buf[:_GUARD].fill_(_GUARD_BYTE)
            buf[buf.numel() - _GUARD:].fill_(_GUARD_BYTE)
        _workspaces[key] = buf
    return buf[_GUARD:buf.numel() - _GUARD] if _GUARD else buf


def check_workspace_guards():
    """Raises if a kernel wrote outside a workspace (only meaningful with DFSFM_GUARD=1)."""
    for key, buf in _workspaces.items():
        if _GUARD and not (bool((buf[:_GUARD] == _GUARD_BYTE).all()) and bool((buf[buf.numel() - _GUARD:] == _GUARD_BYTE).all())):
            raise _lib.DfsfmError(f"workspace {key}: guard band overwritten")


def _as_u8(m: Optional[torch.Tensor]):
    if m is None:
        return None
    if m.dtype == torch.bool:
        return m.contiguous().view(torch.uint8)
    return (m != 0).contiguous().view(torch.uint8)


@_on_device
def linear_attention(q, k, v, q_mask=None, kv_mask=None, q_group=1, kv_group=1, eps=1e-6, out=None,
                     out_split=False):
    """K1.  q [N,L,H,D]; k,v [N,S,H,D] fp32 (row-strided views allowed: stride(-1)=1,
    stride(-2)=D); q_mask [N,L/q_group], kv_mask [N,S/kv_group] bool/uint8 or None.
    out_split=True returns the message as a SplitAct [N*L, H*D] (for the merge GEMM) instead of fp32."""
    _require_cuda(q, k, v)
    N, L, H, D = q.shape
    S = k.shape[1]
    for t, n_rows in ((q, L), (k, S), (v, S)):
        if t.dtype != torch.float32 or t.stride(3) != 1 or t.stride(2) != D or (N > 1 and t.stride(0) != n_rows * t.stride(1)):
            raise _lib.DfsfmError("linear_attention: need fp32 [N,rows,H,D] with dense (H,D) and batch stride rows*ld")
    res = oh = ol = None
    ldo = ldos = H * D
    if out_split:
        res = SplitAct(torch.empty((N * L, H * D), dtype=torch.float16, device=q.device),
                       torch.empty((N * L, H * D), dtype=torch.float16, device=q.device), H * D)
        oh, ol, out = res.hi, res.lo, None
    else:
        if out is None:
            out = torch.empty((N, L, H, D), dtype=torch.float32, device=q.device)
        res, ldo = out, out.stride(1)
    qm, km = _as_u8(q_mask), _as_u8(kv_mask)
    lib = _lib.lib()
    ws_bytes = lib.dfsfm_linear_attention_workspace(N, S, H, D)
    if ws_bytes == 0:
        raise _lib.DfsfmError(f"linear_attention: unsupported shape N={N} S={S} H={H} D={D}")
    ws = _workspace(ws_bytes, q.device)
    rc = lib.dfsfm_linear_attention_f32(_ptr(q), _ptr(k), _ptr(v), _ptr(qm), q_group, _ptr(km), kv_group,
                                        _ptr(out), N, L, S, H, D, q.stride(1), k.stride(1), v.stride(1),
                                        ldo, eps, _ptr(oh), _ptr(ol), ldos, _ptr(ws), ws.numel(), _stream())
    _lib.check(rc, "dfsfm_linear_attention_f32")
    if out_split:
        _range(res, "linear_attention")
    return res


class PendingMatches:
    """Match tables of a ``coarse_match(..., defer=True)`` call whose row count is still on the device: full-capacity buffers + the
    count scalar.  ``result()`` performs the one host read (like ``torch.where`` in the reference) and returns the sliced dict.  A
    caller that loops over batches launches batch k + 1 before it asks for batch k's result, so the device never idles on the read."""

    def __init__(self, ids, mconf, mk, count):
        self.ids, self.mconf, self.mk, self.count = ids, mconf, mk, count

    def result(self):
        M = int(self.count.item())
        return {"b_ids": self.ids[0, :M], "i_ids": self.ids[1, :M], "j_ids": self.ids[2, :M], "mconf": self.mconf[:M],
                "mkpts0_c": self.mk[0, :M], "mkpts1_c": self.mk[1, :M]}


@_on_device
def coarse_match(feat0, feat1, hw0_c, hw1_c, thr, border, temperature, scale0=None, scale1=None,
                 coarse_scale=8.0, mask0=None, mask1=None, defer=False):
    """K3+K4+K5.  feat0 [N,L,C], feat1 [N,S,C]: fp32 contiguous tensors, or SplitAct planes (contiguous, C a
    power of 4) -- the correlation then runs on the fp16x2-split MFMA path.  Returns a dict with
    b_ids,i_ids,j_ids (int64 [M]), mconf [M], mkpts0_c, mkpts1_c [M,2] in ascending (b,i) order; ``defer=True`` returns a
    ``PendingMatches`` instead (no host synchronisation inside the call)."""
    split = isinstance(feat0, SplitAct)
    if split != isinstance(feat1, SplitAct):
        raise _lib.DfsfmError("coarse_match: feat0 and feat1 must both be fp32 or both be split planes")
    if split:
        _require_cuda(feat0.hi, feat0.lo, feat1.hi, feat1.lo)
        for t in (feat0.hi, feat0.lo, feat1.hi, feat1.lo):
            if not t.is_contiguous() or t.dtype != torch.float16:
                raise _lib.DfsfmError("coarse_match: split planes must be contiguous fp16")
        if feat0.hi.shape[-1] != feat0.C or feat1.hi.shape[-1] != feat1.C:
            raise _lib.DfsfmError("coarse_match: split planes must not carry padded channels")
        N, L, C = feat0.hi.shape
        S = feat1.hi.shape[1]
        dev = feat0.hi.device
    else:
        _require_cuda(feat0, feat1)
        feat0, feat1 = feat0.contiguous(), feat1.contiguous()
        N, L, C = feat0.shape
        S = feat1.shape[1]
        dev = feat0.device
    lib = _lib.lib()
    ws = _workspace(lib.dfsfm_coarse_match_workspace(N, L, S), dev)
    cap = N * L
    ids = torch.empty((3, cap), dtype=torch.int64, device=dev)
    mconf = torch.empty((cap,), dtype=torch.float32, device=dev)
    mk = torch.empty((2, cap, 2), dtype=torch.float32, device=dev)
    count = torch.zeros((1,), dtype=torch.int32, device=dev)
    s0 = None if scale0 is None else scale0.to(device=dev, dtype=torch.float32).contiguous()
    s1 = None if scale1 is None else scale1.to(device=dev, dtype=torch.float32).contiguous()
    if (mask0 is None) != (mask1 is None):
        raise _lib.DfsfmError("coarse_match: mask0 and mask1 come together")
    if mask0 is not None:
        if not split:
            raise _lib.DfsfmError("coarse_match: padding masks need the split-plane entry point")
        mk0 = _as_u8(mask0.to(dev).reshape(N, -1))
        mk1 = _as_u8(mask1.to(dev).reshape(N, -1))
        if mk0.shape != (N, L) or mk1.shape != (N, S):
            raise _lib.DfsfmError("coarse_match: masks must be [N, h0c, w0c] / [N, h1c, w1c]")
    tail = (N, L, S, C, float(temperature), float(thr), int(border), hw0_c[0], hw0_c[1], hw1_c[0], hw1_c[1],
            _ptr(s0), _ptr(s1), float(coarse_scale), _ptr(ids[0]), _ptr(ids[1]), _ptr(ids[2]), _ptr(mconf),
            _ptr(mk[0]), _ptr(mk[1]), _ptr(count), _ptr(ws), ws.numel(), _stream())
    if split and mask0 is not None:
        rc = lib.dfsfm_coarse_match_split_masked(_ptr(feat0.hi), _ptr(feat0.lo), _ptr(feat1.hi), _ptr(feat1.lo),
                                                 _ptr(mk0), _ptr(mk1), *tail)
        _lib.check(rc, "dfsfm_coarse_match_split_masked")
    elif split:
        rc = lib.dfsfm_coarse_match_split(_ptr(feat0.hi), _ptr(feat0.lo), _ptr(feat1.hi), _ptr(feat1.lo), *tail)
        _lib.check(rc, "dfsfm_coarse_match_split")
    else:
        rc = lib.dfsfm_coarse_match_f32(_ptr(feat0), _ptr(feat1), *tail)
        _lib.check(rc, "dfsfm_coarse_match_f32")
    pending = PendingMatches(ids, mconf, mk, count)
    return pending if defer else pending.result()      # data-dependent size, like torch.where in the reference


@_on_device
def coarse_conf_matrix(feat0, feat1, temperature):
    """Dense dual-softmax confidence matrix [N,L,S] (parity/debug aid)."""
    _require_cuda(feat0, feat1)
    feat0, feat1 = feat0.contiguous(), feat1.contiguous()
    N, L, C = feat0.shape
    S = feat1.shape[1]
    lib = _lib.lib()
    ws = _workspace(lib.dfsfm_coarse_match_workspace(N, L, S), feat0.device)
    conf = torch.empty((N, L, S), dtype=torch.float32, device=feat0.device)
    rc = lib.dfsfm_coarse_conf_matrix_f32(_ptr(feat0), _ptr(feat1), N, L, S, C, float(temperature), _ptr(conf),
                                          _ptr(ws), ws.numel(), _stream())
    _lib.check(rc, "dfsfm_coarse_conf_matrix_f32")
    return conf


@_on_device
def roi_align(feat, boxes, crop_h, crop_w, box_ind=None, out_slot=None, extrapolation_value=0.0,
              mean=None, std=None, out=None, channels_last=False):
    """K8.  feat [Nimg,C,H,W]; boxes [M,4] (x1,y1,x2,y2); returns / fills out [*,C,crop_h,crop_w]
    (or [*,crop_h,crop_w,C] with channels_last=True)."""
    _require_cuda(feat, boxes)
    feat = feat.contiguous()
    boxes = boxes.to(torch.float32).contiguous()
    Nimg, C, H, W = feat.shape
    M = boxes.shape[0]
    if out is None:
        if out_slot is not None:
            raise _lib.DfsfmError("roi_align: out_slot needs a preallocated `out`")
        shape = (M, crop_h, crop_w, C) if channels_last else (M, C, crop_h, crop_w)
        out = torch.empty(shape, dtype=torch.float32, device=feat.device)
    bi = None if box_ind is None else box_ind.to(torch.int32).contiguous()
    sl = None if out_slot is None else out_slot.to(torch.int64).contiguous()
    rc = _lib.lib().dfsfm_roi_align_f32(_ptr(feat), Nimg, C, H, W, _ptr(boxes), _ptr(bi), _ptr(sl), M, crop_h,
                                        crop_w, float(extrapolation_value), _ptr(mean), _ptr(std), _ptr(out),
                                        1 if channels_last else 0, _stream())
    _lib.check(rc, "dfsfm_roi_align_f32")
    return out


@_on_device
def fine_match(ref, qry, track_mask, movable, W, left, query_pts=None, scale_q=None, ref_pts=None,
               scale_r=None):
    """K11+K12.  ref [T,WW,C], qry [T,Vq,WW,C] fp32 -- or both as SplitAct planes of those shapes (the form the encoder kernels
    write; streamed into the MFMA fragments without a conversion); track_mask [T,Vq]; movable [T] or None.
    query_pts / scale_q [T,2]; ref_pts / scale_r view-major [>=Vq, T, 2] (the reference's [V-1,T,2] layout; strided
    views such as ``x[0, :, i:]`` are taken as they are).  Any float dtype / device placement of the four point
    tensors is accepted like in the reference and converted to device fp32 here.
    Returns dict(best_index, left_norm, coords, std[, query_refined, ref_refined])."""
    split = isinstance(ref, SplitAct)
    if split != isinstance(qry, SplitAct):
        raise _lib.DfsfmError("fine_match: ref and qry must both be fp32 or both be split planes")
    if split:
        _require_cuda(ref.hi, ref.lo, qry.hi, qry.lo)
        for t in (ref.hi, ref.lo, qry.hi, qry.lo):
            if t.dtype != torch.float16 or not t.is_contiguous():
                raise _lib.DfsfmError("fine_match: split planes must be contiguous fp16")
        if ref.hi.shape[-1] != ref.C or qry.hi.shape[-1] != qry.C or ref.lo.shape != ref.hi.shape or qry.lo.shape != qry.hi.shape:
            raise _lib.DfsfmError("fine_match: split planes must not carry padded channels")
        T, Vq, WW, C = qry.hi.shape
        if tuple(ref.hi.shape) != (T, WW, C):
            raise _lib.DfsfmError("fine_match: ref must be [T, W*W, C] matching qry [T, Vq, W*W, C]")
        dev = ref.hi.device
    else:
        _require_cuda(ref, qry)
        if ref.dtype != torch.float32 or qry.dtype != torch.float32 or ref.device != qry.device:
            raise _lib.DfsfmError("fine_match: ref / qry must be fp32 tensors on one device")
        ref, qry = ref.contiguous(), qry.contiguous()
        T, Vq, WW, C = qry.shape
        if ref.shape != (T, WW, C):
            raise _lib.DfsfmError("fine_match: ref must be [T, W*W, C] matching qry [T, Vq, W*W, C]")
        dev = ref.device
    tm = _as_u8(track_mask.to(dev))
    mv = None if movable is None else _as_u8(movable.to(dev))
    if tm.shape != (T, Vq) or (mv is not None and mv.shape != (T,)):
        raise _lib.DfsfmError("fine_match: mask shapes")

    def f32(t, shape):
        if t is None:
            return None
        t = t.to(device=dev, dtype=torch.float32)
        if t.shape != shape:
            raise _lib.DfsfmError(f"fine_match: expected a tensor of shape {shape}, got {tuple(t.shape)}")
        return t

    query_pts, scale_q = f32(query_pts, (T, 2)), f32(scale_q, (T, 2))
    if (query_pts is None) != (scale_q is None) or (ref_pts is None) != (scale_r is None):
        raise _lib.DfsfmError("fine_match: points and their scales come in pairs")
    if query_pts is not None:
        query_pts, scale_q = query_pts.contiguous(), scale_q.contiguous()
    rs_t = rs_n = 0
    if ref_pts is not None:
        if ref_pts.dim() != 3 or ref_pts.shape[0] < Vq or ref_pts.shape[1:] != (T, 2) or scale_r.shape != ref_pts.shape:
            raise _lib.DfsfmError("fine_match: ref_pts / scale_r must be view-major [>=Vq, T, 2]")
        ref_pts, scale_r = f32(ref_pts, ref_pts.shape), f32(scale_r, scale_r.shape)

        def pair_layout(t):
            return t.stride(2) == 1 and t.stride(1) % 2 == 0 and t.stride(0) % 2 == 0
        if not (pair_layout(ref_pts) and pair_layout(scale_r) and ref_pts.stride() == scale_r.stride()):
            ref_pts, scale_r = ref_pts.contiguous(), scale_r.contiguous()
        rs_n, rs_t = ref_pts.stride(0) // 2, ref_pts.stride(1) // 2
    best = torch.empty((T,), dtype=torch.int32, device=dev)
    left_norm = torch.empty((T, 2), dtype=torch.float32, device=dev)
    coords = torch.empty((T, Vq, 2), dtype=torch.float32, device=dev)
    std = torch.empty((T, Vq), dtype=torch.float32, device=dev)
    qref = torch.empty((T, 2), dtype=torch.float32, device=dev) if query_pts is not None else None
    rref = torch.empty((T, Vq, 2), dtype=torch.float32, device=dev) if ref_pts is not None else None
    tail = (_ptr(tm), _ptr(mv), T, Vq, W, left, C, _ptr(query_pts), _ptr(scale_q), _ptr(ref_pts), _ptr(scale_r),
            rs_t, rs_n, _ptr(best), _ptr(left_norm), _ptr(coords), _ptr(std), _ptr(qref), _ptr(rref), _stream())
    if split:
        rc = _lib.lib().dfsfm_fine_match_split(_ptr(ref.hi), _ptr(ref.lo), _ptr(qry.hi), _ptr(qry.lo), *tail)
        _lib.check(rc, "dfsfm_fine_match_split")
    else:
        rc = _lib.lib().dfsfm_fine_match_f32(_ptr(ref), _ptr(qry), *tail)
        _lib.check(rc, "dfsfm_fine_match_f32")
    out = {"best_index": best, "left_norm": left_norm, "coords": coords, "std": std}
    if qref is not None:
        out["query_refined"] = qref
    if rref is not None:
        out["ref_refined"] = rref
    return out


def _rows_ld(t: torch.Tensor, dtype=torch.float32):
    """(rows, row stride) of a tensor whose last dim is dense and whose leading dims flatten
    uniformly (e.g. a column slice of a contiguous [..., 2C] buffer)."""
    if t.dtype != dtype or t.stride(-1) != 1:
        raise _lib.DfsfmError("need rows of the expected dtype with unit inner stride")
    ld = t.stride(-2) if t.dim() >= 2 else t.shape[-1]
    for i in range(t.dim() - 2):
        if t.shape[i] != 1 and t.stride(i) != t.stride(i + 1) * t.shape[i + 1]:
            raise _lib.DfsfmError("leading dims do not flatten to uniformly strided rows")
    rows = 1
    for s in t.shape[:-1]:
        rows *= s
    return rows, ld


@_on_device
def layernorm(x, gamma, beta, eps=1e-5, residual=None, out=None, out_split=None, want_f32=True):
    """(residual or 0) + LayerNorm(x)*gamma + beta over the last dim (row-strided views OK).
    Written as fp32 into ``out`` (allocated if None and want_f32) and/or as split planes into the
    SplitAct view ``out_split`` (e.g. a column slice of the [x | message] split buffer)."""
    _require_cuda(x, gamma, beta)
    rows, ldx = _rows_ld(x)
    C = x.shape[-1]
    ldo = 0
    if out is None and want_f32:
        out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    if out is not None:
        rows_o, ldo = _rows_ld(out)
        if rows_o != rows or out.shape[-1] != C:
            raise _lib.DfsfmError("layernorm: out shape mismatch")
    oh = ol = None
    ldos = 0
    if out_split is not None:
        rows_s, ldos = _rows_ld(out_split.hi, torch.float16)
        if rows_s != rows or out_split.hi.shape[-1] != C or out_split.lo.stride() != out_split.hi.stride():
            raise _lib.DfsfmError("layernorm: split out shape mismatch")
        oh, ol = out_split.hi, out_split.lo
    ldr = 0
    r32 = rh = rl = None
    if isinstance(residual, SplitAct):
        rows_r, ldr = _rows_ld(residual.hi, torch.float16)
        if rows_r != rows or residual.hi.shape[-1] != C or residual.lo.stride() != residual.hi.stride():
            raise _lib.DfsfmError("layernorm: split residual shape mismatch")
        rh, rl = residual.hi, residual.lo
    elif residual is not None:
        rows_r, ldr = _rows_ld(residual)
        if rows_r != rows:
            raise _lib.DfsfmError("layernorm: residual shape mismatch")
        r32 = residual
    rc = _lib.lib().dfsfm_layernorm_f32(_ptr(x), ldx, _ptr(gamma), _ptr(beta), float(eps), _ptr(r32), _ptr(rh),
                                        _ptr(rl), ldr, _ptr(out), ldo, _ptr(oh), _ptr(ol), ldos, rows, C, _stream())
    _lib.check(rc, "dfsfm_layernorm_f32")
    if out_split is not None:
        _range(out_split, "layernorm")
    return out


@_on_device
def split_rows(x, add=None, out=None, out_split=None):
    """out / out_split = x (+ add broadcast over row blocks); x [..., C] fp32 rows, add [R, C] contiguous."""
    _require_cuda(x)
    C = x.shape[-1]
    if (x.dim() == 3 and x.dtype == torch.float32 and x.stride(2) == 1 and x.shape[0] > 1 and x.stride(0) != x.stride(1) * x.shape[1]
            and add is None and out is None and out_split is not None):
        # [B, R, C] blocks with their own outer stride (e.g. the tokens of some views of every track): no contiguous copy
        rows_s, ldos = _rows_ld(out_split.hi, torch.float16)
        if rows_s != x.shape[0] * x.shape[1] or out_split.hi.shape[-1] != C:
            raise _lib.DfsfmError("split_rows: split out shape mismatch")
        rc = _lib.lib().dfsfm_split_rows_blocked_f32(_ptr(x), x.shape[1], x.stride(0), x.stride(1), _ptr(out_split.hi),
                                                     _ptr(out_split.lo), ldos, rows_s, C, _stream())
        _lib.check(rc, "dfsfm_split_rows_blocked_f32")
        _range(out_split, "split_rows")
        return
    rows, ldx = _rows_ld(x)
    ldo = ldos = 0
    if out is not None:
        _, ldo = _rows_ld(out)
    oh = ol = None
    if out_split is not None:
        _, ldos = _rows_ld(out_split.hi, torch.float16)
        oh, ol = out_split.hi, out_split.lo
    add_rows = 0
    if add is not None:
        add = add.contiguous()
        add_rows = add.shape[0]
    rc = _lib.lib().dfsfm_split_rows_f32(_ptr(x), ldx, _ptr(add), add_rows, _ptr(out), ldo, _ptr(oh), _ptr(ol), ldos,
                                         rows, C, _stream())
    _lib.check(rc, "dfsfm_split_rows_f32")
    if out_split is not None:
        _range(out_split, "split_rows")


@_on_device
def dwconv3x3(x, w, bias, mode=0, out_split=False):
    """Depth-wise 3x3 conv (groups = C, pad 1) + bias on x [N,H,W,C] fp32 NHWC (contiguous) with the consumer fused:
    mode 0 plain, 1 ``x * sigmoid(.)`` (MatchFormer Positional), 2 erf-GELU (MatchFormer Mlp).  w [C,1,3,3] or the packed
    [9,C] form; returns fp32 [N,H,W,C] or a SplitAct."""
    _require_cuda(x, w, bias)
    N, H, W, C = x.shape
    if x.dtype != torch.float32 or not x.is_contiguous():
        raise _lib.DfsfmError("dwconv3x3: need a contiguous fp32 NHWC tensor")
    w9c = w if w.dim() == 2 else w.reshape(C, 9).t().contiguous()
    if w9c.shape != (9, C):
        raise _lib.DfsfmError("dwconv3x3: weight must be [C,1,3,3] or [9,C]")
    w9c, bias = w9c.to(torch.float32).contiguous(), bias.to(torch.float32).contiguous()
    out = oh = ol = res = None
    if out_split:
        res = SplitAct.empty(N, H, W, C, x.device)
        oh, ol = res.hi, res.lo
    else:
        res = out = torch.empty((N, H, W, C), dtype=torch.float32, device=x.device)
    rc = _lib.lib().dfsfm_dwconv3x3_nhwc_f32(_ptr(x), N, H, W, C, _ptr(w9c), _ptr(bias), int(mode), _ptr(out), _ptr(oh),
                                             _ptr(ol), _stream())
    _lib.check(rc, "dfsfm_dwconv3x3_nhwc_f32")
    if out_split:
        _range(res, "dwconv3x3")
    return res


@_on_device
def bilinear_up(x, hout, wout):
    """F.interpolate(mode='bilinear', align_corners=True) of x [N,hin,win,C] fp32 NHWC -> [N,hout,wout,C]."""
    _require_cuda(x)
    N, hin, win, C = x.shape
    if x.dtype != torch.float32 or not x.is_contiguous():
        raise _lib.DfsfmError("bilinear_up: need a contiguous fp32 NHWC tensor")
    out = torch.empty((N, hout, wout, C), dtype=torch.float32, device=x.device)
    rc = _lib.lib().dfsfm_bilinear_up_nhwc_f32(_ptr(x), N, hin, win, C, int(hout), int(wout), _ptr(out), _stream())
    _lib.check(rc, "dfsfm_bilinear_up_nhwc_f32")
    return out


# ---------------------------------------------------------------- ASpanFormer pieces (csrc/aspan_ops.hip)
def _split_out(out_split, rows, C, what):
    if out_split is None:
        return None, None, 0
    rows_s, ldos = _rows_ld(out_split.hi, torch.float16)
    if rows_s != rows or out_split.hi.shape[-1] != C or out_split.lo.stride() != out_split.hi.stride():
        raise _lib.DfsfmError(f"{what}: split out shape mismatch")
    return out_split.hi, out_split.lo, ldos


@_on_device
def avgpool(x, k, out=None):
    """F.avg_pool2d(x, k, stride=k) of x [N,H,W,C] fp32 (token-pitched view OK) -> [N,H/k,W/k,C]."""
    _require_cuda(x)
    N, H, W, C = x.shape
    rows, ldx = _rows_ld(x)
    if out is None:
        out = torch.empty((N, H // k, W // k, C), dtype=torch.float32, device=x.device)
    rows_o, ldo = _rows_ld(out)
    if H % k or W % k or rows_o != N * (H // k) * (W // k) or out.shape[-1] != C:
        raise _lib.DfsfmError("avgpool: shape mismatch")
    rc = _lib.lib().dfsfm_avgpool_nhwc_f32(_ptr(x), ldx, N, H, W, C, int(k), _ptr(out), ldo, _stream())
    _lib.check(rc, "dfsfm_avgpool_nhwc_f32")
    return out


def _batched_rows(t, what):
    """(pitch, batch stride) of a [N, L, C] fp32 view with dense channels."""
    if t.dtype != torch.float32 or t.dim() != 3 or t.stride(2) != 1:
        raise _lib.DfsfmError(f"{what}: need fp32 [N, L, C] views with dense channels")
    return t.stride(1), t.stride(0)


@_on_device
def full_attention(q, k, v, nhead, scale, kv_swap=False):
    """softmax(scale * q k^T) v per head; q [N,L,C], k / v [N,S,C] fp32 views; kv_swap pairs batch n with n ^ 1."""
    _require_cuda(q, k, v)
    N, L, C = q.shape
    S = k.shape[1]
    if k.shape != (N, S, C) or v.shape != (N, S, C) or C % nhead:
        raise _lib.DfsfmError("full_attention: shape mismatch")
    (ldq, sq), (ldk, sk), (ldv, sv) = (_batched_rows(t, "full_attention") for t in (q, k, v))
    out = torch.empty((N, L, C), dtype=torch.float32, device=q.device)
    rc = _lib.lib().dfsfm_full_attention_f32(_ptr(q), ldq, sq, _ptr(k), ldk, sk, _ptr(v), ldv, sv, _ptr(out), C, L * C, N, L, S,
                                             int(nhead), C // nhead, 1 if kv_swap else 0, float(scale), _stream())
    _lib.check(rc, "dfsfm_full_attention_f32")
    return out


@_on_device
def span_attention(q, hw, k, v, hw_k, flow, hw0, sample_offset, nhead, nsample, radius_scale, temp=1.0, kv_swap=False):
    """One level of ASpanFormer's hierarchical attention for N images: q [N, h*w, C] (level maps), k / v [N, hk*wk, C] (image
    n takes those of image n ^ kv_swap), flow [N, H0*W0, 4] (full resolution) -> [N, h*w, C] in the reference's (group, member)
    order.  2-D q / k / v / flow are one image."""
    _require_cuda(q, k, v, flow, sample_offset)
    single = q.dim() == 2
    if single:
        q, k, v, flow = q[None], k[None], v[None], flow.reshape(1, -1, 4)
    N, rq, C = q.shape
    (ldq, sq), (ldk, sk), (ldv, sv) = (_batched_rows(t, "span_attention") for t in (q, k, v))
    flow = flow.reshape(N, -1, 4)
    if (rq != hw[0] * hw[1] or k.shape != (N, hw_k[0] * hw_k[1], C) or v.shape != k.shape or not flow.is_contiguous()
            or flow.dtype != torch.float32 or flow.shape[1] != hw0[0] * hw0[1]):
        raise _lib.DfsfmError("span_attention: shape mismatch")
    so = sample_offset.to(torch.float32).contiguous()
    if so.shape != (nsample[1] ** 2, 2):
        raise _lib.DfsfmError("span_attention: sample_offset is [nsample1^2, 2]")
    out = torch.empty((N, rq, C), dtype=torch.float32, device=q.device)
    rc = _lib.lib().dfsfm_span_attention_f32(_ptr(q), ldq, sq, hw[0], hw[1], _ptr(k), ldk, sk, _ptr(v), ldv, sv, hw_k[0], hw_k[1],
                                             _ptr(flow), hw0[0], hw0[1], _ptr(so), int(nhead), C, int(nsample[0]), int(nsample[1]),
                                             float(radius_scale), float(temp), _ptr(out), C, N, 1 if kv_swap else 0, _stream())
    _lib.check(rc, "dfsfm_span_attention_f32")
    return out[0] if single else out


@_on_device
def layernorm2d(x, affine, bias, residual=None, out=None, out_split=None, want_f32=True):
    """(residual or 0) + affine * (x - mean) / (std_unbiased + 1e-6) + bias over the last dim of fp32 rows (views OK);
    residual: SplitAct view; results as fp32 (``out``, allocated when None and want_f32) and / or into ``out_split``."""
    _require_cuda(x, affine, bias)
    rows, ldx = _rows_ld(x)
    C = x.shape[-1]
    ldo = 0
    if out is None and want_f32:
        out = torch.empty((rows, C), dtype=torch.float32, device=x.device)
    if out is not None:
        rows_o, ldo = _rows_ld(out)
        if rows_o != rows or out.shape[-1] != C:
            raise _lib.DfsfmError("layernorm2d: out shape mismatch")
    oh, ol, ldos = _split_out(out_split, rows, C, "layernorm2d")
    rh = rl = None
    ldr = 0
    if residual is not None:
        rows_r, ldr = _rows_ld(residual.hi, torch.float16)
        if rows_r != rows or residual.hi.shape[-1] != C or residual.lo.stride() != residual.hi.stride():
            raise _lib.DfsfmError("layernorm2d: residual shape mismatch")
        rh, rl = residual.hi, residual.lo
    rc = _lib.lib().dfsfm_layernorm2d_f32(_ptr(x), ldx, _ptr(affine), _ptr(bias), _ptr(rh), _ptr(rl), ldr, _ptr(out), ldo,
                                          _ptr(oh), _ptr(ol), ldos, rows, C, _stream())
    _lib.check(rc, "dfsfm_layernorm2d_f32")
    if out_split is not None:
        _range(out_split, "layernorm2d")
    return out


@_on_device
def upsample(x, scale, bilinear, out=None, out_split=None, want_f32=True):
    """F.upsample(x, scale_factor=scale, mode='bilinear' | 'nearest') of x [N,hin,win,C] fp32 (token-pitched view OK) into
    fp32 rows ``out`` [N*hin*scale*win*scale, C] (allocated dense when None and want_f32) and / or ``out_split`` rows."""
    _require_cuda(x)
    N, hin, win, C = x.shape
    _, ldx = _rows_ld(x)
    rows = N * hin * scale * win * scale
    ldo = 0
    if out is None and want_f32:
        out = torch.empty((N, hin * scale, win * scale, C), dtype=torch.float32, device=x.device)
    if out is not None:
        rows_o, ldo = _rows_ld(out)
        if rows_o != rows or out.shape[-1] != C:
            raise _lib.DfsfmError("upsample: out shape mismatch")
    oh, ol, ldos = _split_out(out_split, rows, C, "upsample")
    rc = _lib.lib().dfsfm_upsample_nhwc_f32(_ptr(x), ldx, N, hin, win, C, int(scale), 1 if bilinear else 0, _ptr(out), ldo,
                                            _ptr(oh), _ptr(ol), ldos, _stream())
    _lib.check(rc, "dfsfm_upsample_nhwc_f32")
    return out


@_on_device
def resize_bilinear(x, hout, wout):
    """F.interpolate(x, size=(hout, wout), mode='bilinear', align_corners=False) of fp32 planes x [..., hin, win]."""
    _require_cuda(x)
    if x.dtype != torch.float32 or x.dim() < 2:
        raise _lib.DfsfmError("resize_bilinear: need fp32 planes [..., H, W]")
    x = x.contiguous()
    hin, win = x.shape[-2:]
    N = x.numel() // (hin * win)
    out = torch.empty((*x.shape[:-2], int(hout), int(wout)), dtype=torch.float32, device=x.device)
    rc = _lib.lib().dfsfm_resize_bilinear_f32(_ptr(x), N, hin, win, int(hout), int(wout), _ptr(out), _stream())
    _lib.check(rc, "dfsfm_resize_bilinear_f32")
    return out


@_on_device
def flow_decode(x, wk, hk):
    """(sigmoid(x0) * wk, sigmoid(x1) * hk, x2, x3) of the first four columns of fp32 rows x -> [rows, 4]."""
    _require_cuda(x)
    rows, ldx = _rows_ld(x)
    out = torch.empty((rows, 4), dtype=torch.float32, device=x.device)
    rc = _lib.lib().dfsfm_flow_decode_f32(_ptr(x), ldx, rows, float(wk), float(hk), _ptr(out), _stream())
    _lib.check(rc, "dfsfm_flow_decode_f32")
    return out


@_on_device
def resample_u8(src, bounds_x, kk_x, bounds_y, kk_y, out_u8=False, lut=None, pad_hw=None, want_mask=False):
    """PIL's 8-bit two-pass resampling of src [H,W] or [H,W,3] uint8 with the given int32 tables (images.py builds
    them) -> (u8 [Hn,Wn(,3)] or None, fp32 [C,pad_h,pad_w] = lut[byte] or None, mask [pad_h,pad_w] or None)."""
    _require_cuda(src, bounds_x, kk_x, bounds_y, kk_y)
    if src.dtype != torch.uint8 or src.dim() not in (2, 3) or src.stride(-1) != 1 or (src.dim() == 3 and src.stride(1) != src.shape[2]):
        raise _lib.DfsfmError("resample_u8: need a uint8 [H,W] or [H,W,3] image with dense rows")
    for t in (bounds_x, kk_x, bounds_y, kk_y):
        if t.dtype != torch.int32 or not t.is_contiguous() or t.dim() != 2:
            raise _lib.DfsfmError("resample_u8: coefficient tables are contiguous int32 matrices")
    H, W = src.shape[:2]
    C = 1 if src.dim() == 2 else src.shape[2]
    Wn, Hn = bounds_x.shape[0], bounds_y.shape[0]
    if bounds_x.shape[1] != 2 or bounds_y.shape[1] != 2 or kk_x.shape[0] != Wn or kk_y.shape[0] != Hn:
        raise _lib.DfsfmError("resample_u8: bounds are [out,2], coefficients [out,ksize]")
    if not out_u8 and lut is None:
        raise _lib.DfsfmError("resample_u8: no output requested")
    dev = src.device
    tmp = torch.empty((H * Wn * C,), dtype=torch.uint8, device=dev)
    o8 = torch.empty((Hn, Wn) if src.dim() == 2 else (Hn, Wn, C), dtype=torch.uint8, device=dev) if out_u8 else None
    of, mk, ph, pw = None, None, 0, 0
    if lut is not None:
        _require_cuda(lut)
        if lut.dtype != torch.float32 or lut.numel() != 256 or not lut.is_contiguous():
            raise _lib.DfsfmError("resample_u8: lut is 256 contiguous fp32 values")
        ph, pw = (Hn, Wn) if pad_hw is None else (int(pad_hw[0]), int(pad_hw[1]))
        of = torch.empty((C, ph, pw), dtype=torch.float32, device=dev)
        mk = torch.empty((ph, pw), dtype=torch.float32, device=dev) if want_mask else None
    elif want_mask:
        raise _lib.DfsfmError("resample_u8: the mask comes with the fp32 output")
    rc = _lib.lib().dfsfm_resample_u8(_ptr(src), src.stride(0), H, W, C, _ptr(bounds_x), _ptr(kk_x), kk_x.shape[1], Wn,
                                      _ptr(bounds_y), _ptr(kk_y), kk_y.shape[1], Hn, _ptr(tmp), _ptr(o8), _ptr(of), _ptr(mk),
                                      ph, pw, _ptr(lut), _stream())
    _lib.check(rc, "dfsfm_resample_u8")
    return o8, of, mk


_jpeg_pool = []              # pinned staging buffers not in use (allocating pinned memory per call costs more than a decode)


def _jpeg_staging(nbytes: int):
    for i, buf in enumerate(_jpeg_pool):
        if buf.numel() >= nbytes:
            return _jpeg_pool.pop(i)
    if _jpeg_pool:
        _jpeg_pool.pop()                                     # too small: let it go
    return torch.empty((max(nbytes * 5 // 4, 1 << 20),), dtype=torch.uint8, pin_memory=True)


class JpegDecodeCall:
    """One ``dfsfm_jpeg_decode_u8`` in flight: ``jpeg_decode_launch`` uploads the scan and its small tables in ONE copy and
    queues every kernel on the stream that is current; ``finish()`` reads the 16-byte status back on that stream, continues the
    relaxation (resume) with twice as many sweeps if its fixed point was not reached, raises on corrupt streams and returns
    (uint8 [H,W] or [H,W,3] device tensor, dict(sweeps, sweeps_used, calls)).  Several calls may be in flight on different
    streams (``jpeg.decode_many``): a decode is a chain of small latency-bound launches that leaves most of the chip idle."""

    def __init__(self, pl, out_channels, device, sweeps, max_calls):
        from . import jpeg as _jpeg
        self._jpeg, self.pl, self.out_channels, self.device = _jpeg, pl, out_channels, device
        self.sweeps, self.max_calls = sweeps, max_calls
        self.stream = torch.cuda.current_stream(device)
        L = _lib.lib()
        fr = pl.frame
        self.nbytes = L.dfsfm_jpeg_decode_workspace(ctypes.byref(fr), pl.scan.size, out_channels)
        if self.nbytes == 0:
            raise _jpeg.UnsupportedJpeg("frame outside the device decoder (dfsfm_jpeg_decode_workspace)")
        # one host buffer = one H2D copy: [scan | tab | qt | block_base | seg_beg | seg_end | seg_chunk0 | chunk_seg], 16-byte aligned
        parts = [pl.scan, pl.tab.view(np.uint8), pl.qt.reshape(-1).view(np.uint8), pl.block_base.view(np.uint8),
                 pl.seg_beg.view(np.uint8), pl.seg_end.view(np.uint8), pl.seg_chunk0.view(np.uint8), pl.chunk_seg.view(np.uint8)]
        offs, o = [], 0
        for a in parts:
            offs.append(o)
            o = (o + a.size + 15) // 16 * 16
        self.host = _jpeg_staging(o)                         # pinned; back in the pool when finish() has seen the status
        hv = self.host.numpy()
        for a, at in zip(parts, offs):
            hv[at:at + a.size] = a
        self.devbuf = self.host[:o].to(device, non_blocking=True)
        base = self.devbuf.data_ptr()
        self.ptrs = [ctypes.c_void_p(base + at) for at in offs]
        self.out = torch.empty((fr.height, fr.width) if out_channels == 1 else (fr.height, fr.width, 3), dtype=torch.uint8, device=device)
        self.status = torch.empty((4,), dtype=torch.int32, device=device)
        self.ws = torch.empty((self.nbytes,), dtype=torch.uint8, device=device)
        self.calls, self.total, self.used = 0, 0, 0
        self._launch()

    def _launch(self):
        p, pl = self.ptrs, self.pl
        rc = _lib.lib().dfsfm_jpeg_decode_u8(p[0], pl.scan.size, ctypes.byref(pl.frame), p[1], p[2], p[3], p[4], p[5], p[6], p[7],
                                             _ptr(self.out), self.out.stride(0), self.out_channels, self.sweeps, int(self.calls > 0),
                                             _ptr(self.status), _ptr(self.ws), self.nbytes, ctypes.c_void_p(self.stream.cuda_stream))
        _lib.check(rc, "dfsfm_jpeg_decode_u8")
        self.calls += 1
        self.total += self.sweeps

    def finish(self):
        with torch.cuda.device(self.device), torch.cuda.stream(self.stream):
            try:
                while True:
                    st = self.status.tolist()
                    self.used = self.total - self.sweeps + st[3] if st[3] else self.used
                    if st[0] == 0:
                        break
                    # the relaxation provably settles within pl.launch_bound launches (jpeg.flat_launch_bound); a stream that is
                    # still moving after that many is not one a consistent decoder state sequence explains
                    bound = int(getattr(self.pl, "launch_bound", 0)) or 64 * self.max_calls
                    if self.total >= bound:
                        raise self._jpeg.CorruptJpeg(f"entropy decode did not reach its fixed point in {self.total} sweep launches "
                                                     f"(bound {bound})")
                    self.sweeps = max(1, min(64, 2 * self.sweeps, bound - self.total))
                    self._launch()
            finally:
                _jpeg_pool.append(self.host)
                self.host = None
        if st[1] or st[2]:
            raise self._jpeg.CorruptJpeg(f"corrupt scan: {st[1]} invalid codes, {st[2]} restart intervals with a wrong block count")
        return self.out, dict(sweeps=self.total, sweeps_used=self.used, calls=self.calls)


def jpeg_decode_launch(pl, out_channels: int, device, sweeps: int = 4, max_calls: int = 8) -> JpegDecodeCall:
    device = torch.device(device)
    if device.type != "cuda":
        raise _lib.DfsfmError("HIP ops need device tensors (there is no CPU path)")
    if out_channels not in (1, 3):
        raise _lib.DfsfmError("jpeg_decode: out_channels is 1 (luma) or 3 (RGB)")
    with torch.cuda.device(device):
        return JpegDecodeCall(pl, out_channels, device, sweeps, max_calls)


def jpeg_decode(pl, out_channels: int, device, sweeps: int = 4, max_calls: int = 8):
    """``dfsfm_jpeg_decode_u8`` on a parsed file (``jpeg.Plan``), synchronously: launch + finish of ``JpegDecodeCall``."""
    return jpeg_decode_launch(pl, out_channels, device, sweeps, max_calls).finish()


def jpeg_planes_to_rgb(y, cb, cr, width: int, height: int, luma_sampling):
    """``dfsfm_jpeg_ycc_planes_to_rgb_u8``: the colour stage of the JPEG decode on three component planes (uint8 device tensors of the
    components' real samples) -> RGB [height, width, 3]."""
    _require_cuda(y, cb, cr)
    h0, v0 = int(luma_sampling[0]), int(luma_sampling[1])
    cw, chh = -(-width // h0), -(-height // v0)
    for t, shp in ((y, (height, width)), (cb, (chh, cw)), (cr, (chh, cw))):
        if t.dtype != torch.uint8 or tuple(t.shape) != shp or t.stride(1) != 1:
            raise _lib.DfsfmError("jpeg_planes_to_rgb: planes must be uint8 [rows, cols] of the components' real samples")
    if cb.stride(0) != cr.stride(0):
        raise _lib.DfsfmError("jpeg_planes_to_rgb: cb / cr strides differ")
    out = torch.empty((height, width, 3), dtype=torch.uint8, device=y.device)
    rc = _lib.lib().dfsfm_jpeg_ycc_planes_to_rgb_u8(_ptr(y), y.stride(0), _ptr(cb), _ptr(cr), cb.stride(0), width, height, h0, v0,
                                                    _ptr(out), out.stride(0), _stream())
    _lib.check(rc, "dfsfm_jpeg_ycc_planes_to_rgb_u8")
    return out


class JpegJob(ctypes.Structure):
    """``dfsfm_jpeg_job`` of include/dfsfm_hip.h."""
    _fields_ = [("scan", ctypes.c_void_p), ("scan_bytes", ctypes.c_int64), ("frame_host", ctypes.c_void_p),
                ("huff_tab", ctypes.c_void_p), ("qt", ctypes.c_void_p), ("block_base", ctypes.c_void_p), ("seg_beg", ctypes.c_void_p),
                ("seg_end", ctypes.c_void_p), ("seg_chunk0", ctypes.c_void_p), ("chunk_seg", ctypes.c_void_p), ("out", ctypes.c_void_p),
                ("out_stride", ctypes.c_int64), ("status", ctypes.c_void_p), ("workspace", ctypes.c_void_p), ("workspace_bytes", ctypes.c_size_t)]


class JpegBatchCall:
    """``dfsfm_jpeg_decode_batch_u8`` in flight for a list of parsed files (``jpeg.Plan``): ONE pinned staging buffer and one H2D copy
    for all scans and tables, one workspace allocation, one [n, 4] status tensor, one set of launches per seven files.
    ``finish()`` reads the statuses back once; files whose relaxation has not settled (rare: flat frames beyond the default launches)
    continue as a smaller batch with ``resume``; returns the list of (uint8 tensor or the exception of that file)."""

    def __init__(self, plans, out_channels, device, sweeps):
        from . import jpeg as _jpeg
        self._jpeg, self.plans, self.out_channels, self.device, self.sweeps = _jpeg, list(plans), out_channels, device, sweeps
        self.stream = torch.cuda.current_stream(device)
        L = _lib.lib()
        n = len(self.plans)
        self.ws_bytes, parts_of, offs_of, o = [], [], [], 0
        for pl in self.plans:
            nb = L.dfsfm_jpeg_decode_workspace(ctypes.byref(pl.frame), pl.scan.size, out_channels)
            if nb == 0:
                raise _jpeg.UnsupportedJpeg("frame outside the device decoder (dfsfm_jpeg_decode_workspace)")
            self.ws_bytes.append((nb + 255) // 256 * 256)
            parts = [pl.scan, pl.tab.view(np.uint8), pl.qt.reshape(-1).view(np.uint8), pl.block_base.view(np.uint8),
                     pl.seg_beg.view(np.uint8), pl.seg_end.view(np.uint8), pl.seg_chunk0.view(np.uint8), pl.chunk_seg.view(np.uint8)]
            offs = []
            for a in parts:
                offs.append(o)
                o = (o + a.size + 15) // 16 * 16
            parts_of.append(parts)
            offs_of.append(offs)
        self.host = _jpeg_staging(max(o, 16))
        hv = self.host.numpy()
        for parts, offs in zip(parts_of, offs_of):
            for a, at in zip(parts, offs):
                hv[at:at + a.size] = a
        self.devbuf = self.host[:max(o, 16)].to(device, non_blocking=True)
        self.ws = torch.empty((sum(self.ws_bytes),), dtype=torch.uint8, device=device)
        self.status = torch.empty((n, 4), dtype=torch.int32, device=device)
        self.outs = [torch.empty((pl.frame.height, pl.frame.width) if out_channels == 1 else (pl.frame.height, pl.frame.width, 3),
                                 dtype=torch.uint8, device=device) for pl in self.plans]
        base, wbase = self.devbuf.data_ptr(), self.ws.data_ptr()
        self.jobs = (JpegJob * max(n, 1))()
        w = 0
        for i, (pl, offs) in enumerate(zip(self.plans, offs_of)):
            j = self.jobs[i]
            j.scan, j.scan_bytes, j.frame_host = base + offs[0], pl.scan.size, ctypes.addressof(pl.frame)
            j.huff_tab, j.qt, j.block_base, j.seg_beg, j.seg_end, j.seg_chunk0, j.chunk_seg = (base + offs[k] for k in range(1, 8))
            j.out, j.out_stride = self.outs[i].data_ptr(), self.outs[i].stride(0)
            j.status, j.workspace, j.workspace_bytes = self.status.data_ptr() + 16 * i, wbase + w, self.ws_bytes[i]
            w += self.ws_bytes[i]
        self.total = [0] * n
        self._launch(list(range(n)), resume=False)

    def _launch(self, idx, resume):
        if not idx:
            return
        jobs = (JpegJob * len(idx))(*[self.jobs[i] for i in idx])
        rc = _lib.lib().dfsfm_jpeg_decode_batch_u8(ctypes.byref(jobs), len(idx), self.out_channels, self.sweeps, int(resume),
                                                   ctypes.c_void_p(self.stream.cuda_stream))
        _lib.check(rc, "dfsfm_jpeg_decode_batch_u8")
        for i in idx:
            self.total[i] += self.sweeps

    def finish(self):
        results = [None] * len(self.plans)
        with torch.cuda.device(self.device), torch.cuda.stream(self.stream):
            try:
                pending = list(range(len(self.plans)))
                while pending:
                    st = self.status.tolist()
                    again = []
                    for i in pending:
                        bound = int(getattr(self.plans[i], "launch_bound", 0)) or 512
                        if st[i][0] != 0 and self.total[i] < bound:
                            again.append(i)
                        elif st[i][0] != 0:
                            results[i] = self._jpeg.CorruptJpeg(f"entropy decode did not reach its fixed point in {self.total[i]} sweep launches")
                        elif st[i][1] or st[i][2]:
                            results[i] = self._jpeg.CorruptJpeg(f"corrupt scan: {st[i][1]} invalid codes, {st[i][2]} restart intervals with a "
                                                                "wrong block count")
                        else:
                            results[i] = self.outs[i]
                    if again:
                        self.sweeps = min(64, 2 * self.sweeps)
                        self._launch(again, resume=True)
                    pending = again
            finally:
                _jpeg_pool.append(self.host)
                self.host = None
        return results


def jpeg_decode_batch_launch(plans, out_channels: int, device, sweeps: int = 4) -> JpegBatchCall:
    device = torch.device(device)
    if device.type != "cuda":
        raise _lib.DfsfmError("HIP ops need device tensors (there is no CPU path)")
    if out_channels not in (1, 3):
        raise _lib.DfsfmError("jpeg_decode_batch: out_channels is 1 (luma) or 3 (RGB)")
    with torch.cuda.device(device):
        return JpegBatchCall(plans, out_channels, device, sweeps)


@_on_device
def resample_separable(y, By, Bx, out=None):
    """out[m, oy*wout+ox, c] = sum By[oy,qy] Bx[ox,qx] y[m,qy,qx,c];  y [M,hin,win,C] fp32 contiguous (NHWC patches),
    By [hout,hin], Bx [wout,win] fp32 -> [M, hout*wout, C]."""
    _require_cuda(y, By, Bx)
    M, hin, win, C = y.shape
    hout, wout = By.shape[0], Bx.shape[0]
    if y.dtype != torch.float32 or not y.is_contiguous() or By.shape[1] != hin or Bx.shape[1] != win:
        raise _lib.DfsfmError("resample_separable: need contiguous fp32 y [M,hin,win,C] and By [hout,hin], Bx [wout,win]")
    By, Bx = By.to(torch.float32).contiguous(), Bx.to(torch.float32).contiguous()
    if out is None:
        out = torch.empty((M, hout * wout, C), dtype=torch.float32, device=y.device)
    rc = _lib.lib().dfsfm_resample_separable_f32(_ptr(y), M, hin, win, C, _ptr(By), _ptr(Bx), hout, wout, _ptr(out), _stream())
    _lib.check(rc, "dfsfm_resample_separable_f32")
    return out


@_on_device
def add_scatter_tokens(a, b, slot, dst):
    """dst[slot[m], p, c] = a[m, c, p] (+ b[m, c, p]);  a,b [M,C,P] contiguous, dst [*,P,C] contiguous."""
    _require_cuda(a, dst)
    a = a.contiguous()
    b = None if b is None else b.contiguous()
    M, C, P = a.shape
    if not dst.is_contiguous() or dst.shape[-2:] != (P, C):
        raise _lib.DfsfmError("add_scatter_tokens: dst must be contiguous [*, P, C]")
    sl = None if slot is None else slot.to(torch.int64).contiguous()
    rc = _lib.lib().dfsfm_add_scatter_tokens_f32(_ptr(a), _ptr(b), _ptr(sl), _ptr(dst), M, C, P, _stream())
    _lib.check(rc, "dfsfm_add_scatter_tokens_f32")
    return dst


class SplitAct:
    """Activation tensor in the split form the conv kernel DMAs straight into LDS:
    value = hi + lo/2048 with fp16 planes hi, lo of shape [N,H,W,Cpad]; channels >= C are zeros."""

    def __init__(self, hi, lo, C):
        self.hi, self.lo, self.C = hi, lo, C

    @property
    def shape(self):
        return (*self.hi.shape[:-1], self.C)

    @staticmethod
    def empty_rows(shape, C, device):
        """Split planes [*shape, C] for row tensors (C % 8 == 0)."""
        return SplitAct(torch.empty((*shape, C), dtype=torch.float16, device=device),
                        torch.empty((*shape, C), dtype=torch.float16, device=device), C)

    def cols(self, a, b):
        """Column slice view [..., a:b] (both planes)."""
        return SplitAct(self.hi[..., a:b], self.lo[..., a:b], b - a)

    def __getitem__(self, idx):
        return SplitAct(self.hi[idx], self.lo[idx], self.C)

    @staticmethod
    def empty(N, H, W, C, device):
        cpad = (C + 7) // 8 * 8
        return SplitAct(torch.empty((N, H, W, cpad), dtype=torch.float16, device=device),
                        torch.empty((N, H, W, cpad), dtype=torch.float16, device=device), C)

    def crop(self, y0, y1, x0, x1):
        return SplitAct(self.hi[:, y0:y1, x0:x1], self.lo[:, y0:y1, x0:x1], self.C)

    def float(self):
        """Exact fp32 value (test/debug aid)."""
        return (self.hi.float() + self.lo.float() / 2048.0)[..., :self.C]


# (Cin, kh, kw, Cout) -> (stride, pad) served by dfsfm_conv2d_direct_f32, and the shapes routed to it by default
# (the 7x7 stem measures faster on the implicit-GEMM kernel: 0.43 vs 0.65 ms at 16 x 480x640)
_DIRECT_SHAPES = {(1, 7, 7, 128): (2, 3), (3, 3, 3, 64): (1, 1)}
_DIRECT_DEFAULT = {(3, 3, 3, 64)}


class PackedDense:
    """Weights of one conv / linear layer in the layout dfsfm_conv2d_nhwc_f32 consumes:
    fp16 hi / lo [ceil128(Cout), Kpad], K = kh*kw*Cin_pad in (ky,kx,ci) order, w = hi + lo/2048.
    ``cin_pad`` (>= Cin, multiple of 8) matches the channel padding of a SplitAct input."""

    def __init__(self, w: torch.Tensor, bias=None, cin_pad=None, tap_padded=False):
        """tap_padded: every tap's channel run is padded to a multiple of 32 (K = (ky,kx,ceil32(Cin))), the
        layout of the activation-reuse kernel for stride-1 'same' 3x3 / 5x5 convolutions on split inputs."""
        if w.dim() == 2:
            w = w[:, :, None, None]
        Cout, Cin, kh, kw = w.shape
        cp = Cin if cin_pad is None else cin_pad
        self.Cin_act = cp                       # channels of the activation tensor this layer reads
        self.tap_padded = bool(tap_padded)
        if tap_padded:
            cp = (cp + 31) // 32 * 32
        K = kh * kw * cp
        self.Cout, self.Cin, self.kh, self.kw = Cout, cp, kh, kw
        self.Kpad = (K + 31) // 32 * 32
        npad = (Cout + 127) // 128 * 128
        # float64 weights (a BN fold done in double) are split from their exact values; fp32 weights are unchanged by this
        wd = torch.float64 if w.dtype == torch.float64 else torch.float32
        wk = torch.zeros((Cout, kh, kw, cp), dtype=wd, device=w.device)
        wk[..., :Cin] = w.detach().to(wd).permute(0, 2, 3, 1)
        full = torch.zeros((npad, self.Kpad), dtype=wd, device=w.device)
        full[:Cout, :K] = wk.reshape(Cout, K)
        if not bool(torch.isfinite(full).all()) or float(full.abs().max()) >= FP16_MAX:
            raise _lib.DfsfmError(f"PackedDense: weights must be finite with |w| < {FP16_MAX:.0f} (split-plane range)")
        hi = torch.where(full.abs() >= 2.0 ** -14, full, torch.zeros_like(full)).half()
        self.hi = hi.contiguous()
        self.lo = ((full - hi.to(wd)) * 2048.0).half().contiguous()
        self.bias = None if bias is None else bias.detach().float().contiguous()
        # first layers (K = kh*kw*Cin of 49 / 27): fp32 weights [K][Cout] for the direct FMA kernel
        self.w32 = None
        self.use_direct = False
        if cin_pad is None and not tap_padded and (Cin, kh, kw, Cout) in _DIRECT_SHAPES:
            self.w32 = w.detach().float().permute(2, 3, 1, 0).reshape(kh * kw * Cin, Cout).contiguous()
            self.use_direct = (Cin, kh, kw, Cout) in _DIRECT_DEFAULT


@_on_device
def conv2d_nhwc(x, pw: PackedDense, stride=1, pad=0, residual=None, relu=False, out=None, out_split=False):
    """K6/K9.  x: fp32 [N,H,W,Cin] NHWC view, or a SplitAct.  residual: fp32 [.., Cout] view or SplitAct.
    Returns fp32 [N,Ho,Wo,Cout] (or fills ``out``), or a SplitAct when ``out_split`` (True: a new one; a SplitAct: filled)."""
    split_in = isinstance(x, SplitAct)
    xt = x.hi if split_in else x
    _require_cuda(xt)
    N, H, W, Cin = xt.shape
    want = torch.float16 if split_in else torch.float32
    if Cin != pw.Cin_act or xt.dtype != want or (Cin > 1 and xt.stride(3) != 1):
        raise _lib.DfsfmError("conv2d_nhwc: bad input")
    if pw.tap_padded and not (split_in and stride == 1 and pw.kh == pw.kw and pad == pw.kw // 2):
        raise _lib.DfsfmError("conv2d_nhwc: tap-padded weights need a split-input stride-1 'same' convolution")
    if split_in and (x.lo.shape != x.hi.shape or x.lo.stride() != x.hi.stride()):
        raise _lib.DfsfmError("conv2d_nhwc: hi/lo planes differ")
    Ho = (H + 2 * pad - pw.kh) // stride + 1
    Wo = (W + 2 * pad - pw.kw) // stride + 1
    dev = xt.device
    o32 = oh = ol = None
    ldo = ldo_s = cout_s = 0
    result = None
    if isinstance(out_split, SplitAct):             # fill the caller's split planes (rows of Cout channels, any row stride)
        result = out_split
        rows_s, ldo_s = _rows_ld(result.hi, torch.float16)
        if rows_s != N * Ho * Wo or result.C != pw.Cout or result.hi.shape[-1] != pw.Cout or result.lo.shape != result.hi.shape \
                or result.lo.stride() != result.hi.stride():
            raise _lib.DfsfmError("conv2d_nhwc: out_split shape mismatch")
        oh, ol, cout_s = result.hi, result.lo, pw.Cout
    elif out_split:
        result = SplitAct.empty(N, Ho, Wo, pw.Cout, dev)
        oh, ol, cout_s, ldo_s = result.hi, result.lo, result.hi.shape[-1], result.hi.shape[-1]
    else:
        if out is None:
            out = torch.empty((N, Ho, Wo, pw.Cout), dtype=torch.float32, device=dev)
        rows_o, ldo = _rows_ld(out)
        if rows_o != N * Ho * Wo or out.shape[-1] != pw.Cout:
            raise _lib.DfsfmError("conv2d_nhwc: out shape mismatch")
        o32, result = out, out
    r32 = rh = rl = None
    ldr = 0
    if residual is not None:
        if isinstance(residual, SplitAct):
            rh, rl = residual.hi, residual.lo
            if not rh.is_contiguous() or rh.numel() != N * Ho * Wo * rh.shape[-1] or residual.C != pw.Cout:
                raise _lib.DfsfmError("conv2d_nhwc: split residual mismatch")
            ldr = rh.shape[-1]
        else:
            rows_r, ldr = _rows_ld(residual)
            if rows_r != N * Ho * Wo or residual.shape[-1] != pw.Cout:
                raise _lib.DfsfmError("conv2d_nhwc: residual shape mismatch")
            r32 = residual
    sxn = xt.stride(0) if N > 1 else H * xt.stride(1)
    if (pw.use_direct and pw.w32 is not None and not split_in and residual is None and pw.w32.device == dev
            and _DIRECT_SHAPES[(Cin, pw.kh, pw.kw, pw.Cout)] == (stride, pad)):
        rc = _lib.lib().dfsfm_conv2d_direct_f32(
            _ptr(x), sxn, xt.stride(1), xt.stride(2), N, H, W, Cin, _ptr(pw.w32), pw.Cout, pw.kh, pw.kw, stride, pad,
            _ptr(pw.bias), 1 if relu else 0, _ptr(o32), ldo, _ptr(oh), _ptr(ol), ldo_s, _stream())
        _lib.check(rc, "dfsfm_conv2d_direct_f32")
        if out_split:
            _range(result, "conv2d_nhwc(direct)")
        return result
    rc = _lib.lib().dfsfm_conv2d_nhwc_f32(
        None if split_in else _ptr(x), _ptr(x.hi) if split_in else None, _ptr(x.lo) if split_in else None,
        sxn, xt.stride(1), xt.stride(2), N, H, W, Cin, _ptr(pw.hi), _ptr(pw.lo), pw.Cout, pw.Kpad, pw.kh, pw.kw,
        stride, pad, _ptr(pw.bias), _ptr(r32), _ptr(rh), _ptr(rl), ldr, int(relu),
        _ptr(o32), ldo, _ptr(oh), _ptr(ol), ldo_s, cout_s, 1 if pw.tap_padded else 0, None, None, 0.0, _stream())
    _lib.check(rc, "dfsfm_conv2d_nhwc_f32")
    if out_split:
        _range(result, f"conv2d_nhwc({pw.kh}x{pw.kw}, {Cin}->{pw.Cout})")
    return result


def linear(x, pw: PackedDense, residual=None, relu=False, out=None, out_split=False):
    """K2.  x [rows, K] fp32 (row-strided view OK) or a SplitAct of such rows; @ W^T (+bias, +residual, relu).
    Returns fp32 [rows, Cout], or a SplitAct [rows, Cout] when out_split."""
    if isinstance(x, SplitAct):
        rows, ld = _rows_ld(x.hi, torch.float16)
        K = x.hi.shape[-1]
        x4 = SplitAct(x.hi.as_strided((1, 1, rows, K), (0, 0, ld, 1)), x.lo.as_strided((1, 1, rows, K), (0, 0, ld, 1)), K)
        dev = x.hi.device
    else:
        rows, ld = _rows_ld(x)
        x4 = x.as_strided((1, 1, rows, x.shape[-1]), (0, 0, ld, 1))
        dev = x.device
    if out_split:
        y = conv2d_nhwc(x4, pw, 1, 0, residual, relu, out_split=True)
        return SplitAct(y.hi.view(rows, -1), y.lo.view(rows, -1), pw.Cout)
    if out is None:
        out = torch.empty((rows, pw.Cout), dtype=torch.float32, device=dev)
    conv2d_nhwc(x4, pw, 1, 0, residual, relu, out)
    return out


@_on_device
def linear_ln(x: "SplitAct", pw: PackedDense, gamma, beta, eps=1e-5, residual=None, out=None, out_split=None):
    """residual + LayerNorm(x @ W^T + bias) * gamma + beta in one kernel (LayerNorm fused into the GEMM epilogue;
    Cout must be 64, 128 or 256 so a row sits in one workgroup tile).  x: SplitAct rows [rows, K]; residual: fp32 [rows, Cout]
    row-strided view, a SplitAct view, or None; results go to the fp32 view ``out`` and/or the SplitAct view ``out_split``."""
    if not isinstance(x, SplitAct) or (out is None and out_split is None):
        raise _lib.DfsfmError("linear_ln: needs split input rows and at least one output")
    _require_cuda(x.hi, gamma, beta)
    rows, ld = _rows_ld(x.hi, torch.float16)
    K = x.hi.shape[-1]
    if K != pw.Cin_act or x.lo.stride() != x.hi.stride():
        raise _lib.DfsfmError("linear_ln: bad input")
    ldo = ldo_s = ldr = 0
    oh = ol = None
    if out is not None:
        rows_o, ldo = _rows_ld(out)
        if rows_o != rows or out.shape[-1] != pw.Cout:
            raise _lib.DfsfmError("linear_ln: out shape mismatch")
    if out_split is not None:
        rows_s, ldo_s = _rows_ld(out_split.hi, torch.float16)
        if rows_s != rows or out_split.hi.shape[-1] != pw.Cout or out_split.lo.stride() != out_split.hi.stride():
            raise _lib.DfsfmError("linear_ln: split out shape mismatch")
        oh, ol = out_split.hi, out_split.lo
    r32 = rh = rl = None
    if isinstance(residual, SplitAct):
        rows_r, ldr = _rows_ld(residual.hi, torch.float16)
        if rows_r != rows or residual.hi.shape[-1] != pw.Cout or residual.lo.stride() != residual.hi.stride():
            raise _lib.DfsfmError("linear_ln: split residual shape mismatch")
        rh, rl = residual.hi, residual.lo
    elif residual is not None:
        rows_r, ldr = _rows_ld(residual)
        if rows_r != rows or residual.shape[-1] != pw.Cout:
            raise _lib.DfsfmError("linear_ln: residual shape mismatch")
        r32 = residual
    rc = _lib.lib().dfsfm_conv2d_nhwc_f32(
        None, _ptr(x.hi), _ptr(x.lo), rows * ld, rows * ld, ld, 1, 1, rows, K, _ptr(pw.hi), _ptr(pw.lo), pw.Cout,
        pw.Kpad, 1, 1, 1, 0, _ptr(pw.bias), _ptr(r32), _ptr(rh), _ptr(rl), ldr, 0, _ptr(out), ldo, _ptr(oh), _ptr(ol),
        ldo_s, pw.Cout if oh is not None else 0, 0, _ptr(gamma), _ptr(beta), float(eps), _stream())
    _lib.check(rc, "dfsfm_conv2d_nhwc_f32(ln)")
    if out_split is not None:
        _range(out_split, "linear_ln")
    return out


# ---------------------------------------------------------------- fused encoder layer (csrc/encoder_fused.hip)
ENC_C, ENC_SLAB, ENC_KV_IMAGE = 128, 16384, 16 * 1024 + 512


def _split_planes(w):
    """fp32 / fp64 matrix -> (hi, lo) fp16 planes, w = hi + lo / 2048 (the rule of PackedDense)."""
    w = w.detach()
    wd = w.dtype if w.dtype == torch.float64 else torch.float32
    w = w.to(wd)
    if not bool(torch.isfinite(w).all()) or float(w.abs().max()) >= FP16_MAX:
        raise _lib.DfsfmError(f"EncoderFusedWeights: weights must be finite with |w| < {FP16_MAX:.0f} (split-plane range)")
    hi = torch.where(w.abs() >= 2.0 ** -14, w, torch.zeros_like(w)).half()
    return hi, ((w - hi.to(wd)) * 2048.0).half()


def _kslots(kstep: int, d_order: bool) -> torch.Tensor:
    """Column indices [2 (lane half), 8 (slot)] an MFMA 32x32x16 fragment holds for k-step ``kstep``.
    natural: 16 kstep + 8 half + j.  D order (the k index the accumulator of a previous MFMA imposes when it is fed back as
    an operand: register r of lane half h holds channel (r & 3) + 8 (r >> 2) + 4 h): 16 kstep + (j & 3) + 8 (j >> 2) + 4 half."""
    j = torch.arange(8)
    h = torch.arange(2)[:, None]
    return 16 * kstep + ((j & 3) + 8 * (j >> 2) + 4 * h if d_order else 8 * h + j)


def _fragment(plane: torch.Tensor, row0: int, kstep: int, d_order: bool) -> torch.Tensor:
    """One 1-KB MFMA operand fragment [64 lanes, 8 halves] of rows row0..row0+31: lane l = (row l & 31, half l >> 5)."""
    cols = _kslots(kstep, d_order).to(plane.device)                     # [2, 8]
    rows = plane[row0:row0 + 32]                                        # [32, K]
    return torch.stack([rows[:, cols[0]], rows[:, cols[1]]], 0).reshape(64, 8)


class EncoderFusedWeights:
    """Weights of one LoFTREncoderLayer (d_model 128, 8 heads) as the two fragment streams of csrc/encoder_fused.hip.

    A stream is a sequence of 16-KB slabs; a slab is 16 fragments of 1 KB (64 lanes x 16 bytes, lane-linear), (hi, lo)
    pairs in the order the kernel consumes them, so that the kernel's global -> LDS copy is linear and its LDS reads are
    ``base + fragment * 1024 + lane * 16``.
      kv stream (8 slabs, natural k order; enc_kv_kernel): slab 2 p + u = k-steps 4u..4u+3 of the rows
          [W_k rows of head pair p (32) | W_v rows of head pair p (32)]; fragment order (ks, block, plane).
      apply stream (32 slabs, D-order k; enc_apply_kernel): 4 slabs W_q (k-steps 2s, 2s+1 x 4 row blocks), 4 slabs merge,
          then per 64-channel chunk hc of the MLP's hidden layer: 4 slabs of mlp.0 rows 64hc..64hc+63 (k-steps 4u..4u+3 of
          the 256 input columns [x | norm1(message)] x 2 row blocks) and 2 slabs of mlp.2 (its columns 64hc + 16 (2v + ks),
          4 row blocks)."""

    def __init__(self, wq, wk, wv, wmerge, w1, w2, n1, n2, nhead=8):
        C = ENC_C
        if nhead != 8 or tuple(wq.shape) != (C, C) or tuple(w1.shape) != (2 * C, 2 * C) or tuple(w2.shape) != (C, 2 * C):
            raise _lib.DfsfmError("EncoderFusedWeights: the fused encoder layer is built for d_model 128, 8 heads")
        dev = wq.device
        # the streams are assembled on the host (640 row gathers per layer: thousands of tiny launches on the device) and
        # uploaded once
        planes = {k: _split_planes(v.detach().cpu()) for k, v in (("q", wq), ("k", wk), ("v", wv), ("m", wmerge), ("1", w1), ("2", w2))}
        # the values the kernels multiply with (hi + lo / 2048), under the reference's parameter names: what the CPU
        # stand-ins of the tests evaluate the layer with
        self.values = {n: (planes[k][0].float() + planes[k][1].float() / 2048.0).to(dev) for n, k in
                       (("q_proj.weight", "q"), ("k_proj.weight", "k"), ("v_proj.weight", "v"), ("merge.weight", "m"),
                        ("mlp.0.weight", "1"), ("mlp.2.weight", "2"))}

        def pair(name, row0, kstep, d_order):
            return [_fragment(planes[name][0], row0, kstep, d_order), _fragment(planes[name][1], row0, kstep, d_order)]
        frags = []
        for p in range(4):                               # kv stream
            for u in range(2):
                for ks in range(4):
                    frags += pair("k", 32 * p, 4 * u + ks, False) + pair("v", 32 * p, 4 * u + ks, False)
        self.kv_stream = torch.stack(frags, 0).contiguous().to(dev)          # [128, 64, 8] fp16 = 8 slabs
        frags = []
        for name in ("q", "m"):
            for s_ in range(4):
                for ks in range(2):
                    for b in range(4):
                        frags += pair(name, 32 * b, 2 * s_ + ks, True)
        for hc in range(4):
            for u in range(4):
                for ks in range(4):
                    for b in range(2):
                        frags += pair("1", 64 * hc + 32 * b, 4 * u + ks, True)
            for v2 in range(2):
                for ks in range(2):
                    for b in range(4):
                        frags += pair("2", 32 * b, 4 * hc + 2 * v2 + ks, True)
        self.apply_stream = torch.stack(frags, 0).contiguous().to(dev)       # [512, 64, 8] fp16 = 32 slabs
        assert self.kv_stream.numel() * 2 == 8 * ENC_SLAB and self.apply_stream.numel() * 2 == 32 * ENC_SLAB
        self.n1 = tuple(t.detach().float().contiguous() for t in n1)
        self.n2 = tuple(t.detach().float().contiguous() for t in n2)
        self.values.update({"norm1.weight": self.n1[0], "norm1.bias": self.n1[1], "norm2.weight": self.n2[0],
                            "norm2.bias": self.n2[1]})


def _enc_rows(x: "SplitAct", what):
    """(N, rows per sequence, row stride) of a SplitAct [N, L, 128] view whose rows are uniformly strided."""
    if x.hi.dim() != 3 or x.hi.shape[-1] != ENC_C or x.hi.dtype != torch.float16 or x.lo.stride() != x.hi.stride():
        raise _lib.DfsfmError(f"{what}: need split planes [N, L, 128]")
    rows, ld = _rows_ld(x.hi, torch.float16)
    return x.hi.shape[0], x.hi.shape[1], ld


@_on_device
def encoder_kv(src: "SplitAct", fw: EncoderFusedWeights, kv_mask=None, kv_group=1):
    """First half of a fused encoder layer: source tokens [N, S, 128] (split planes) -> the per-sequence attention state
    (KV^T fragments + Ksum) for ``encoder_apply``; k and v never reach memory."""
    _require_cuda(src.hi)
    N, S, ld = _enc_rows(src, "encoder_kv")
    img = torch.empty((N, ENC_KV_IMAGE), dtype=torch.uint8, device=src.hi.device)
    km = _as_u8(kv_mask)
    if km is not None and km.shape != (N, (S + kv_group - 1) // kv_group):
        raise _lib.DfsfmError("encoder_kv: kv_mask must be [N, ceil(S / kv_group)]")
    rc = _lib.lib().dfsfm_encoder_kv_f32(_ptr(src.hi), _ptr(src.lo), ld, N, S, _ptr(fw.kv_stream), _ptr(km), int(kv_group),
                                         _ptr(img), _stream())
    _lib.check(rc, "dfsfm_encoder_kv_f32")
    return img


@_on_device
def encoder_apply(x: "SplitAct", fw: EncoderFusedWeights, kv_image, S, q_mask=None, q_group=1, out_split=None, out=None,
                  eps=1e-5, attn_eps=1e-6, debug_stage=0):
    """Second half: x tokens [N, L, 128] (split planes) + attention state -> x + norm2(mlp([x | norm1(merge(attention))])) into
    the SplitAct view ``out_split`` and / or the fp32 view ``out`` ([N, L, 128] each).  debug_stage != 0 additionally returns
    an fp32 [N*L, 128] dump of that intermediate (tests)."""
    _require_cuda(x.hi, kv_image)
    N, L, ldx = _enc_rows(x, "encoder_apply")
    if kv_image.shape != (N, ENC_KV_IMAGE) or kv_image.dtype != torch.uint8 or not kv_image.is_contiguous():
        raise _lib.DfsfmError("encoder_apply: kv_image must come from encoder_kv for the same N")
    oh = ol = None
    ldo = ldo32 = 0
    if out_split is not None:
        n2, l2, ldo = _enc_rows(out_split, "encoder_apply(out_split)")
        if (n2, l2) != (N, L):
            raise _lib.DfsfmError("encoder_apply: out_split shape mismatch")
        oh, ol = out_split.hi, out_split.lo
    if out is not None:
        rows_o, ldo32 = _rows_ld(out)
        if rows_o != N * L or out.shape[-1] != ENC_C:
            raise _lib.DfsfmError("encoder_apply: out shape mismatch")
    qm = _as_u8(q_mask)
    if qm is not None and qm.shape != (N, (L + q_group - 1) // q_group):
        raise _lib.DfsfmError("encoder_apply: q_mask must be [N, ceil(L / q_group)]")
    # debug_stage 100: stage time stamps of wave 0 of every tile (tools/bench_encoder_fused.py)
    dbg = torch.zeros((max(N * L, (N * L + 127) // 128), ENC_C), dtype=torch.float32, device=x.hi.device) if debug_stage else None
    rc = _lib.lib().dfsfm_encoder_apply_f32(_ptr(x.hi), _ptr(x.lo), ldx, N, L, int(S), _ptr(fw.apply_stream), _ptr(kv_image),
                                            _ptr(qm), int(q_group), _ptr(fw.n1[0]), _ptr(fw.n1[1]), float(eps), _ptr(fw.n2[0]),
                                            _ptr(fw.n2[1]), float(eps), float(attn_eps), _ptr(oh), _ptr(ol), ldo, _ptr(out),
                                            ldo32, _ptr(dbg), int(debug_stage), _stream())
    _lib.check(rc, "dfsfm_encoder_apply_f32")
    if out_split is not None:
        _range(out_split, "encoder_apply")
    return dbg


# ---------------------------------------------------------------- fused encoder layer, d_model 256 (csrc/encoder256.hip)
ENC256_C, ENC256_KV_IMAGE, ENC256_NSLAB = 256, 32 * 1024 + 1024, 128


def _kslots16(kstep: int) -> torch.Tensor:
    """Column indices [4 (lane group g), 8 (slot j)] of a v_mfma_f32_16x16x32_f16 operand fragment for 32-wide k-step
    ``kstep`` in the order encoder256.hip chains accumulators into operands: two consecutive 16-channel accumulator blocks
    (lane group g holds rows 4 g + r) become one k-step, slot (g, j) = channel 32 kstep + 16 (j >> 2) + 4 g + (j & 3)."""
    j = torch.arange(8)
    g = torch.arange(4)[:, None]
    return 32 * kstep + 16 * (j >> 2) + 4 * g + (j & 3)


def _fragment16(plane: torch.Tensor, row0: int, kstep: int) -> torch.Tensor:
    """One 1-KB A fragment [64 lanes, 8 halves] of rows row0..row0+15: lane l = (row l & 15, group l >> 4)."""
    cols = _kslots16(kstep).to(plane.device)                            # [4, 8]
    rows = plane[row0:row0 + 16]                                        # [16, K]
    return torch.stack([rows[:, cols[g]] for g in range(4)], 0).reshape(64, 8)


def _fragment16_nat(plane: torch.Tensor, row0: int, kstep: int) -> torch.Tensor:
    """The same with the k columns in natural order (lane (i, g) slot j = column 32 kstep + 8 g + j): the B fragments of
    enc256_kv_kernel, whose A operand (the x tile) is read from memory in natural order as well."""
    rows = plane[row0:row0 + 16, 32 * kstep:32 * kstep + 32]          # [16, 32]
    return rows.reshape(16, 4, 8).permute(1, 0, 2).reshape(64, 8)


class Encoder256Weights:
    """Weights of one LoFTREncoderLayer with d_model 256, 8 heads (the coarse transformer) as the fragment stream of
    csrc/encoder256.hip: 128 slabs of 16 KB = 16 fragments of 1 KB (64 lanes x 16 bytes, lane-linear), (hi, lo) pairs in
    the order the kernel consumes them:
      q      16 slabs: k-step ks (0..7) x row half nb2 (0..1): rows 16 (8 nb2 + b), b = 0..7
      merge  16 slabs: the same
      per 64-channel chunk hc (0..7) of the MLP's hidden layer:
        mlp.0   8 slabs: slab u = k-steps 2u, 2u+1 of the 512 input columns [x | norm1(message)] x rows 64 hc + 16 b, b = 0..3
        mlp.2   4 slabs: k-step 2 hc + t (t = 0..1) of its 512 columns x row half nb2: rows 16 (8 nb2 + b), b = 0..7
    k columns inside a k-step are in ``_kslots16`` order.
    ``kv_stream`` (when wk / wv are given; enc256_kv_kernel): 32 slabs, head h = slabs 4 h .. 4 h + 3, slab u = k-steps 2 u,
    2 u + 1 x rows [W_k rows 32 h + 16 b (b = 0, 1) | W_v rows 32 h + 16 b], natural k order; fragment order (ks, block, plane)."""

    def __init__(self, wq, wmerge, w1, w2, n1, n2, nhead=8, wk=None, wv=None):
        C = ENC256_C
        if nhead != 8 or tuple(wq.shape) != (C, C) or tuple(wmerge.shape) != (C, C) or tuple(w1.shape) != (2 * C, 2 * C) or \
                tuple(w2.shape) != (C, 2 * C):
            raise _lib.DfsfmError("Encoder256Weights: the fused layer is built for d_model 256, 8 heads")
        dev = wq.device
        planes = {k: _split_planes(v.detach().cpu()) for k, v in (("q", wq), ("m", wmerge), ("1", w1), ("2", w2))}
        self.values = {n: (planes[k][0].float() + planes[k][1].float() / 2048.0).to(dev) for n, k in
                       (("q_proj.weight", "q"), ("merge.weight", "m"), ("mlp.0.weight", "1"), ("mlp.2.weight", "2"))}

        def pair(name, row0, kstep):
            return [_fragment16(planes[name][0], row0, kstep), _fragment16(planes[name][1], row0, kstep)]
        frags = []
        for name in ("q", "m"):
            for ks in range(8):
                for nb2 in range(2):
                    for b in range(8):
                        frags += pair(name, 16 * (8 * nb2 + b), ks)
        for hc in range(8):
            for u in range(8):
                for ks in range(2):
                    for b in range(4):
                        frags += pair("1", 64 * hc + 16 * b, 2 * u + ks)
            for t in range(2):
                for nb2 in range(2):
                    for b in range(8):
                        frags += pair("2", 16 * (8 * nb2 + b), 2 * hc + t)
        self.stream = torch.stack(frags, 0).contiguous().to(dev)            # [2048, 64, 8] fp16 = 128 slabs
        assert self.stream.numel() * 2 == ENC256_NSLAB * ENC_SLAB
        self.kv_stream = None
        if wk is not None:
            if tuple(wk.shape) != (C, C) or tuple(wv.shape) != (C, C):
                raise _lib.DfsfmError("Encoder256Weights: k_proj / v_proj must be [256, 256]")
            pk, pv = _split_planes(wk.detach().cpu()), _split_planes(wv.detach().cpu())
            self.values["k_proj.weight"] = (pk[0].float() + pk[1].float() / 2048.0).to(dev)
            self.values["v_proj.weight"] = (pv[0].float() + pv[1].float() / 2048.0).to(dev)
            frags = []
            for h in range(8):
                for u in range(4):
                    for ks in range(2):
                        for b in range(4):
                            pl, r0 = (pk, 32 * h + 16 * b) if b < 2 else (pv, 32 * h + 16 * (b - 2))
                            frags += [_fragment16_nat(pl[0], r0, 2 * u + ks), _fragment16_nat(pl[1], r0, 2 * u + ks)]
            self.kv_stream = torch.stack(frags, 0).contiguous().to(dev)      # [512, 64, 8] fp16 = 32 slabs
            assert self.kv_stream.numel() * 2 == 32 * ENC_SLAB
        self.n1 = tuple(t.detach().float().contiguous() for t in n1)
        self.n2 = tuple(t.detach().float().contiguous() for t in n2)
        self.values.update({"norm1.weight": self.n1[0], "norm1.bias": self.n1[1], "norm2.weight": self.n2[0],
                            "norm2.bias": self.n2[1]})


@_on_device
def encoder256_state(k, v, kv_mask=None, kv_group=1):
    """Source side of a d_model-256 fused layer application: k, v [N, S, 256] fp32 (column slices of the k|v projection:
    stride(-1) = 1, uniform row stride) -> the per-sequence attention state for ``encoder256_apply`` [N, ENC256_KV_IMAGE]
    bytes (KV^T per head as MFMA fragments + Ksum)."""
    _require_cuda(k, v)
    if k.dim() != 3 or k.shape != v.shape or k.shape[-1] != ENC256_C or k.dtype != torch.float32 or v.dtype != torch.float32:
        raise _lib.DfsfmError("encoder256_state: need fp32 k, v [N, S, 256]")
    N, S, _ = k.shape
    _, ldk = _rows_ld(k)
    _, ldv = _rows_ld(v)
    km = _as_u8(kv_mask)
    if km is not None and km.shape != (N, (S + kv_group - 1) // kv_group):
        raise _lib.DfsfmError("encoder256_state: kv_mask must be [N, ceil(S / kv_group)]")
    lib = _lib.lib()
    ws = _workspace(lib.dfsfm_encoder256_state_workspace(N, S), k.device)
    img = torch.empty((N, ENC256_KV_IMAGE), dtype=torch.uint8, device=k.device)
    rc = lib.dfsfm_encoder256_state_f32(_ptr(k), _ptr(v), ldk, ldv, _ptr(km), int(kv_group), N, S, _ptr(img), _ptr(ws),
                                        ws.numel(), _stream())
    _lib.check(rc, "dfsfm_encoder256_state_f32")
    return img


@_on_device
def encoder256_kv(src: "SplitAct", fw: Encoder256Weights, kv_mask=None, kv_group=1):
    """Source side of a d_model-256 fused layer application in ONE projection launch: source tokens [N, S, 256] (split planes)
    -> k | v = W_kv x (never stored) -> phi(K)^T V / S and sum phi(K) per head -> the attention state for ``encoder256_apply``
    (the same image ``encoder256_state`` builds from a materialised k | v)."""
    _require_cuda(src.hi)
    if fw.kv_stream is None:
        raise _lib.DfsfmError("encoder256_kv: these weights were packed without k_proj / v_proj")
    if src.hi.dim() != 3 or src.hi.shape[-1] != ENC256_C or src.hi.dtype != torch.float16 or src.lo.stride() != src.hi.stride():
        raise _lib.DfsfmError("encoder256_kv: need split planes [N, S, 256]")
    N, S = src.hi.shape[0], src.hi.shape[1]
    _, ld = _rows_ld(src.hi, torch.float16)
    km = _as_u8(kv_mask)
    if km is not None and km.shape != (N, (S + kv_group - 1) // kv_group):
        raise _lib.DfsfmError("encoder256_kv: kv_mask must be [N, ceil(S / kv_group)]")
    lib = _lib.lib()
    ws = _workspace(lib.dfsfm_encoder256_kv_workspace(N, S), src.hi.device)
    img = torch.empty((N, ENC256_KV_IMAGE), dtype=torch.uint8, device=src.hi.device)
    rc = lib.dfsfm_encoder256_kv_f32(_ptr(src.hi), _ptr(src.lo), ld, N, S, _ptr(fw.kv_stream), _ptr(km), int(kv_group), _ptr(img),
                                     _ptr(ws), ws.numel(), _stream())
    _lib.check(rc, "dfsfm_encoder256_kv_f32")
    return img


@_on_device
def encoder256_apply(x: "SplitAct", fw: Encoder256Weights, kv_image, S, q_mask=None, q_group=1, out_split=None, out=None,
                     eps=1e-5, attn_eps=1e-6, debug_stage=0):
    """Query side: x tokens [N, L, 256] (split planes) + attention state -> x + norm2(mlp([x | norm1(merge(attention))])) into
    the SplitAct view ``out_split`` and / or the fp32 view ``out`` ([N, L, 256] each) in ONE launch.  debug_stage != 0
    additionally returns an fp32 [N*L, 256] dump of that intermediate (tests)."""
    _require_cuda(x.hi, kv_image)

    def rows_of(t, what):
        if t.hi.dim() != 3 or t.hi.shape[-1] != ENC256_C or t.hi.dtype != torch.float16 or t.lo.stride() != t.hi.stride():
            raise _lib.DfsfmError(f"{what}: need split planes [N, L, 256]")
        _, ld = _rows_ld(t.hi, torch.float16)
        return t.hi.shape[0], t.hi.shape[1], ld
    N, L, ldx = rows_of(x, "encoder256_apply")
    if kv_image.shape != (N, ENC256_KV_IMAGE) or kv_image.dtype != torch.uint8 or not kv_image.is_contiguous():
        raise _lib.DfsfmError("encoder256_apply: kv_image must come from encoder256_state for the same N")
    oh = ol = None
    ldo = ldo32 = 0
    if out_split is not None:
        n2, l2, ldo = rows_of(out_split, "encoder256_apply(out_split)")
        if (n2, l2) != (N, L):
            raise _lib.DfsfmError("encoder256_apply: out_split shape mismatch")
        oh, ol = out_split.hi, out_split.lo
    if out is not None:
        rows_o, ldo32 = _rows_ld(out)
        if rows_o != N * L or out.shape[-1] != ENC256_C:
            raise _lib.DfsfmError("encoder256_apply: out shape mismatch")
    qm = _as_u8(q_mask)
    if qm is not None and qm.shape != (N, (L + q_group - 1) // q_group):
        raise _lib.DfsfmError("encoder256_apply: q_mask must be [N, ceil(L / q_group)]")
    # debug_stage 100: stage time stamps of wave 0 of every tile (tools/bench_enc256.py profile)
    dbg = torch.zeros((N * L, ENC256_C), dtype=torch.float32, device=x.hi.device) if debug_stage else None
    rc = _lib.lib().dfsfm_encoder256_apply_f32(_ptr(x.hi), _ptr(x.lo), ldx, N, L, int(S), _ptr(fw.stream), _ptr(kv_image),
                                               _ptr(qm), int(q_group), _ptr(fw.n1[0]), _ptr(fw.n1[1]), float(eps),
                                               _ptr(fw.n2[0]), _ptr(fw.n2[1]), float(eps), float(attn_eps), _ptr(oh), _ptr(ol),
                                               ldo, _ptr(out), ldo32, _ptr(dbg), int(debug_stage), _stream())
    _lib.check(rc, "dfsfm_encoder256_apply_f32")
    if out_split is not None:
        _range(out_split, "encoder256_apply")
    return dbg


@_on_device
def merge_keypoints(rows, img0, img1, n_images):
    """Scene-wide keypoint merge + match re-indexing (coarse_match.py:203-237 on the device).
    rows [M,5] fp32 (x0,y0,x1,y1,conf), img0/img1 [M] image index of each side (device tensors).
    Returns (kpts [K,2] fp32, scores [K] fp32, offsets [n_images+1] int64, match_ids [M,2] int64): image i owns
    keypoints offsets[i]:offsets[i+1] in id order."""
    _require_cuda(rows, img0, img1)
    rows = rows.to(torch.float32).contiguous().view(-1, 5)
    img0 = img0.to(torch.int32).contiguous()
    img1 = img1.to(torch.int32).contiguous()
    M = rows.shape[0]
    if img0.numel() != M or img1.numel() != M:
        raise _lib.DfsfmError("merge_keypoints: img0/img1 must have one entry per match row")
    dev = rows.device
    lib = _lib.lib()
    ws = _workspace(lib.dfsfm_merge_keypoints_workspace(M), dev)
    kpts = torch.empty((2 * M, 2), dtype=torch.float32, device=dev)
    scores = torch.empty((2 * M,), dtype=torch.float32, device=dev)
    offsets = torch.empty((n_images + 1,), dtype=torch.int64, device=dev)
    ids = torch.empty((M, 2), dtype=torch.int64, device=dev)
    nk = torch.zeros((1,), dtype=torch.int64, device=dev)
    status = torch.zeros((1,), dtype=torch.int32, device=dev)
    rc = lib.dfsfm_merge_keypoints(_ptr(rows), _ptr(img0), _ptr(img1), M, int(n_images), _ptr(kpts), _ptr(scores),
                                   _ptr(offsets), _ptr(ids), _ptr(nk), _ptr(status), _ptr(ws), ws.numel(), _stream())
    _lib.check(rc, "dfsfm_merge_keypoints")
    K, st = int(nk.item()), int(status.item())
    if st != 0:
        raise _lib.DfsfmError("merge_keypoints: image index or keypoint coordinate out of range")
    return kpts[:K], scores[:K], offsets, ids


@_on_device
class S2dFrontWeights:
    """Weights of S2DNet's conv1_1 / conv1_2 in the layout dfsfm_s2d_front_f32 reads: conv1_1 as MFMA A fragments of its split
    planes -- fp16 [2 halves of 32 channels][2 blocks of 16][hi, lo][64 lanes][8], element j of lane l = w1[32 half + 16 block +
    (l & 15)][k = 8 (l >> 4) + j], k = 3 (3 ky + kx) + ci, zero for k >= 27 -- conv1_2 as the tap-padded split planes of
    ``PackedDense`` (k = (ky*3 + kx)*64 + ci)."""

    def __init__(self, w1, b1, w2, b2):
        if tuple(w1.shape) != (64, 3, 3, 3) or tuple(w2.shape) != (64, 64, 3, 3):
            raise _lib.DfsfmError("S2dFrontWeights: conv1_1 [64,3,3,3] and conv1_2 [64,64,3,3] expected")
        wk = torch.zeros((64, 32), dtype=torch.float32, device=w1.device)
        wk[:, :27] = w1.detach().float().permute(0, 2, 3, 1).reshape(64, 27)                 # [co][k = (ky, kx, ci)]
        if not bool(torch.isfinite(wk).all()) or float(wk.abs().max()) >= FP16_MAX:
            raise _lib.DfsfmError(f"S2dFrontWeights: weights must be finite with |w| < {FP16_MAX:.0f} (split-plane range)")
        hi = wk.half()
        lo = ((wk - hi.float()) * 2048.0).half()
        planes = torch.stack([hi, lo], 0).reshape(2, 2, 2, 16, 4, 8)                        # [plane][half][block][channel][kslot][j]
        self.w1f = planes.permute(1, 2, 0, 4, 3, 5).reshape(2, 2, 2, 64, 8).contiguous()    # lane = channel + 16 kslot
        self.b1 = b1.detach().float().contiguous()
        self.conv2 = PackedDense(w2, b2, cin_pad=64, tap_padded=True)


SUPPORTED_S2D_FRONT_PATCH = 35


@_on_device
def s2d_front(patches, fw: S2dFrontWeights, c0: int, c1: int):
    """K9 front end in one launch (csrc/s2d_front.hip): patches fp32 [n, 35, 35, 3] (normalised NHWC) ->
    (relu1_2[:, c0:c1, c0:c1] as SplitAct [n, c1-c0, c1-c0, 64], MaxPool2d(3, 2, 1)(relu1_2) as SplitAct [n, 18, 18, 64])."""
    _require_cuda(patches)
    n, P, P2, C = patches.shape
    if P != P2 or C != 3 or patches.dtype != torch.float32 or not patches.is_contiguous():
        raise _lib.DfsfmError("s2d_front: dense fp32 [n, P, P, 3] patches expected")
    dev = patches.device
    if fw.w1f.device != dev:
        raise _lib.DfsfmError("s2d_front: weights on another device")
    crop = SplitAct.empty(n, c1 - c0, c1 - c0, 64, dev)
    pool = SplitAct.empty(n, (P + 1) // 2, (P + 1) // 2, 64, dev)
    pw = fw.conv2
    rc = _lib.lib().dfsfm_s2d_front_f32(_ptr(patches), n, P, _ptr(fw.w1f), _ptr(fw.b1), _ptr(pw.hi), _ptr(pw.lo), pw.hi.shape[0],
                                        pw.Kpad, _ptr(pw.bias), c0, c1, _ptr(crop.hi), _ptr(crop.lo), _ptr(pool.hi), _ptr(pool.lo),
                                        _stream())
    _lib.check(rc, "dfsfm_s2d_front_f32")
    _range(crop, "s2d_front crop")
    _range(pool, "s2d_front pool")
    return crop, pool


def maxpool3x3s2_nhwc(x):
    """nn.MaxPool2d(3, 2, 1) on a contiguous NHWC tensor (fp32) or SplitAct."""
    if isinstance(x, SplitAct):
        _require_cuda(x.hi)
        if not (x.hi.is_contiguous() and x.lo.is_contiguous()):
            raise _lib.DfsfmError("maxpool: split input must be dense")
        N, H, W, Cp = x.hi.shape
        y = SplitAct(torch.empty((N, (H - 1) // 2 + 1, (W - 1) // 2 + 1, Cp), dtype=torch.float16, device=x.hi.device),
                     torch.empty((N, (H - 1) // 2 + 1, (W - 1) // 2 + 1, Cp), dtype=torch.float16, device=x.hi.device), x.C)
        rc = _lib.lib().dfsfm_maxpool3x3s2_nhwc_f32(None, _ptr(x.hi), _ptr(x.lo), N, H, W, Cp, None, _ptr(y.hi),
                                                    _ptr(y.lo), _stream())
        _lib.check(rc, "dfsfm_maxpool3x3s2_nhwc_f32")
        return y
    _require_cuda(x)
    x = x.contiguous()
    N, H, W, C = x.shape
    out = torch.empty((N, (H - 1) // 2 + 1, (W - 1) // 2 + 1, C), dtype=torch.float32, device=x.device)
    rc = _lib.lib().dfsfm_maxpool3x3s2_nhwc_f32(_ptr(x), None, None, N, H, W, C, _ptr(out), None, None, _stream())
    _lib.check(rc, "dfsfm_maxpool3x3s2_nhwc_f32")
    return out
