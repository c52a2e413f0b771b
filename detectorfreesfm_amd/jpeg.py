"""Baseline JPEG decode on the device: the decode in front of ``images.read_grayscale`` / ``read_rgb``.

The reference decodes every frame on the host with ``cv2.imread(path, IMREAD_GRAYSCALE)`` (src/dataset/utils.py:127, 183) or
``cv2.imread(path, IMREAD_COLOR)`` + BGR2RGB (:86-92), i.e. libjpeg-turbo with its defaults.  ``decode(buf, color, device)``
returns the same bytes from ``dfsfm_jpeg_decode_u8`` (csrc/jpeg_decode.hip): the host parses the marker segments (tables, frame
header, restart positions: ``plan``), the entropy decode, the inverse DCT, the chroma upsampling and the colour conversion
run on the GPU.  EXIF orientation is applied as cv2.imread does (index bookkeeping on the decoded bytes).

What the device path takes: baseline / extended-sequential Huffman files (SOF0, SOF1), 8 bit, one interleaved scan (or, r06, one
scan per component: ``plan_components``), grey or YCbCr with 4:4:4 / 4:2:2 / 4:2:0 sampling -- what cameras and ``cv2.imwrite`` / Pillow write by default, and all eight frames of
the reference's example scene -- and (r06) 4:4:0 / 4:1:1 (what a lossless 90-degree rotation of a 4:2:2 file and DV-derived
material carry; libjpeg-turbo's h1v2 fancy filter / plain replication), with or without restart markers.  Everything else
(progressive, arithmetic, 12 bit, CMYK / RGB colour spaces, other sampling ratios, partly interleaved multi-scan files) raises
``UnsupportedJpeg``;
``images._decode`` then falls back to the host decoder the reference itself uses.
"""
import ctypes
from dataclasses import dataclass, field
from functools import lru_cache
from typing import List, Optional, Tuple

import numpy as np

CHUNK_BYTES = 128            # bytes of the compacted scan per decoder thread
DEFAULT_SWEEPS = 4           # sweep launches per call before the first status check (each runs rounds inside its workgroups)
SWEEP_WG = 256               # csrc/jpeg_core.h: chunks per workgroup of the sweep kernel
UNSTUFF_BLOCK = 4096         # csrc/jpeg_core.h

ZIGZAG = np.array([0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14,
                   21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60,
                   61, 54, 47, 55, 62, 63], dtype=np.int64)


class UnsupportedJpeg(Exception):
    """Not a file the device decoder takes (the message says why)."""


class CorruptJpeg(Exception):
    pass


class Frame(ctypes.Structure):
    """``dfsfm_jpeg_frame`` of include/dfsfm_hip.h."""
    _fields_ = [("width", ctypes.c_int32), ("height", ctypes.c_int32), ("ncomp", ctypes.c_int32),
                ("h", ctypes.c_int32 * 3), ("v", ctypes.c_int32 * 3),
                ("dc_slot", ctypes.c_int32 * 3), ("ac_slot", ctypes.c_int32 * 3),
                ("restart", ctypes.c_int32), ("nseg", ctypes.c_int32), ("nchunks", ctypes.c_int32),
                ("chunk_bytes", ctypes.c_int32)]


@dataclass
class Plan:
    """Everything ``dfsfm_jpeg_decode_u8`` takes besides the output, as host arrays."""
    frame: Frame
    scan: np.ndarray                 # uint8: the entropy-coded bytes of the scan as in the file
    tab_key: bytes                   # the DHT payloads of the scan's tables, in slot order
    tab: np.ndarray                  # uint32 [4 * 596]: huff_tab of include/dfsfm_hip.h
    qt: np.ndarray                   # uint16 [3, 64] natural order
    block_base: np.ndarray           # uint32 [ceil(len(scan) / 4096)] entropy bytes in front of each block of the scan
    seg_beg: np.ndarray              # uint32 [nseg] byte range of each restart interval in the compacted scan
    seg_end: np.ndarray
    seg_chunk0: np.ndarray           # int32 [nseg]
    chunk_seg: np.ndarray            # int32 [nchunks]
    orientation: int = 1
    sampling: List[Tuple[int, int]] = field(default_factory=list)
    launch_bound: int = 0            # sweep launches that guarantee the fixed point (flat_launch_bound)

    @property
    def width(self):
        return int(self.frame.width)

    @property
    def height(self):
        return int(self.frame.height)


TAB_SLOT_BYTES = 2384        # csrc/jpeg_core.h: uint16 l1[1024] | uint32 long[6][3] | uint8 vals[256], padded to 16


def huffman_table(bits: bytes, vals: bytes) -> np.ndarray:
    """One slot of ``huff_tab`` (include/dfsfm_hip.h) for the code ITU T.81 Annex C generates from BITS / HUFFVAL: the direct
    table of the 10-bit prefixes and, for the lengths 11..16, (limit, first code, index of its first symbol) in left-aligned
    16-bit code space -- codes of one length are consecutive and longer codes follow numerically."""
    slot = np.zeros(TAB_SLOT_BYTES, dtype=np.uint8)
    l1 = slot[:2048].view(np.uint16)
    lg = slot[2048:2048 + 72].view(np.uint32)
    code, k = 0, 0
    for length in range(1, 17):
        n = bits[length - 1]
        if k + n > len(vals) or code + n > (1 << length):
            raise CorruptJpeg("bad Huffman table")
        if length <= 10:
            for j in range(n):
                lo = (code + j) << (10 - length)
                l1[lo:lo + (1 << (10 - length))] = (length << 8) | vals[k + j]
        else:
            lg[3 * (length - 11):3 * (length - 11) + 3] = ((code + n) << (16 - length), code << (16 - length), k)
        code = (code + n) << 1
        k += n
    slot[2048 + 72:2048 + 72 + len(vals)] = np.frombuffer(vals, dtype=np.uint8)
    return slot


def huffman_decode_prefix(slot: np.ndarray, prefix16: int) -> int:
    """The device's lookup (csrc/jpeg_core.h huff_lookup) restated: (length << 8) | symbol of the code that starts the 16 bits, 0
    if none does."""
    e = int(slot[:2048].view(np.uint16)[prefix16 >> 6])
    if e == 0:
        lg = slot[2048:2048 + 72].view(np.uint32)
        for length in range(11, 17):
            limit, base, off = (int(v) for v in lg[3 * (length - 11):3 * (length - 11) + 3])
            if prefix16 < limit:
                return (length << 8) | int(slot[2048 + 72 + off + ((prefix16 - base) >> (16 - length))])
    return e


@lru_cache(maxsize=32)
def _tab_block(key: bytes) -> np.ndarray:
    """uint32 [4 * 596]: the slots of up to four (bits, vals) tables serialised in ``key``."""
    out = np.zeros((4, TAB_SLOT_BYTES), dtype=np.uint8)
    o, slot = 0, 0
    while o < len(key):
        n = sum(key[o:o + 16])
        out[slot] = huffman_table(key[o:o + 16], key[o + 16:o + 16 + n])
        o += 16 + n
        slot += 1
    return out.reshape(-1).view(np.uint32)


def _exif_orientation(seg: bytes) -> int:
    t = seg[6:]
    if len(t) < 8 or t[:2] not in (b"II", b"MM"):
        return 1
    bo = "little" if t[:2] == b"II" else "big"
    ifd = int.from_bytes(t[4:8], bo)
    if ifd + 2 > len(t):
        return 1
    n = int.from_bytes(t[ifd:ifd + 2], bo)
    for e in range(n):
        o = ifd + 2 + 12 * e
        if o + 12 > len(t):
            break
        if int.from_bytes(t[o:o + 2], bo) == 0x0112:
            v = int.from_bytes(t[o + 8:o + 10], bo)
            return v if 1 <= v <= 8 else 1
    return 1


class MultiScanJpeg(UnsupportedJpeg):
    """A sequential file whose components come in separate scans: not one ``Plan`` -- ``plan_components`` takes it."""


def plan(buf, chunk_bytes: int = CHUNK_BYTES) -> Plan:
    """Parse the marker segments of a JPEG file (bytes / uint8 array) and cut its scan into decoder chunks.  Raises
    ``UnsupportedJpeg`` / ``CorruptJpeg`` only: a damaged header never surfaces as an IndexError."""
    try:
        return _plan(buf, chunk_bytes)
    except (IndexError, ValueError, OverflowError) as e:
        raise CorruptJpeg(f"damaged marker segment ({type(e).__name__}: {e})") from None


@dataclass
class ComponentPlans:
    """A multi-scan sequential file (T.81 A.2.2: each component in a scan of its own, one block per MCU): one grey-frame ``Plan`` per
    component, ``plans[c].width x .height`` = the component's real samples, + what the colour stage needs."""
    width: int
    height: int
    sampling: List[Tuple[int, int]]
    orientation: int
    plans: List[Plan]


def plan_components(buf, chunk_bytes: int = CHUNK_BYTES) -> ComponentPlans:
    try:
        return _plan_components(buf, chunk_bytes)
    except (IndexError, ValueError, OverflowError) as e:
        raise CorruptJpeg(f"damaged marker segment ({type(e).__name__}: {e})") from None


class _Header:
    """What the marker segments in front of a scan have said so far."""

    def __init__(self):
        self.qts, self.huff = {}, {}             # id -> natural-order steps; (class, id) -> (bits, vals)
        self.sof, self.restart, self.orientation, self.jfif, self.adobe = None, 0, 1, False, None


def _as_bytes(buf):
    data = np.frombuffer(buf, dtype=np.uint8) if isinstance(buf, (bytes, bytearray, memoryview)) else np.ascontiguousarray(buf, dtype=np.uint8)
    b = data.tobytes() if not isinstance(buf, bytes) else buf
    if len(b) < 4 or b[0] != 0xFF or b[1] != 0xD8:
        raise CorruptJpeg("no SOI marker")
    return data, b


def _walk(b, p, st):
    """Marker segments from offset p up to and including the next SOS header: (SOS payload, offset of the entropy-coded data)."""
    n = len(b)
    while p + 4 <= n:
        if b[p] != 0xFF:
            raise CorruptJpeg("marker expected")
        while p < n and b[p] == 0xFF:
            p += 1
        m = b[p]
        p += 1
        if m == 0xD8 or 0xD0 <= m <= 0xD7 or m == 0x01:
            continue
        if m == 0xD9:
            raise CorruptJpeg("EOI before SOS")
        ln = (b[p] << 8) | b[p + 1]
        if ln < 2 or p + ln > n:
            raise CorruptJpeg("truncated marker segment")
        seg = b[p + 2:p + ln]
        if m == 0xDB:
            i = 0
            while i < len(seg):
                pq, tq = seg[i] >> 4, seg[i] & 15
                i += 1
                if pq:
                    # 16-bit steps: coefficient x step can leave the int32 range of the device's dequantisation / IDCT (libjpeg's
                    # JLONG is 64 bits wide) -- a file for the host decoder
                    raise UnsupportedJpeg("16-bit quantisation table")
                else:
                    q = np.frombuffer(seg[i:i + 64], dtype=np.uint8).astype(np.uint16)
                    i += 64
                if q.size != 64 or tq > 3:
                    raise CorruptJpeg("bad DQT")
                nat = np.zeros(64, dtype=np.uint16)
                nat[ZIGZAG] = q
                st.qts[tq] = nat
        elif m == 0xC4:
            i = 0
            while i < len(seg):
                tc, th = seg[i] >> 4, seg[i] & 15
                bits = seg[i + 1:i + 17]
                cnt = sum(bits)
                vals = seg[i + 17:i + 17 + cnt]
                if len(bits) != 16 or len(vals) != cnt or tc > 1 or th > 3:
                    raise CorruptJpeg("bad DHT")
                st.huff[(tc, th)] = (bytes(bits), bytes(vals))
                i += 17 + cnt
        elif m in (0xC0, 0xC1, 0xC2):
            if seg[0] != 8:
                raise UnsupportedJpeg(f"{seg[0]}-bit samples")
            st.sof = dict(progressive=m == 0xC2, height=(seg[1] << 8) | seg[2], width=(seg[3] << 8) | seg[4], ncomp=seg[5],
                          comps=[(seg[6 + 3 * c], seg[7 + 3 * c] >> 4, seg[7 + 3 * c] & 15, seg[8 + 3 * c]) for c in range(seg[5])])
        elif 0xC3 <= m <= 0xCF and m not in (0xC4, 0xC8, 0xCC):
            raise UnsupportedJpeg("lossless / differential / arithmetic-coded JPEG")
        elif m == 0xDD:
            st.restart = (seg[0] << 8) | seg[1]
        elif m == 0xE0 and seg[:5] == b"JFIF\0":
            st.jfif = True
        elif m == 0xEE and seg[:5] == b"Adobe" and len(seg) >= 12:
            st.adobe = seg[11]
        elif m == 0xE1 and seg[:6] == b"Exif\0\0":
            st.orientation = _exif_orientation(seg)
        elif m == 0xDA:
            return seg, p + ln
        p += ln
    raise CorruptJpeg("no frame / scan header")


def _check_frame(st, sos):
    sof = st.sof
    if sof is None:
        raise CorruptJpeg("no frame / scan header")
    if sof["progressive"]:
        raise UnsupportedJpeg("progressive JPEG")
    ncomp = sof["ncomp"]
    if ncomp not in (1, 3):
        raise UnsupportedJpeg(f"{ncomp} components")
    if sof["width"] == 0 or sof["height"] == 0:
        raise UnsupportedJpeg("frame size given by a DNL marker")
    if sof["width"] * sof["height"] > (1 << 28):
        raise UnsupportedJpeg("frame larger than 2^28 pixels")
    sampling = [(h, v) for (_, h, v, _) in sof["comps"]]
    if ncomp == 3:
        ids = [c[0] for c in sof["comps"]]
        ycc = True                   # jdapimin.c default_decompress_parms
        if not st.jfif and st.adobe == 0:
            ycc = False
        if not st.jfif and st.adobe is None and ids == [ord("R"), ord("G"), ord("B")]:
            ycc = False
        if not ycc:
            raise UnsupportedJpeg("RGB-coded JPEG")
        if sampling[1] != (1, 1) or sampling[2] != (1, 1) or sampling[0] not in ((1, 1), (2, 1), (2, 2), (1, 2), (4, 1)):
            raise UnsupportedJpeg(f"sampling {sampling}")
    if sos[1 + 2 * sos[0]] != 0 or sos[2 + 2 * sos[0]] != 63:
        raise UnsupportedJpeg("spectral selection in a sequential frame")
    return sampling


def _scan_plan(st, data, scan_start, fr, qt, tables, nmcu, chunk_bytes, sampling) -> Plan:
    """The scan that starts at scan_start, as decoder chunks: up to the first marker that is not RSTn; restart markers cut it into
    segments."""
    restart = st.restart
    d = data[scan_start:]
    nseg = -(-nmcu // restart) if restart else 1
    scan_len, n_rst, block_base, seg_beg, seg_end = SCAN_INDEX(d, nseg)
    if restart:
        if n_rst != nseg - 1:
            raise CorruptJpeg(f"{n_rst} restart markers for {nseg} intervals")
    elif n_rst:
        raise CorruptJpeg("restart markers without DRI")
    if np.any(seg_end <= seg_beg):
        raise CorruptJpeg("empty restart interval")
    per = -(-(seg_end - seg_beg).astype(np.int64) // chunk_bytes)
    seg_chunk0 = np.concatenate([[0], np.cumsum(per)[:-1]]).astype(np.int32)
    chunk_seg = np.repeat(np.arange(nseg, dtype=np.int32), per)
    fr.restart, fr.nseg, fr.nchunks, fr.chunk_bytes = restart, nseg, int(chunk_seg.size), chunk_bytes
    key = b"".join(st.huff[t][0] + st.huff[t][1] for t in tables)
    padded = np.zeros((scan_len + 31) // 16 * 16, dtype=np.uint8)    # the compaction pass loads aligned 16-byte pieces
    padded[:scan_len] = d[:scan_len]
    return Plan(frame=fr, scan=padded[:scan_len], tab_key=key, tab=_tab_block(key), qt=qt, block_base=block_base, seg_beg=seg_beg,
                seg_end=seg_end, seg_chunk0=seg_chunk0, chunk_seg=chunk_seg, orientation=st.orientation, sampling=sampling,
                launch_bound=flat_launch_bound(seg_chunk0, per))


def _plan(buf, chunk_bytes: int) -> Plan:
    data, b = _as_bytes(buf)
    st = _Header()
    sos, scan_start = _walk(b, 2, st)
    sampling = _check_frame(st, sos)
    sof = st.sof
    ncomp = sof["ncomp"]
    if sos[0] != ncomp:
        if sos[0] == 1 and ncomp == 3:
            raise MultiScanJpeg("multi-scan sequential JPEG (one component per scan): plan_components")
        raise UnsupportedJpeg("multi-scan sequential JPEG with partly interleaved scans")
    fr = Frame()
    fr.width, fr.height, fr.ncomp = sof["width"], sof["height"], ncomp
    qt = np.ones((3, 64), dtype=np.uint16)
    tables = []                      # distinct (class, id) in slot order
    for c in range(ncomp):
        cid, h, v, tq = sof["comps"][c]
        if sos[1 + 2 * c] != cid:
            raise UnsupportedJpeg("scan components out of frame order")
        td, ta = sos[2 + 2 * c] >> 4, sos[2 + 2 * c] & 15
        if tq not in st.qts or (0, td) not in st.huff or (1, ta) not in st.huff:
            raise CorruptJpeg("missing table")
        qt[c] = st.qts[tq]
        for key in ((0, td), (1, ta)):
            if key not in tables:
                tables.append(key)
        fr.dc_slot[c], fr.ac_slot[c] = tables.index((0, td)), tables.index((1, ta))
        fr.h[c], fr.v[c] = (1, 1) if ncomp == 1 else (h, v)
    if len(tables) > 4:
        raise UnsupportedJpeg("more than four Huffman tables in one scan")
    hmax, vmax = fr.h[0], fr.v[0]
    nmcu = (-(-fr.width // (8 * hmax))) * (-(-fr.height // (8 * vmax)))
    return _scan_plan(st, data, scan_start, fr, qt, tables, nmcu, chunk_bytes, sampling)


def _plan_components(buf, chunk_bytes: int) -> ComponentPlans:
    data, b = _as_bytes(buf)
    st = _Header()
    p, plans, sampling = 2, [None, None, None], None
    while any(pl is None for pl in plans):
        sos, scan_start = _walk(b, p, st)
        sampling = _check_frame(st, sos)
        sof = st.sof
        if sof["ncomp"] != 3 or sos[0] != 1:
            raise UnsupportedJpeg("not a one-component-per-scan sequential JPEG")
        ids = [c[0] for c in sof["comps"]]
        if sos[1] not in ids:
            raise CorruptJpeg("scan of an unknown component")
        c = ids.index(sos[1])
        if plans[c] is not None:
            raise CorruptJpeg("a component in two scans")
        _, h, v, tq = sof["comps"][c]
        td, ta = sos[2] >> 4, sos[2] & 15
        if tq not in st.qts or (0, td) not in st.huff or (1, ta) not in st.huff:
            raise CorruptJpeg("missing table")
        hmax, vmax = sampling[0]
        fr = Frame()                 # the component as a grey frame of its real samples: ceil(W h / hmax) x ceil(H v / vmax)
        fr.width, fr.height, fr.ncomp = -(-sof["width"] * h // hmax), -(-sof["height"] * v // vmax), 1
        fr.h[0] = fr.v[0] = 1
        fr.dc_slot[0], fr.ac_slot[0] = 0, 1
        qt = np.ones((3, 64), dtype=np.uint16)
        qt[0] = st.qts[tq]
        nmcu = (-(-fr.width // 8)) * (-(-fr.height // 8))
        plans[c] = _scan_plan(st, data, scan_start, fr, qt, [(0, td), (1, ta)], nmcu, chunk_bytes, [(1, 1)])
        p = scan_start + int(plans[c].scan.size)
    return ComponentPlans(width=st.sof["width"], height=st.sof["height"], sampling=sampling, orientation=st.orientation, plans=plans)


def flat_launch_bound(seg_chunk0, per) -> int:
    """Sweep launches after which the relaxation HAS reached its fixed point whatever the scan looks like: a launch settles every
    workgroup (SWEEP_WG consecutive chunks) whose first entry is final, a segment's first chunk is final from the start, so the
    truth crosses one workgroup boundary per launch at worst; + 1 for the launch that finds nothing to do."""
    first = np.asarray(seg_chunk0, dtype=np.int64)
    last = first + np.asarray(per, dtype=np.int64) - 1
    return int((last // SWEEP_WG - first // SWEEP_WG).max()) + 2


def _scan_index_numpy(d: np.ndarray, nseg_expected: int):
    """The index of an entropy-coded scan, as array passes -- the SPECIFICATION of ``dfsfm_jpeg_scan_index`` (tests hold the library
    function to it; nothing on the product path calls it): (scan_len, restart markers found, block_base, seg_beg, seg_end)."""
    ff = np.flatnonzero(d == 0xFF)
    nxt = np.where(ff + 1 < d.size, d[np.minimum(ff + 1, max(d.size - 1, 0))], 0xD9) if d.size else ff   # the byte after each FF (EOI past the end)
    marker = (nxt != 0) & (nxt != 0xFF)                         # FF 00 is a stuffed data byte, FF FF a fill byte
    ends = ff[marker & ~((nxt >= 0xD0) & (nxt <= 0xD7))]        # the first marker that is not RSTn ends the scan
    scan_len = int(ends[0]) if ends.size else d.size
    inside = ff < scan_len
    ff, nxt = ff[inside], nxt[inside]
    nxt = np.where(ff + 1 < scan_len, nxt, 0xD9)                # what follows the scan is a marker
    rst = ff[(nxt >= 0xD0) & (nxt <= 0xD7)]
    # bytes the device's compaction pass drops (csrc/jpeg_core.h unstuff_mask, the same three rules): the stuffed zero after a
    # data FF, the FF of a marker / fill byte (an FF not followed by 00), the code byte of a restart marker
    drops = np.sort(np.concatenate([ff[nxt == 0] + 1, ff[nxt != 0], rst + 1]))
    clean = lambda x: np.asarray(x, dtype=np.int64) - np.searchsorted(drops, x, side="left")
    nblocks = -(-scan_len // UNSTUFF_BLOCK)
    block_base = clean(np.arange(nblocks, dtype=np.int64) * UNSTUFF_BLOCK).astype(np.uint32)
    seg_beg = clean(np.concatenate([[0], rst + 2])).astype(np.uint32)
    seg_end = clean(np.concatenate([rst, [scan_len]])).astype(np.uint32)
    return scan_len, int(rst.size), block_base, seg_beg, seg_end


def _scan_index_lib(d: np.ndarray, nseg_expected: int):
    """The same through ``dfsfm_jpeg_scan_index`` (host code of the library, csrc/jpeg_decode.hip): one memchr walk outside the
    interpreter lock instead of ten array passes inside it."""
    from . import _lib
    d = np.ascontiguousarray(d)
    nblk = -(-d.size // UNSTUFF_BLOCK)
    block_base = np.empty(max(nblk, 1), dtype=np.uint32)
    seg_beg = np.empty(max(nseg_expected, 1), dtype=np.uint32)
    seg_end = np.empty(max(nseg_expected, 1), dtype=np.uint32)
    n_rst = ctypes.c_int64(0)
    scan_len = _lib.lib().dfsfm_jpeg_scan_index(d.ctypes.data, d.size, block_base.ctypes.data, block_base.size, seg_beg.ctypes.data,
                                                seg_end.ctypes.data, seg_beg.size, ctypes.byref(n_rst))
    if scan_len < 0:
        raise CorruptJpeg(f"dfsfm_jpeg_scan_index -> {scan_len}")
    n = int(n_rst.value)
    if n + 1 > seg_beg.size:             # more restart markers than intervals: the caller reports the counts
        return int(scan_len), n, block_base[:0], seg_beg[:0], seg_end[:0]
    return int(scan_len), n, block_base[:-(-int(scan_len) // UNSTUFF_BLOCK)], seg_beg[:n + 1], seg_end[:n + 1]


SCAN_INDEX = _scan_index_lib             # tests swap in _scan_index_numpy to hold the two to each other


def is_jpeg(buf) -> bool:
    return len(buf) >= 3 and buf[0] == 0xFF and buf[1] == 0xD8 and buf[2] == 0xFF


# ---------------------------------------------------------------------------------------------------------------------
# device side
# ---------------------------------------------------------------------------------------------------------------------
def apply_orientation(img, orientation: int):
    """EXIF orientation the way cv2.imread applies it (loadsave.cpp ExifTransform); ``img`` [H,W] or [H,W,3] tensor."""
    if orientation == 2:
        img = img.flip(1)
    elif orientation == 3:
        img = img.flip(0, 1)
    elif orientation == 4:
        img = img.flip(0)
    elif orientation == 5:
        img = img.transpose(0, 1)
    elif orientation == 6:
        img = img.transpose(0, 1).flip(1)
    elif orientation == 7:
        img = img.transpose(0, 1).flip(0, 1)
    elif orientation == 8:
        img = img.transpose(0, 1).flip(0)
    return img.contiguous()


def decode(buf, color: bool, device="cuda", sweeps: int = DEFAULT_SWEEPS, chunk_bytes: int = CHUNK_BYTES, orient: bool = True,
           return_info: bool = False):
    """The bytes of ``cv2.imread(IMREAD_GRAYSCALE)`` ([H,W] uint8, ``color=False``) / of ``cv2.imread(IMREAD_COLOR)`` after
    BGR2RGB ([H,W,3], ``color=True``) as a device tensor.  Raises ``UnsupportedJpeg`` for files outside the device path and
    ``CorruptJpeg`` for streams that do not decode; reads 16 bytes of status back once per call (more sweep launches follow only if
    the fixed point was not reached: photographs settle in two or three, a frame that is one flat colour needs one launch per 256
    chunks of its longest restart interval -- ``Plan.launch_bound`` -- and gets them)."""
    import torch
    from . import ops
    device = torch.device(device)
    try:
        pl = plan(buf, chunk_bytes)
    except MultiScanJpeg:
        pl = plan_components(buf, chunk_bytes)
        out, info = _decode_components(pl, color, device, sweeps)
    else:
        out, info = ops.jpeg_decode(pl, 3 if color else 1, device, sweeps)
    if orient and pl.orientation != 1:
        out = apply_orientation(out, pl.orientation)
    return (out, info) if return_info else out


def _decode_components(cp: ComponentPlans, color: bool, device, sweeps: int):
    """A multi-scan sequential file: the luma scan alone for grey output; else the three component scans as ONE batched call of grey
    frames, then the colour stage on the planes (``dfsfm_jpeg_ycc_planes_to_rgb_u8``)."""
    from . import ops
    if not color:
        return ops.jpeg_decode(cp.plans[0], 1, device, sweeps)
    planes = ops.jpeg_decode_batch_launch(cp.plans, 1, device, sweeps).finish()
    for r in planes:
        if isinstance(r, Exception):
            raise r
    return ops.jpeg_planes_to_rgb(planes[0], planes[1], planes[2], cp.width, cp.height, cp.sampling[0]), dict(calls=1, components=3)


def _plan_any(buf):
    """``plan``, or ``plan_components`` for a multi-scan sequential file."""
    try:
        return plan(buf)
    except MultiScanJpeg:
        return plan_components(buf)


def decode_batch(bufs, color: bool, device="cuda", sweeps: int = DEFAULT_SWEEPS, orient: bool = True, plans=None):
    """``decode`` for a list of files as ONE batched call (``dfsfm_jpeg_decode_batch_u8``: grid.y = file, one set of launches per seven
    files, one upload, one status read).  Returns the list of device tensors; a file that does not decode raises its error after
    the others are done (``UnsupportedJpeg`` is raised by ``plan`` before anything is uploaded)."""
    import torch
    from . import ops
    plans = [_plan_any(b) for b in bufs] if plans is None else plans
    if not plans:
        return []
    device = torch.device(device)
    one = [i for i, pl in enumerate(plans) if isinstance(pl, Plan)]
    res = [None] * len(plans)
    if one:
        for i, r in zip(one, ops.jpeg_decode_batch_launch([plans[i] for i in one], 3 if color else 1, device, sweeps).finish()):
            res[i] = r
    for i, pl in enumerate(plans):                   # multi-scan sequential files: a batched call of their own each
        if not isinstance(pl, Plan):
            res[i] = _decode_components(pl, color, device, sweeps)[0]
    for r in res:
        if isinstance(r, Exception):
            raise r
    return [apply_orientation(r, pl.orientation) if orient and pl.orientation != 1 else r for r, pl in zip(res, plans)]


def decode_many(bufs, color: bool, device="cuda", streams: int = 2, sweeps: int = DEFAULT_SWEEPS, orient: bool = True, batch: int = 14,
                workers: int = 2):
    """``decode`` for a list of files: batches of ``batch`` files (``decode_batch``: one set of launches per seven files) on
    ``streams`` side streams, while ``workers`` helper threads parse the marker segments of the next files (numpy's passes over a scan
    release the GIL).  r05 launched every file by itself (~27 launches each, four in flight) and was bound by the launching thread
    at 0.44 ms per file; a batch costs the launching thread ~15 launches per seven files.  At most ``streams`` batches are open,
    which bounds the memory held.  Returns the list of device tensors (usable on the current stream).  A file the device path
    does not take raises ``UnsupportedJpeg`` when its turn comes."""
    import torch
    from concurrent.futures import ThreadPoolExecutor
    from . import ops
    device = torch.device(device)
    cur = torch.cuda.current_stream(device)
    groups = [list(range(i, min(len(bufs), i + max(1, batch)))) for i in range(0, len(bufs), max(1, batch))]
    side = [torch.cuda.Stream(device) for _ in range(max(1, min(streams, len(groups))))]
    for s in side:
        s.wait_stream(cur)
    outs, open_calls = [None] * len(bufs), []

    def close_oldest():
        idx, plans, call = open_calls.pop(0)
        for i, pl, r in zip(idx, plans, call.finish()):
            if isinstance(r, Exception):
                raise r
            r.record_stream(cur)
            outs[i] = apply_orientation(r, pl.orientation) if orient and pl.orientation != 1 else r
    with ThreadPoolExecutor(max_workers=max(1, workers)) as pool:
        parsed = [pool.submit(_plan_any, b) for b in bufs]
        multi = []
        for gi, idx in enumerate(groups):
            got = [(i, parsed[i].result()) for i in idx]
            multi += [(i, pl) for i, pl in got if not isinstance(pl, Plan)]
            idx, plans = [i for i, pl in got if isinstance(pl, Plan)], [pl for _, pl in got if isinstance(pl, Plan)]
            if not plans:
                continue
            if len(open_calls) >= len(side):
                close_oldest()
            with torch.cuda.stream(side[gi % len(side)]):
                open_calls.append((idx, plans, ops.jpeg_decode_batch_launch(plans, 3 if color else 1, device, sweeps)))
    while open_calls:
        close_oldest()
    for s in side:
        cur.wait_stream(s)
    for i, cp in multi:                              # multi-scan sequential files (rare): one by one on the caller's stream
        r = _decode_components(cp, color, device, sweeps)[0]
        outs[i] = apply_orientation(r, cp.orientation) if orient and cp.orientation != 1 else r
    return outs
