"""A small baseline JPEG ENCODER for the tests: YCbCr frames with a luma sampling factor Pillow's encoder does not offer
(4:4:0 = 1x2, 4:1:1 = 4x1; also 1x1, 2x1, 2x2 to check the encoder itself), one interleaved scan, optional restart markers.
Test infrastructure only -- the files it writes are decoded by libjpeg-turbo (Pillow), by the oracle and by the device path, and
libjpeg-turbo's bytes are the truth.  Quantisation and Huffman tables are taken from a Pillow-written file of the same quality
(its DQT / DHT segments: the Annex K tables), so nothing here restates table constants."""
import io

import numpy as np
from PIL import Image
from scipy.fft import dctn

ZIGZAG = [0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35, 42,
          49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63]


def _segments(buf):
    p = 2
    while p < len(buf):
        assert buf[p] == 0xFF
        m = buf[p + 1]
        ln = (buf[p + 2] << 8) | buf[p + 3]
        yield m, buf[p + 4:p + 2 + ln]
        if m == 0xDA:
            return
        p += 2 + ln


def _tables_like_pillow(quality):
    b = io.BytesIO()
    Image.fromarray(np.zeros((8, 8, 3), np.uint8)).save(b, "JPEG", quality=quality, subsampling=0)
    raw = b.getvalue()
    dqt, dht, q, huff = b"", b"", {}, {}
    for m, seg in _segments(raw):
        if m == 0xDB:
            dqt += bytes([0xFF, 0xDB]) + (len(seg) + 2).to_bytes(2, "big") + seg
            p = 0
            while p < len(seg):
                assert seg[p] >> 4 == 0
                q[seg[p] & 15] = np.frombuffer(seg[p + 1:p + 65], np.uint8).astype(np.int64)      # zig-zag order
                p += 65
        elif m == 0xC4:
            dht += bytes([0xFF, 0xC4]) + (len(seg) + 2).to_bytes(2, "big") + seg
            p = 0
            while p < len(seg):
                tc, th = seg[p] >> 4, seg[p] & 15
                bits = list(seg[p + 1:p + 17])
                n = sum(bits)
                vals = list(seg[p + 17:p + 17 + n])
                code, k, enc = 0, 0, {}
                for ln in range(1, 17):
                    for _ in range(bits[ln - 1]):
                        enc[vals[k]] = (code, ln)
                        code += 1
                        k += 1
                    code <<= 1
                huff[(tc, th)] = enc
                p += 17 + n
    return dqt, dht, q, huff


class _Bits:
    def __init__(self):
        self.out, self.acc, self.n = bytearray(), 0, 0

    def put(self, code, ln):
        self.acc = (self.acc << ln) | (code & ((1 << ln) - 1))
        self.n += ln
        while self.n >= 8:
            byte = (self.acc >> (self.n - 8)) & 0xFF
            self.out.append(byte)
            if byte == 0xFF:
                self.out.append(0)
            self.n -= 8
        self.acc &= (1 << self.n) - 1

    def flush(self):
        if self.n:
            self.put((1 << (8 - self.n)) - 1, 8 - self.n)


def _magnitude(v):
    a = abs(v)
    s = a.bit_length()
    return s, (v if v >= 0 else v + (1 << s) - 1)


def encode(rgb, luma=(1, 2), quality=85, restart=0, per_component=False):
    """rgb [H, W, 3] uint8 -> bytes of a baseline JFIF file, Y sampled luma = (h, v), Cb / Cr 1x1; one interleaved scan, or
    (per_component) three non-interleaved scans -- a multi-scan SEQUENTIAL file."""
    H, W = rgb.shape[:2]
    h, v = luma
    dqt, dht, q, huff = _tables_like_pillow(quality)
    f = rgb.astype(np.float64)
    ycc = np.stack([0.299 * f[..., 0] + 0.587 * f[..., 1] + 0.114 * f[..., 2],
                    -0.168736 * f[..., 0] - 0.331264 * f[..., 1] + 0.5 * f[..., 2] + 128.0,
                    0.5 * f[..., 0] - 0.418688 * f[..., 1] - 0.081312 * f[..., 2] + 128.0], -1)
    mx, my = -(-W // (8 * h)), -(-H // (8 * v))
    PW, PH = mx * 8 * h, my * 8 * v
    pad = np.pad(ycc, ((0, PH - H), (0, PW - W), (0, 0)), mode="edge")
    planes = [pad[..., 0]] + [pad[..., c].reshape(PH // v, v, PW // h, h).mean((1, 3)) for c in (1, 2)]
    coefs = []
    for c, pl in enumerate(planes):
        qt = q[0 if c == 0 else 1]
        bh, bw = pl.shape[0] // 8, pl.shape[1] // 8
        blk = pl.reshape(bh, 8, bw, 8).transpose(0, 2, 1, 3) - 128.0
        d = dctn(blk, type=2, norm="ortho", axes=(2, 3)).reshape(bh, bw, 64)[..., ZIGZAG]
        coefs.append(np.rint(d / qt).astype(np.int64))
    out = bytearray(b"\xFF\xD8\xFF\xE0\x00\x10JFIF\x00\x01\x01\x00\x00\x01\x00\x01\x00\x00")
    out += dqt
    out += b"\xFF\xC0\x00\x11\x08" + H.to_bytes(2, "big") + W.to_bytes(2, "big") + bytes([3, 1, (h << 4) | v, 0, 2, 0x11, 1, 3, 0x11, 1])
    out += dht
    if restart:
        out += b"\xFF\xDD\x00\x04" + restart.to_bytes(2, "big")
    def block(bits, c, z, pred):
        dc_t, ac_t = huff[(0, 0 if c == 0 else 1)], huff[(1, 0 if c == 0 else 1)]
        s_, extra = _magnitude(int(z[0]) - pred[c])
        pred[c] = int(z[0])
        bits.put(*dc_t[s_])
        if s_:
            bits.put(extra, s_)
        run = 0
        last = int(np.flatnonzero(z).max(initial=0))
        for k in range(1, last + 1):
            if z[k] == 0:
                run += 1
                continue
            while run > 15:
                bits.put(*ac_t[0xF0])
                run -= 16
            s_, extra = _magnitude(int(z[k]))
            bits.put(*ac_t[(run << 4) | s_])
            bits.put(extra, s_)
            run = 0
        if last < 63:
            bits.put(*ac_t[0])

    def scan(header, units):
        """units: per MCU the list of (component, zig-zag block); restart markers every `restart` MCUs"""
        nonlocal out
        out += header
        bits, pred, rst = _Bits(), [0, 0, 0], 0
        for m, unit in enumerate(units):
            if restart and m and m % restart == 0:
                bits.flush()
                out += bits.out + bytes([0xFF, 0xD0 + (rst & 7)])
                bits, pred, rst = _Bits(), [0, 0, 0], rst + 1
            for c, z in unit:
                block(bits, c, z, pred)
        bits.flush()
        out += bits.out

    if not per_component:
        units = []
        for m in range(mx * my):
            r, cm = divmod(m, mx)
            units.append([(c, coefs[c][r * cv + by, cm * ch + bx]) for c in range(3) for (ch, cv) in (((h, v) if c == 0 else (1, 1)),)
                          for by in range(cv) for bx in range(ch)])
        scan(b"\xFF\xDA\x00\x0C\x03\x01\x00\x02\x11\x03\x11\x00\x3F\x00", units)
    else:
        # three non-interleaved scans (T.81 A.2.2): a component's own block grid, ceil(samples / 8) each way, one block per MCU
        for c in range(3):
            ch, cv = (h, v) if c == 0 else (1, 1)
            rw, rh = -(-W * ch // h), -(-H * cv // v)
            nbx, nby = -(-rw // 8), -(-rh // 8)
            units = [[(c, coefs[c][by, bx])] for by in range(nby) for bx in range(nbx)]
            scan(b"\xFF\xDA\x00\x08\x01" + bytes([c + 1, 0x00 if c == 0 else 0x11]) + b"\x00\x3F\x00", units)
    out += b"\xFF\xD9"
    return bytes(out)
