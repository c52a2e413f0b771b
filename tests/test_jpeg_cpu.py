"""JPEG decode, CPU side: (1) the oracle (oracle/jpeg_baseline.c, the sequential restatement of libjpeg-turbo's default
decompression path = what the reference's cv2.imread calls run) is pinned byte for byte to libjpeg-turbo itself through the
installed Pillow; (2) the CPU lane model of the DEVICE decoder (tests/jpeg_emul.cpp: the thread functions of csrc/jpeg_core.h
compiled with g++, kernels as loops, three thread orders) is held to the oracle on every supported sampling / restart / table
combination, through the resume path, and on the reference's own example-scene files where /root/reference is mounted;
(3) the host parser (detectorfreesfm_amd/jpeg.py) refuses what the device path does not take and reads the EXIF orientation."""
import io
import os
import sys

import numpy as np
import pytest
from PIL import Image

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import jpeg_emul                                                     # noqa: E402
import jpeg_testenc                                                  # noqa: E402
from cpu_standins import cpu_ops                                     # noqa: E402
from detectorfreesfm_amd import images, jpeg                         # noqa: E402
from oracle import restate_jpeg as rj                                # noqa: E402

REF_SCENE = "/root/reference/SfM_dataset/example_dataset/example_scene/images"


def synth(h, w, color=True, seed=0):
    """Smooth structure + noise + a bright patch + salt: long and short Huffman codes, big and small DC steps."""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w]
    base = np.stack([128 + 100 * np.sin(x / 17. + k) * np.cos(y / 23. - k) for k in range(3)], -1)
    base += rng.normal(0, 12, (h, w, 3))
    base[h // 3:h // 2, w // 4:w // 2] += 80
    base = (base + (rng.random((h, w, 1)) > 0.995) * 120).clip(0, 255).astype(np.uint8)
    return base if color else np.ascontiguousarray(base[..., 0])


def encode(img, **kw):
    b = io.BytesIO()
    Image.fromarray(img).save(b, "JPEG", **kw)
    return b.getvalue()


def pil_gray(buf):
    im = Image.open(io.BytesIO(buf))
    im.draft("L", im.size)                       # libjpeg-turbo's own JCS_GRAYSCALE output: what cv2.IMREAD_GRAYSCALE asks for
    return np.asarray(im)


def pil_rgb(buf):
    return np.asarray(Image.open(io.BytesIO(buf)).convert("RGB"))


def cases(sizes, subs=(0, 1, 2, "gray"), qualities=(5, 50, 92), restarts=(0, 1, 5), optimize=(False, True)):
    k = 0
    for (h, w) in sizes:
        for sub in subs:
            for q in qualities:
                for rst in restarts:
                    opt = optimize[k % len(optimize)]
                    k += 1
                    kw = dict(quality=q, optimize=opt)
                    if sub != "gray":
                        kw["subsampling"] = sub
                    if rst:
                        kw["restart_marker_blocks"] = rst
                    try:
                        yield (h, w, sub, q, rst, opt), encode(synth(h, w, sub != "gray", seed=k), **kw)
                    except OSError:              # Pillow's encoder buffer is too small for a few quality-100 optimised files
                        continue


def test_oracle_pinned_to_libjpeg_turbo():
    n = 0
    for key, buf in cases([(1, 1), (8, 8), (7, 5), (16, 16), (17, 33), (100, 75), (241, 319)], qualities=(5, 50, 92, 100)):
        assert np.array_equal(rj.decode(buf, False), pil_gray(buf)), key
        assert np.array_equal(rj.decode(buf, True), pil_rgb(buf)), key
        n += 1
    assert n > 300


@pytest.mark.skipif(not os.path.isdir(REF_SCENE), reason="reference example scene not present")
def test_oracle_and_lane_model_on_the_reference_scene():
    """The reference's only data fixture: eight 4:2:0 baseline JPEGs.  Oracle == libjpeg-turbo, lane model == oracle, with
    every sweep seeing only the previous sweep's states (order 1) at two chunk sizes."""
    for name in sorted(os.listdir(REF_SCENE)):
        buf = open(os.path.join(REF_SCENE, name), "rb").read()
        g, c = rj.decode(buf, False), rj.decode(buf, True)
        assert np.array_equal(g, pil_gray(buf)) and np.array_equal(c, pil_rgb(buf)), name
        for cb, color in ((128, False), (256, True)):
            out, info = jpeg_emul.decode(jpeg.plan(buf, cb), color, sweeps=12, order=1)
            assert info["status"][:3].tolist() == [0, 0, 0], (name, info)
            assert np.array_equal(out, c if color else g), name


def test_lane_model_matches_oracle_on_every_supported_layout():
    n = 0
    for key, buf in cases([(1, 1), (8, 8), (7, 5), (17, 33), (100, 75), (241, 319)]):
        pl = jpeg.plan(buf, 64 if n % 2 else 16)
        for color in (False, True):
            ref = rj.decode(buf, color)
            out, info = jpeg_emul.decode(pl, color, sweeps=3, order=n % 3)          # 3 sweeps per call: the resume path
            assert info["status"][:3].tolist() == [0, 0, 0], (key, info)
            assert np.array_equal(out, ref), (key, color, info)
        n += 1
    assert n > 200


def cases_440_411(sizes=((40, 56), (17, 33), (3, 3), (4, 2), (100, 75), (241, 319)), lumas=((1, 2), (4, 1))):
    """Files Pillow's encoder cannot write (4:4:0 = Y 1x2, 4:1:1 = Y 4x1) from the tests' own encoder (tests/jpeg_testenc.py;
    lumas (2, 1) / (2, 2) check that encoder against the Pillow-encoded cases)."""
    k = 0
    for luma in lumas:
        for (h, w) in sizes:
            for rst in (0, 3):
                for q in (30, 92):
                    k += 1
                    yield (luma, h, w, rst, q), jpeg_testenc.encode(synth(h, w, True, seed=k), luma, quality=q, restart=rst)


def test_440_and_411_sampling_oracle_pinned_and_lane_model():
    """cv2.imread takes 4:4:0 and 4:1:1 files (src/dataset/utils.py:86-92,127,183); r06: so does the device path.  libjpeg-turbo's
    jinit_upsampler picks h1v2_fancy_upsample for 4:4:0, int_upsample (replication) for 4:1:1, and the NON-fancy 2h routines when a
    chroma plane is at most two samples wide (frames up to four pixels wide): oracle == libjpeg-turbo, lane model == oracle."""
    n = 0
    for key, buf in cases_440_411(lumas=((1, 2), (4, 1), (2, 1), (2, 2))):
        g, c = rj.decode(buf, False), rj.decode(buf, True)
        assert np.array_equal(g, pil_gray(buf)) and np.array_equal(c, pil_rgb(buf)), key
        pl = jpeg.plan(buf, 64 if n % 2 else 16)
        assert pl.sampling[0] == key[0]
        for color in (False, True):
            out, info = jpeg_emul.decode(pl, color, sweeps=3, order=n % 3)
            assert info["status"][:3].tolist() == [0, 0, 0], (key, info)
            assert np.array_equal(out, c if color else g), (key, color)
        n += 1
    assert n == 96
    for sub in (1, 2):                                   # the narrow-frame rule on Pillow-written files as well
        for (h, w) in ((3, 3), (5, 4), (2, 2), (9, 1)):
            buf = encode(synth(h, w, True, seed=h * w), quality=90, subsampling=sub)
            assert np.array_equal(rj.decode(buf, True), pil_rgb(buf)), (sub, h, w)
            out, info = jpeg_emul.decode(jpeg.plan(buf, 16), True, sweeps=3, order=1)
            assert np.array_equal(out, rj.decode(buf, True)), (sub, h, w)


def cases_multiscan(sizes=((40, 56), (17, 33), (3, 3), (100, 75), (8, 8), (1, 1)), lumas=((1, 1), (2, 2), (2, 1), (1, 2), (4, 1))):
    """Sequential files with one component per scan (T.81 A.2.2; one block per MCU over the component's own block grid) from the tests'
    encoder -- Pillow's writes only interleaved scans."""
    k = 0
    for luma in lumas:
        for (h, w) in sizes:
            for rst in (0, 3):
                k += 1
                yield (luma, h, w, rst), jpeg_testenc.encode(synth(h, w, True, seed=k), luma, quality=80, restart=rst, per_component=True)


def test_multiscan_sequential_files():
    """cv2.imread reads sequential files whose components come in separate scans; r06: oracle == libjpeg-turbo on them, ``jpeg.plan``
    says MultiScanJpeg, ``plan_components`` turns the file into three grey frames, and the lane model (a grey decode per component +
    the colour stage on the planes) == oracle.  Partly interleaved files stay refused."""
    n = 0
    for key, buf in cases_multiscan():
        g, c = rj.decode(buf, False), rj.decode(buf, True)
        assert np.array_equal(g, pil_gray(buf)) and np.array_equal(c, pil_rgb(buf)), key
        with pytest.raises(jpeg.MultiScanJpeg):
            jpeg.plan(buf)
        cp = jpeg.plan_components(buf, 64 if n % 2 else 16)
        assert cp.sampling[0] == key[0] and (cp.width, cp.height) == (key[2], key[1])
        assert np.array_equal(jpeg_emul.decode_components(cp, False, sweeps=3, order=n % 3), g), key
        assert np.array_equal(jpeg_emul.decode_components(cp, True, sweeps=3, order=n % 3), c), key
        n += 1
    assert n == 60
    # Y alone, then Cb and Cr interleaved: one SOS with two components -- not taken (the host decoder reads it)
    buf = jpeg_testenc.encode(synth(24, 24, True, seed=3), (1, 1), per_component=True)
    i = buf.index(b"\xFF\xDA\x00\x08\x01\x02")
    bad = buf[:i] + b"\xFF\xDA\x00\x0A\x02\x02\x11\x03\x11\x00\x3F\x00" + buf[i + 10:]
    with pytest.raises(jpeg.UnsupportedJpeg):
        jpeg.plan_components(bad)
    with pytest.raises((jpeg.CorruptJpeg, jpeg.UnsupportedJpeg)):
        jpeg.plan_components(buf[:i + 40])                                # the file ends inside the second scan


def test_thread_order_does_not_change_the_fixed_point():
    buf = encode(synth(480, 640, True, seed=3), quality=90, subsampling=2)
    pl = jpeg.plan(buf)
    ref = rj.decode(buf, True)
    sweeps = []
    for order in (0, 1, 2):
        out, info = jpeg_emul.decode(pl, True, sweeps=64, order=order)
        assert np.array_equal(out, ref) and info["calls"] == 1
        sweeps.append(info["sweeps"])
    assert sweeps[0] == 1                    # ascending order = sequential decode: one sweep
    assert 1 < sweeps[1] <= 64               # previous-sweep states only: a handful (self-synchronisation), not nchunks
    assert pl.frame.nchunks > 200


def test_large_frame_multi_entry_scan_spans():
    """A frame large enough that the single-workgroup scans (block counts over the chunks, DC predictions over the MCU groups)
    give every thread a SPAN of entries: 30 000 MCUs, restart intervals that end inside spans, 16-byte chunks."""
    for kw, cb in ((dict(quality=90, subsampling=0, restart_marker_rows=7), 16), (dict(quality=85, subsampling=2), 32)):
        buf = encode(synth(1200, 1600, True, seed=11), **kw)
        pl = jpeg.plan(buf, cb)
        assert pl.frame.nchunks > 4 * 1024
        out, info = jpeg_emul.decode(pl, True, sweeps=64, order=1)
        assert info["status"][:3].tolist() == [0, 0, 0] and np.array_equal(out, rj.decode(buf, True))


def test_huffman_tables_are_the_canonical_code():
    """jpeg.huffman_table (the LDS image of a code) through the device's two-step lookup, restated in
    jpeg.huffman_decode_prefix: every code of every table of an optimised and of a standard file decodes to its (length, symbol)
    whatever bits follow it; prefixes no code owns give 0."""
    for buf in (encode(synth(64, 64), quality=75, optimize=True), encode(synth(64, 64), quality=100)):
        pl = jpeg.plan(buf)
        slots = pl.tab.view(np.uint8).reshape(4, jpeg.TAB_SLOT_BYTES)
        o, slot = 0, 0
        while o < len(pl.tab_key):
            bits = pl.tab_key[o:o + 16]
            nv = sum(bits)
            vals = pl.tab_key[o + 16:o + 16 + nv]
            o += 16 + nv
            code, k, last = 0, 0, 0
            for length in range(1, 17):
                for _ in range(bits[length - 1]):
                    lo, hi = code << (16 - length), ((code + 1) << (16 - length)) - 1
                    for prefix in (lo, hi, (lo + hi) // 2):
                        assert jpeg.huffman_decode_prefix(slots[slot], prefix) == ((length << 8) | vals[k]), (slot, length, code)
                    last = hi
                    code += 1
                    k += 1
                code <<= 1
            for prefix in {last + 1, 0xFFFF} - {0x10000}:
                if prefix > last:
                    assert jpeg.huffman_decode_prefix(slots[slot], prefix) == 0
            slot += 1
        assert slot >= 2


def test_parser_refuses_what_the_device_path_does_not_take():
    img = synth(64, 48)
    with pytest.raises(jpeg.UnsupportedJpeg):
        jpeg.plan(encode(img, progressive=True))
    with pytest.raises(jpeg.UnsupportedJpeg):
        b = io.BytesIO()
        Image.fromarray(img).convert("CMYK").save(b, "JPEG")
        jpeg.plan(b.getvalue())
    with pytest.raises(jpeg.CorruptJpeg):
        jpeg.plan(b"\x89PNG\r\n\x1a\n" + bytes(64))
    good = encode(img, quality=80)
    with pytest.raises(jpeg.CorruptJpeg):
        jpeg.plan(good[:200])                                           # cut inside the tables
    assert not jpeg.is_jpeg(b"\x89PNG") and jpeg.is_jpeg(good)
    info = rj.info(encode(img, progressive=True))
    assert info["progressive"] and not info["supported"]
    with cpu_ops():
        with pytest.raises(jpeg.UnsupportedJpeg):
            jpeg.decode(encode(img, progressive=True), False, device="cpu")


def test_corrupt_scans_terminate_and_are_order_independent():
    """Bit errors inside the entropy-coded bytes.  JPEG has no checksum: most flips just change coefficients, some change the
    block count of the segment (status[2]) or hit a prefix no code has (status[1]).  What must hold for ALL of them: the
    relaxation terminates, nothing is written out of bounds (the lane model runs under the kernels' own bounds), and the result
    -- image AND status -- is the same whatever order the threads of a sweep run in (the fixed point is unique)."""
    buf = bytearray(encode(synth(120, 160), quality=85, subsampling=2))
    pl0 = jpeg.plan(bytes(buf))
    start = bytes(buf).index(pl0.scan.tobytes()[:16])
    rng = np.random.default_rng(1)
    flagged = 0
    for trial in range(40):
        bad = bytearray(buf)
        for pos in rng.integers(start + 8, start + pl0.scan.size - 8, 1 + trial % 6):
            if bad[pos] != 0xFF and bad[pos - 1] != 0xFF:
                bad[pos] ^= 1 << int(rng.integers(0, 8))
                if bad[pos] == 0xFF:
                    bad[pos] = 0xFE
        pl = jpeg.plan(bytes(bad), 32)
        outs = [jpeg_emul.decode(pl, False, sweeps=8, order=o) for o in (0, 1, 2)]
        for out, info in outs:
            assert out.shape == (120, 160) and info["status"][0] == 0
            assert np.array_equal(out, outs[0][0]) and info["status"][:3].tolist() == outs[0][1]["status"][:3].tolist()
        flagged += int(outs[0][1]["status"][1] != 0 or outs[0][1]["status"][2] != 0)
    assert flagged >= 1
    # a truncated scan: blocks are missing, every caller sees it
    cut = bytes(buf[:start + pl0.scan.size // 2]) + b"\xff\xd9"
    out, info = jpeg_emul.decode(jpeg.plan(cut), False, sweeps=16, order=1)
    assert info["status"][2] != 0
    with cpu_ops():
        with pytest.raises(jpeg.CorruptJpeg):
            jpeg.decode(cut, False, device="cpu")


def test_exif_orientation_like_cv2_imread():
    img = synth(40, 56)
    from PIL import ImageOps
    for orientation in range(1, 9):
        exif = Image.Exif()
        exif[0x0112] = orientation
        b = io.BytesIO()
        Image.fromarray(img).save(b, "JPEG", quality=90, exif=exif)
        buf = b.getvalue()
        assert jpeg.plan(buf).orientation == orientation and rj.info(buf)["orientation"] == orientation
        want = np.asarray(ImageOps.exif_transpose(Image.open(io.BytesIO(buf))).convert("RGB"))
        assert np.array_equal(rj.apply_orientation(rj.decode(buf, True), orientation), want)
        with cpu_ops():
            got = jpeg.decode(buf, True, device="cpu")
        assert np.array_equal(got.numpy(), want), orientation


def test_readers_decode_files_on_the_product_path(tmp_path):
    """images.read_grayscale / read_rgb given a FILE NAME: decode='auto' sends a baseline JPEG through jpeg.decode (here on the
    lane model) and equals the reader fed the library's own decode; a progressive file falls back to the host decoder; the
    'device' setting refuses it."""
    img = synth(150, 200)
    p = tmp_path / "frame.jpg"
    p.write_bytes(encode(img, quality=88, subsampling=2))
    pp = tmp_path / "prog.jpg"
    pp.write_bytes(encode(img, quality=88, progressive=True))
    with cpu_ops():
        a = images.read_grayscale(str(p), resize=(96,), df=8, device="cpu")
        b = images.read_grayscale(pil_gray(p.read_bytes()), resize=(96,), df=8, device="cpu")
        assert np.array_equal(a.numpy(), b.numpy())
        a = images.read_rgb(str(p), resize=(96,), df=8, device="cpu", decode="device")
        b = images.read_rgb(pil_rgb(p.read_bytes()), resize=(96,), df=8, device="cpu")
        assert np.array_equal(a.numpy(), b.numpy())
        a = images.read_grayscale(str(pp), resize=(96,), df=8, device="cpu")            # auto -> host (Pillow here: the luma plane)
        b = images.read_grayscale(pil_gray(pp.read_bytes()), resize=(96,), df=8, device="cpu")
        assert np.array_equal(a.numpy(), b.numpy())
        with pytest.raises(jpeg.UnsupportedJpeg):
            images.read_grayscale(str(pp), device="cpu", decode="device")
        a = images.read_grayscale(str(p), resize=(96,), df=8, device="cpu", decode="host")
        assert np.array_equal(a.numpy(), images.read_grayscale(pil_gray(p.read_bytes()), resize=(96,), df=8, device="cpu").numpy())


def test_damaged_headers_raise_or_decode_within_bounds():
    """Files cut at random places and files with random bytes changed in the marker segments (frame size, sampling factors, table
    contents, restart interval, segment lengths): ``plan`` answers with CorruptJpeg / UnsupportedJpeg or a plan; every plan that
    comes back is run through the lane model of the device decoder, which works under the kernels' own bounds -- it must
    terminate and stay inside its buffers whatever the header claims (a wrong geometry shows up as status, not as a crash)."""
    rng = np.random.default_rng(7)
    good = [encode(synth(97, 131, True, seed=1), quality=80, subsampling=2, restart_marker_blocks=3),
            encode(synth(64, 64, False, seed=2), quality=60, optimize=True)]
    planned = refused = 0
    for trial in range(300):
        buf = bytearray(good[trial % 2])
        scan_at = bytes(buf).index(jpeg.plan(bytes(buf)).scan.tobytes()[:12])
        if trial % 3 == 0:
            buf = buf[:int(rng.integers(2, len(buf)))]
        else:
            for pos in rng.integers(2, scan_at, 1 + trial % 4):
                buf[pos] = int(rng.integers(0, 256))
        try:
            pl = jpeg.plan(bytes(buf), 32)
        except (jpeg.CorruptJpeg, jpeg.UnsupportedJpeg):
            refused += 1
            continue
        if pl.width * pl.height > 1 << 22:                     # a damaged size field: fine for the device, slow for the model
            continue
        planned += 1
        for color in (False, True):
            out, info = jpeg_emul.decode(pl, color, sweeps=8, order=1, max_calls=4)
            assert out.shape[:2] == (pl.height, pl.width)
    assert planned > 30 and refused > 30


def test_flat_and_saturated_frames_settle_in_a_few_launches():
    """an exactly flat area is one short code repeated, a decoder that enters it out of phase never
    re-synchronises, and the truth advances one chunk per relaxation round.  With the rounds INSIDE the sweep launch (256 chunks per
    workgroup) a flat frame settles in one launch per workgroup boundary of its longest restart interval -- ``Plan.launch_bound`` --
    in every workgroup order, and the bytes are libjpeg-turbo's."""
    flat_l = np.full((1024, 1024), 128, np.uint8)
    flat_c = np.full((1024, 1536, 3), 77, np.uint8)
    sky = synth(1500, 2000, True, seed=3)
    sky[:700] = 255                                                 # a blown-out sky above a textured ground
    big_flat = np.full((3000, 4000, 3), 200, np.uint8)              # 12 megapixels of one colour: ~1400 chunks, 6 workgroups
    for name, img, kw in [("flat L", flat_l, dict(quality=90)), ("flat 4:2:0", flat_c, dict(quality=90, subsampling=2)),
                          ("saturated sky", sky, dict(quality=92, subsampling=2)), ("12 MP flat", big_flat, dict(quality=95, subsampling=2)),
                          ("flat + restarts", flat_c, dict(quality=90, subsampling=0, restart_marker_rows=4))]:
        buf = encode(img, **kw)
        color = img.ndim == 3
        pl = jpeg.plan(buf)
        want = pil_rgb(buf) if color else pil_gray(buf)
        for order in (0, 1, 2):
            out, info = jpeg_emul.decode(pl, color, sweeps=4, order=order, max_calls=64)
            assert info["status"][0] == 0 and np.array_equal(out, want), (name, order, info)
            assert info["sweeps"] <= pl.launch_bound, (name, order, info, pl.launch_bound)
        assert name == "saturated sky" or pl.launch_bound <= 8, (name, pl.launch_bound)     # flat frames: a handful of workgroups


def test_auto_decode_falls_back_to_the_host_like_the_reference(tmp_path):
    """decode='auto' returns what the reference's reader returns for files libjpeg takes and the device path refuses: a wrong
    restart-marker count (CorruptJpeg from the parser), a 16-bit quantisation table (UnsupportedJpeg); decode='device' raises."""
    img = synth(96, 128)
    good = encode(img, quality=85, subsampling=2, restart_marker_rows=1)
    pl = jpeg.plan(good)
    k = good.rindex(b"\xff\xd3") if b"\xff\xd3" in good else None
    assert k is not None
    damaged = good[:k] + good[k + 2:]                               # one RSTn marker removed: libjpeg resynchronises with a warning
    p = tmp_path / "damaged.jpg"
    p.write_bytes(damaged)
    with pytest.raises(jpeg.CorruptJpeg):
        jpeg.plan(damaged)
    with cpu_ops():
        import warnings
        images._warned_device_fallback = False
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            a = images.read_grayscale(str(p), resize=(64,), df=8, device="cpu")
        assert any("device JPEG decode failed" in str(x.message) for x in w)
        b = images.read_grayscale(images._decode_host(str(p), False), resize=(64,), df=8, device="cpu")
        assert np.array_equal(a.numpy(), b.numpy())
        with pytest.raises(jpeg.CorruptJpeg):
            images.read_grayscale(str(p), device="cpu", decode="device")
    # a 16-bit DQT entry (pq = 1): refused by the parser, the host decodes it
    dqt = good.index(b"\xff\xdb")
    ln = (good[dqt + 2] << 8) | good[dqt + 3]
    seg = good[dqt + 4:dqt + 2 + ln]
    assert seg[0] >> 4 == 0
    wide = bytes([0x10 | (seg[0] & 15)]) + b"".join(bytes([0, v]) for v in seg[1:65]) + seg[65:]
    b16 = good[:dqt + 2] + (len(wide) + 2).to_bytes(2, "big") + wide + good[dqt + 2 + ln:]
    with pytest.raises(jpeg.UnsupportedJpeg, match="16-bit"):
        jpeg.plan(b16)
    assert np.array_equal(pil_gray(b16), pil_gray(good))            # the same image for libjpeg


def test_native_scan_index_equals_the_array_specification():
    """``dfsfm_jpeg_scan_index`` (host code of the product library: one memchr walk) against ``jpeg._scan_index_numpy`` (the rules as
    array passes) on encoder output of every layout -- restart markers, stuffed FFs, optimised tables -- and on adversarial byte
    strings: fill bytes (FF FF ...), an FF as the last byte, RST pairs back to back, an RST right before the end marker, no end
    marker at all, empty input; and whole plans built either way are identical."""
    import ctypes
    rng = np.random.default_rng(7)
    strings = [bytes(), b"\xff", b"\xff\xff", b"\x01\xff", b"\xff\x00", b"\xff\xd9", b"ab\xff\x00cd\xff\xd0\xff\xd1ef\xff\xff\xff\xd9zz",
               b"\xff\xd0" * 5 + b"\xff\xd9", b"x" * 5000 + b"\xff\x00" * 3000 + b"\xff\xd3" + b"y" * 9000 + b"\xff\xff\xd9", b"q" * 4096,
               b"q" * 4095 + b"\xff", b"q" * 4095 + b"\xff\x00" + b"r" * 10, b"\xff\xd7\xff\xd9", b"abc\xff\xc4zzz"]
    for _ in range(40):                                                   # random soups rich in FF / 00 / D0..D9
        n = int(rng.integers(1, 20000))
        a = rng.choice(np.array([0xFF, 0x00, 0xD0, 0xD5, 0xD9, 0x12, 0x80, 0xFF, 0x00], dtype=np.uint8), size=n,
                       p=[0.2, 0.2, 0.05, 0.05, 0.002, 0.2, 0.198, 0.05, 0.05])
        strings.append(a.tobytes())
    for sbytes in strings:
        d = np.frombuffer(sbytes, dtype=np.uint8)
        want = jpeg._scan_index_numpy(d, 1)
        nseg = want[1] + 1
        got = jpeg._scan_index_lib(d, nseg)
        assert got[0] == want[0] and got[1] == want[1], (sbytes[:40], got[:2], want[:2])
        for g, w in zip(got[2:], want[2:]):
            assert np.array_equal(g, w), (sbytes[:40], g[:8], w[:8])
        short = jpeg._scan_index_lib(d, max(1, nseg - 1))                # too few intervals expected: the counts still come back
        assert short[0] == want[0] and short[1] == want[1]
    n = 0
    for key, buf in cases([(17, 33), (100, 75), (241, 319)], qualities=(5, 92), restarts=(0, 1, 5)):
        try:
            jpeg.SCAN_INDEX = jpeg._scan_index_numpy
            a = jpeg.plan(buf)
        finally:
            jpeg.SCAN_INDEX = jpeg._scan_index_lib
        b = jpeg.plan(buf)
        for f in ("scan", "block_base", "seg_beg", "seg_end", "seg_chunk0", "chunk_seg"):
            assert np.array_equal(getattr(a, f), getattr(b, f)), (key, f)
        assert a.launch_bound == b.launch_bound and bytes(a.frame) == bytes(b.frame)
        n += 1
    assert n > 20
    from detectorfreesfm_amd import _lib
    assert _lib.lib().dfsfm_jpeg_scan_index(None, 0, None, 0, None, None, 1, None) == -1
