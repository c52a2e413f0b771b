"""Device JPEG decode (csrc/jpeg_decode.hip through the C ABI) against the oracle (oracle/jpeg_baseline.c) and against
libjpeg-turbo itself (the installed Pillow): bytes identical for every supported sampling / restart / table layout, odd sizes,
EXIF orientations, a 12-megapixel frame; corrupt and unsupported files raise; the readers take file names."""
import io
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from test_jpeg_cpu import cases, cases_440_411, cases_multiscan, encode, pil_gray, pil_rgb, synth   # noqa: E402
from detectorfreesfm_amd import _lib, images, jpeg                   # noqa: E402
from oracle import restate_jpeg as rj                                # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_device_decode_matches_oracle_and_libjpeg_turbo():
    n = 0
    for key, buf in cases([(1, 1), (8, 8), (7, 5), (17, 33), (100, 75), (241, 319)]):
        for color in (False, True):
            ref = rj.decode(buf, color)
            assert np.array_equal(ref, pil_rgb(buf) if color else pil_gray(buf)), key
            out, info = jpeg.decode(buf, color, DEV, chunk_bytes=32 if n % 2 else 128, return_info=True)
            assert out.dtype == torch.uint8 and out.is_cuda
            assert np.array_equal(out.cpu().numpy(), ref), (key, color, info)
        n += 1
    assert n > 200


def test_440_and_411_sampling_on_the_device():
    """4:4:0 (h1v2 fancy upsampling) and 4:1:1 (replication) files from the tests' encoder, narrow frames included: the device's bytes
    are libjpeg-turbo's; a camera-sized 4:4:0 frame settles within the default sweep budget."""
    n = 0
    for key, buf in cases_440_411():
        for color in (False, True):
            ref = pil_rgb(buf) if color else pil_gray(buf)
            out, info = jpeg.decode(buf, color, DEV, chunk_bytes=32 if n % 2 else 128, return_info=True)
            assert np.array_equal(out.cpu().numpy(), ref), (key, color, info)
        n += 1
    assert n == 48
    import jpeg_testenc
    for luma in ((1, 2), (4, 1)):
        buf = jpeg_testenc.encode(synth(480, 640, True, seed=11), luma, quality=88, restart=0)
        out, info = jpeg.decode(buf, True, DEV, return_info=True)
        assert np.array_equal(out.cpu().numpy(), pil_rgb(buf)) and info["calls"] <= 2, (luma, info)
        assert np.array_equal(jpeg.decode_batch([buf, buf], True, DEV)[1].cpu().numpy(), pil_rgb(buf))


def test_multiscan_sequential_files_on_the_device():
    """One component per scan (what jpeg.plan alone refuses): three grey frames in one batched call + the colour stage on the planes
    (dfsfm_jpeg_ycc_planes_to_rgb_u8) = libjpeg-turbo's bytes; mixed into decode_batch / decode_many lists as well."""
    import jpeg_testenc
    n = 0
    for key, buf in cases_multiscan():
        for color in (False, True):
            out = jpeg.decode(buf, color, DEV, chunk_bytes=32 if n % 2 else 128)
            assert np.array_equal(out.cpu().numpy(), pil_rgb(buf) if color else pil_gray(buf)), (key, color)
        n += 1
    assert n == 60
    big = jpeg_testenc.encode(synth(480, 640, True, seed=21), (2, 2), quality=88, per_component=True)
    plain = encode(synth(240, 320, True, seed=22), quality=85, subsampling=2)
    for outs in (jpeg.decode_batch([plain, big, plain], True, DEV), jpeg.decode_many([plain, big, plain, big], True, DEV, batch=2)):
        for o, b in zip(outs, [plain, big, plain, big]):
            assert np.array_equal(o.cpu().numpy(), pil_rgb(b))
    assert np.array_equal(jpeg.decode(big, False, DEV).cpu().numpy(), pil_gray(big))


def test_camera_sized_frames_and_sweep_counts():
    """640x480 / 1600x1200 / 4000x3000 frames, 4:2:0 and 4:4:4, with and without restart markers: identical bytes; the relaxation
    settles within the default sweep budget (one call) on all of them."""
    for (h, w, kw) in [(480, 640, dict(quality=90, subsampling=2)), (1200, 1600, dict(quality=85, subsampling=2)),
                       (1200, 1600, dict(quality=95, subsampling=0)), (1200, 1600, dict(quality=85, subsampling=1, restart_marker_rows=1)),
                       (3000, 4000, dict(quality=90, subsampling=2))]:
        buf = encode(synth(h, w, True, seed=h), **kw)
        for color in (False, True):
            out, info = jpeg.decode(buf, color, DEV, return_info=True)
            assert np.array_equal(out.cpu().numpy(), rj.decode(buf, color)), (h, w, kw, color)
            assert info["calls"] <= 2, info
    buf = encode(synth(1200, 1600, False, seed=5), quality=80)             # a grey file, as RGB too
    assert np.array_equal(jpeg.decode(buf, True, DEV).cpu().numpy(), rj.decode(buf, True))
    assert np.array_equal(jpeg.decode(buf, False, DEV).cpu().numpy(), rj.decode(buf, False))


def test_few_sweeps_resume_to_the_same_bytes():
    buf = encode(synth(480, 640, True, seed=9), quality=90, subsampling=2)
    ref = rj.decode(buf, True)
    out, info = jpeg.decode(buf, True, DEV, sweeps=1, return_info=True)
    assert info["calls"] > 1 and np.array_equal(out.cpu().numpy(), ref)
    again = jpeg.decode(buf, True, DEV, sweeps=64)
    assert torch.equal(out, again)


def test_orientation_corrupt_and_unsupported():
    from PIL import Image, ImageOps
    img = synth(40, 56)
    for orientation in (1, 3, 6, 8, 5):
        exif = Image.Exif()
        exif[0x0112] = orientation
        b = io.BytesIO()
        Image.fromarray(img).save(b, "JPEG", quality=90, exif=exif)
        want = np.asarray(ImageOps.exif_transpose(Image.open(io.BytesIO(b.getvalue()))).convert("RGB"))
        assert np.array_equal(jpeg.decode(b.getvalue(), True, DEV).cpu().numpy(), want)
    good = encode(synth(120, 160), quality=85, subsampling=2)
    pl = jpeg.plan(good)
    start = good.index(pl.scan.tobytes()[:16])
    with pytest.raises(jpeg.CorruptJpeg):
        jpeg.decode(good[:start + pl.scan.size // 2] + b"\xff\xd9", False, DEV)
    with pytest.raises(jpeg.UnsupportedJpeg):
        jpeg.decode(encode(img, progressive=True), False, DEV)
    # the C ABI refuses bad arguments before anything is launched
    L = _lib.lib()
    assert L.dfsfm_jpeg_decode_workspace(None, 0, 1) == 0
    assert L.dfsfm_jpeg_decode_u8(None, 0, None, None, None, None, None, None, None, None, None, 0, 1, 1, 0, None, None, 0, None) == -1


def test_readers_take_file_names(tmp_path):
    img = synth(300, 400)
    p = tmp_path / "frame.jpg"
    p.write_bytes(encode(img, quality=88, subsampling=2))
    pp = tmp_path / "prog.jpg"
    pp.write_bytes(encode(img, quality=88, progressive=True))
    for path in (p, pp):                                                   # device decode / host fallback: the same luma plane
        a = images.read_grayscale(str(path), resize=(256,), df=8, device=DEV)
        b = images.read_grayscale(pil_gray(path.read_bytes()), resize=(256,), df=8, device=DEV)
        assert torch.equal(a, b)
    a = images.read_rgb(str(p), resize=(256,), df=8, device=DEV, decode="device")
    b = images.read_rgb(pil_rgb(p.read_bytes()), resize=(256,), df=8, device=DEV)
    assert torch.equal(a, b)
    with pytest.raises(jpeg.UnsupportedJpeg):
        images.read_grayscale(str(pp), device=DEV, decode="device")


def test_decode_many_overlaps_files_on_streams():
    bufs = [encode(synth(240 + 16 * i, 320 + 8 * i, True, seed=i), quality=70 + 3 * i, subsampling=(0, 1, 2)[i % 3],
                   **({"restart_marker_rows": 2} if i % 4 == 0 else {})) for i in range(9)]
    for color in (False, True):
        outs = jpeg.decode_many(bufs, color, DEV, streams=4)
        for buf, out in zip(bufs, outs):
            assert np.array_equal(out.cpu().numpy(), rj.decode(buf, color))
    assert torch.equal(jpeg.decode_many(bufs[:1], True, DEV, streams=1)[0], jpeg.decode(bufs[0], True, DEV))


def test_reference_example_scene_on_the_device():
    """The reference's own eight camera JPEGs (tests/golden/example_scene, copied by oracle/make_jpeg_golden.py together with the
    hashes of what libjpeg-turbo decodes them to): the device decode returns exactly those bytes, luma plane and RGB, and the
    sequential oracle agrees."""
    import hashlib
    import json
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "example_scene")
    man = json.load(open(os.path.join(root, "manifest.json")))["files"]
    assert len(man) == 8
    for name, m in man.items():
        buf = open(os.path.join(root, name), "rb").read()
        assert hashlib.sha256(buf).hexdigest() == m["file_sha256"]
        for color, key in ((False, "gray_sha256"), (True, "rgb_sha256")):
            out, info = jpeg.decode(buf, color, DEV, return_info=True)
            assert out.shape[:2] == (m["height"], m["width"])
            assert hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest() == m[key], (name, color, info)
            assert info["calls"] == 1, (name, info)                              # photographs settle within the default launches
        assert hashlib.sha256(rj.decode(buf, False).tobytes()).hexdigest() == m["gray_sha256"]
    outs = jpeg.decode_many([open(os.path.join(root, n), "rb").read() for n in man], False, DEV)
    for out, m in zip(outs, man.values()):
        assert hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest() == m["gray_sha256"]


def test_flat_and_saturated_frames_on_the_device():
    """valid files with large exactly-flat areas (no Huffman self-synchronisation there) decode on the device --
    the relaxation rounds run inside the sweep launches -- within ``Plan.launch_bound`` launches, to libjpeg-turbo's bytes."""
    flat_l = np.full((1024, 1024), 128, np.uint8)
    flat_c = np.full((1024, 1536, 3), 77, np.uint8)
    sky = synth(3000, 4000, True, seed=3)
    sky[:1400] = 255
    big_flat = np.full((3000, 4000, 3), 200, np.uint8)
    for name, img, kw in [("flat L", flat_l, dict(quality=90)), ("flat 4:2:0", flat_c, dict(quality=90, subsampling=2)),
                          ("saturated sky", sky, dict(quality=92, subsampling=2)), ("12 MP flat", big_flat, dict(quality=95, subsampling=2)),
                          ("flat + restarts", flat_c, dict(quality=90, subsampling=0, restart_marker_rows=4))]:
        buf = encode(img, **kw)
        color = img.ndim == 3
        pl = jpeg.plan(buf)
        want = pil_rgb(buf) if color else pil_gray(buf)
        for c in ((False, True) if color else (False,)):
            out, info = jpeg.decode(buf, c, DEV, return_info=True)
            ref = want if c == color else pil_gray(buf)
            assert np.array_equal(out.cpu().numpy(), ref), (name, c, info)
            assert info["sweeps_used"] <= pl.launch_bound, (name, info, pl.launch_bound)


def test_decode_batch_equals_single_decodes():
    """``dfsfm_jpeg_decode_batch_u8`` (grid.y = file, one set of launches per seven files): a mixed list -- sizes from 1 x 1 to
    1600 x 1200, grey / 4:4:4 / 4:2:2 / 4:2:0, with and without restart markers, a flat frame that needs extra launches, EXIF
    orientations, 17 files = two full groups and a ragged one -- gives exactly the bytes of the one-file entry point; a corrupt
    member raises after the batch, an unsupported one before anything is uploaded."""
    from PIL import Image
    bufs = []
    for i, (h, w) in enumerate([(1, 1), (8, 8), (17, 33), (100, 75), (241, 319), (480, 640), (1200, 1600), (64, 64), (333, 77),
                                (480, 640), (96, 128), (7, 5), (600, 800), (50, 1000), (1000, 50), (256, 256), (480, 640)]):
        kw = dict(quality=(35, 60, 85, 95)[i % 4])
        if i % 5 != 4:
            kw["subsampling"] = i % 3
        if i % 4 == 1:
            kw["restart_marker_rows"] = 1 + i % 3
        bufs.append(encode(synth(h, w, i % 5 != 4, seed=100 + i), **kw))
    bufs[7] = encode(np.full((1024, 1536, 3), 77, np.uint8), quality=90, subsampling=2)        # flat: extra sweep launches for one member
    exif = Image.Exif()
    exif[0x0112] = 6
    b = io.BytesIO()
    Image.fromarray(synth(40, 56)).save(b, "JPEG", quality=90, exif=exif)
    bufs[3] = b.getvalue()
    for color in (False, True):
        outs = jpeg.decode_batch(bufs, color, DEV)
        assert len(outs) == len(bufs)
        for buf, out in zip(bufs, outs):
            one = jpeg.decode(buf, color, DEV)
            assert out.shape == one.shape and torch.equal(out, one)
        many = jpeg.decode_many(bufs, color, DEV, streams=2, batch=5)
        for a, o in zip(many, outs):
            assert torch.equal(a, o)
    assert jpeg.decode_batch([], True, DEV) == []
    good = bufs[5]
    pl = jpeg.plan(good)
    start = good.index(pl.scan.tobytes()[:16])
    cut = good[:start + pl.scan.size // 2] + b"\xff\xd9"
    with pytest.raises(jpeg.CorruptJpeg):
        jpeg.decode_batch([bufs[4], cut, bufs[6]], False, DEV)
    with pytest.raises(jpeg.UnsupportedJpeg):
        jpeg.decode_batch([bufs[4], encode(synth(40, 56), progressive=True)], False, DEV)
    # the C ABI refuses bad arguments before anything is launched
    assert _lib.lib().dfsfm_jpeg_decode_batch_u8(None, 1, 1, 4, 0, None) == -1
