"""Image feeding (last part), CPU side: the numpy restatement of Pillow's 8-bit LANCZOS resize and
of the reference's readers is pinned against Pillow itself and against the reference's own reader functions; the
product's coefficient tables and host logic (detectorfreesfm_amd/images.py) are checked with the kernel arithmetic
emulated in numpy (tests/cpu_standins.py)."""
import os

import numpy as np
import pytest
import torch

from detectorfreesfm_amd import images
from oracle import ref_import, restate_resize as rr
from oracle.make_golden import read_image_cases, read_image_frame
from cpu_standins import cpu_ops

GOLD = os.path.join(os.path.dirname(__file__), "golden", "read_image.npz")

SIZES = [(157, 203, 1, (96, 72)), (157, 203, 3, (96, 72)), (60, 80, 1, (160, 120)), (100, 100, 1, (100, 40)),
         (100, 100, 3, (37, 100)), (480, 640, 1, (640, 480)), (33, 500, 1, (20, 33)), (64, 64, 3, (64, 64)),
         (17, 9, 1, (3, 2)), (2, 3, 3, (31, 17))]


@pytest.mark.parametrize("H,W,C,size", SIZES)
def test_oracle_resize_equals_pillow(H, W, C, size):
    """oracle/restate_resize.py vs the installed Pillow (the library the reference calls): identical bytes."""
    from PIL import Image
    rng = np.random.default_rng(H * 1000 + W)
    img = rng.integers(0, 256, (H, W) if C == 1 else (H, W, 3), dtype=np.uint8)
    ref = np.asarray(Image.fromarray(img).resize(size, resample=Image.LANCZOS))
    assert np.array_equal(rr.pil_resize_lanczos(img, size), ref)


@pytest.mark.parametrize("H,W,C,size", SIZES)
def test_product_tables_equal_oracle(H, W, C, size):
    for n_in, n_out in ((W, size[0]), (H, size[1])):
        b, k = images.lanczos_tables(n_in, n_out)
        assert b.dtype == np.int32 and k.dtype == np.int32 and b.shape == (n_out, 2) and k.shape[0] == n_out
        if n_in == n_out:
            assert np.array_equal(b[:, 0], np.arange(n_out)) and (b[:, 1] == 1).all() and (k == 1 << 22).all()
            continue
        for i, (lo, taps) in enumerate(rr.coeffs(n_in, n_out)):
            assert b[i, 0] == lo and b[i, 1] == len(taps) and np.array_equal(k[i, :len(taps)], taps)
            assert (k[i, len(taps):] == 0).all()


def _check_against(gold_get, reader_out, name, kw):
    img, scales, ohw = reader_out[:3]
    assert np.array_equal(np.asarray(img), gold_get(name + "/image"))
    assert np.array_equal(np.asarray(scales), gold_get(name + "/scales"))
    assert np.array_equal(np.asarray(ohw), gold_get(name + "/original_hw"))
    if kw.get("ret_pad_mask"):
        assert np.array_equal(np.asarray(reader_out[3]), gold_get(name + "/mask"))


def test_oracle_readers_equal_golden():
    """oracle read_image vs tests/golden/read_image.npz (the reference's own read_grayscale / read_rgb on real Pillow)."""
    gold = np.load(GOLD)
    for name, color, H, W, kw in read_image_cases():
        kw2 = {k: v for k, v in kw.items() if k != "ret_pad_mask"}
        out = rr.read_image(read_image_frame(name, color, H, W), **kw2)
        _check_against(lambda k: gold[k], out, name, kw)
        assert out[0].dtype == np.float32 and gold[name + "/image"].dtype == np.float32


def test_product_readers_host_logic_equal_golden():
    """images.read_grayscale / read_rgb (signature, resize rule, padding, masks, return layout) with the kernel emulated."""
    gold = np.load(GOLD)
    with cpu_ops():
        for name, color, H, W, kw in read_image_cases():
            frame = read_image_frame(name, color, H, W)
            out = (images.read_rgb if color else images.read_grayscale)(frame, ret_scales=True, device="cpu", **kw)
            _check_against(lambda k: gold[k], [o.numpy() if isinstance(o, torch.Tensor) else o for o in out], name, kw)
            assert out[0].dtype == torch.float32 and out[1].dtype == torch.float32 and out[2].dtype == torch.int64
        frame = read_image_frame("gray_df8", False, 157, 203)
        assert isinstance(images.read_grayscale(frame, resize=(96,), df=8, device="cpu"), torch.Tensor)     # bare tensor
        assert images.read_grayscale(frame, pad_to=None, ret_pad_mask=True, device="cpu")[1] is None        # utils.py:158
        with pytest.raises(ValueError):
            images.read_rgb(frame, device="cpu")
        with pytest.raises(NotImplementedError):
            images.read_grayscale(frame, augmentor=object(), device="cpu")


@pytest.mark.skipif(not ref_import.reference_available(), reason="needs /root/reference")
def test_golden_is_what_the_reference_returns_now():
    frames = {}
    read_gray, read_rgb = ref_import.import_image_readers(frames)
    gold = np.load(GOLD)
    for name, color, H, W, kw in read_image_cases():
        frames[name] = read_image_frame(name, color, H, W)
        out = (read_rgb if color else read_gray)(name, ret_scales=True, **kw)
        _check_against(lambda k: gold[k], [o.numpy() for o in out], name, kw)


def test_example_jpeg_through_the_readers():
    """One of the reference's example JPEGs through oracle and product at the pipeline's setting (larger side 640, df 8): the
    product decodes the FILE (jpeg.decode on the CPU lane model of the device decoder) to the luma plane -- what
    cv2.imread(IMREAD_GRAYSCALE) and Pillow's draft('L') both get from libjpeg-turbo -- and then agrees with the oracle reader
    and with Pillow's own resize of those bytes."""
    from PIL import Image
    root = "/root/reference/SfM_dataset/example_dataset/example_scene/images"
    if not os.path.isdir(root):
        pytest.skip("reference example scene not present")
    path = os.path.join(root, sorted(os.listdir(root))[0])
    im = Image.open(path)
    im.draft("L", im.size)
    gray = np.asarray(im)
    h, w = gray.shape
    o_img, o_scales, o_hw, _ = rr.read_image(gray, resize=(640,), df=8)
    w_new, h_new = rr.process_resize(w, h, (640,), 8)
    pil = np.asarray(Image.fromarray(gray).resize((w_new, h_new), resample=Image.LANCZOS), dtype=np.float32) / 255.
    assert np.array_equal(o_img[0], pil.astype(np.float32))
    with cpu_ops():
        p_img, p_scales, p_hw = images.read_grayscale(path, resize=(640,), df=8, ret_scales=True, device="cpu")
    assert np.array_equal(p_img.numpy(), o_img) and np.array_equal(p_scales.numpy(), o_scales)
    assert p_hw.tolist() == [h, w]
