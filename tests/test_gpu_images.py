"""Image feeding on the device (last part): `dfsfm_resample_u8` through the C ABI returns the bytes of
the installed Pillow (the library the reference resizes with), and `images.read_grayscale / read_rgb` return what the
reference's own readers returned for the same frames (tests/golden/read_image.npz).  Integer / byte work: bit-exact."""
import os

import numpy as np
import pytest
import torch

from detectorfreesfm_amd import _lib, images, ops
from oracle import restate_resize as rr
from oracle.make_golden import read_image_cases, read_image_frame

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GOLD = os.path.join(os.path.dirname(__file__), "golden", "read_image.npz")


@pytest.mark.parametrize("H,W,C,size", [(157, 203, 1, (96, 72)), (157, 203, 3, (96, 72)), (60, 80, 1, (160, 120)),
                                        (100, 100, 1, (100, 40)), (100, 100, 3, (37, 100)), (64, 64, 3, (64, 64)),
                                        (17, 9, 1, (3, 2)), (2, 3, 3, (31, 17)), (1200, 1600, 1, (640, 480)),
                                        (968, 1296, 3, (1200, 896)), (3000, 4000, 1, (640, 480)), (480, 640, 1, (1293, 971))])
def test_resize_lanczos_equals_pillow(built_lib, H, W, C, size):
    from PIL import Image
    rng = np.random.default_rng(H * 1000 + W)
    img = rng.integers(0, 256, (H, W) if C == 1 else (H, W, 3), dtype=np.uint8)
    ref = np.asarray(Image.fromarray(img).resize(size, resample=Image.LANCZOS))
    out = images.resize_lanczos(img, size, device=DEV)
    assert out.dtype == torch.uint8 and tuple(out.shape) == ref.shape
    assert np.array_equal(out.cpu().numpy(), ref)


def test_resize_extreme_values_and_strided_rows(built_lib):
    """Saturating content (0 / 255 checkerboards drive the negative lobes past both ends of clip8) and a frame that is a
    view into a wider buffer (row pitch > W * C)."""
    from PIL import Image
    yy, xx = np.mgrid[0:240, 0:320]
    img = (((yy // 3 + xx // 5) % 2) * 255).astype(np.uint8)
    for size in ((130, 97), (640, 480)):
        ref = np.asarray(Image.fromarray(img).resize(size, resample=Image.LANCZOS))
        assert np.array_equal(images.resize_lanczos(img, size, device=DEV).cpu().numpy(), ref)
    wide = torch.from_numpy(np.random.default_rng(0).integers(0, 256, (200, 512), dtype=np.uint8)).to(DEV)
    view = wide[:, 37:337]
    bx, kx = images._device_tables(300, 111, torch.device(DEV))
    by, ky = images._device_tables(200, 77, torch.device(DEV))
    out = ops.resample_u8(view, bx, kx, by, ky, out_u8=True)[0]
    ref = np.asarray(Image.fromarray(view.cpu().numpy()).resize((111, 77), resample=Image.LANCZOS))
    assert np.array_equal(out.cpu().numpy(), ref)


def test_readers_equal_reference_golden(built_lib):
    gold = np.load(GOLD)
    for name, color, H, W, kw in read_image_cases():
        frame = read_image_frame(name, color, H, W)
        out = (images.read_rgb if color else images.read_grayscale)(frame, ret_scales=True, device=DEV, **kw)
        assert out[0].is_cuda and out[0].dtype == torch.float32
        assert np.array_equal(out[0].cpu().numpy(), gold[name + "/image"]), name
        assert np.array_equal(out[1].numpy(), gold[name + "/scales"]) and np.array_equal(out[2].numpy(), gold[name + "/original_hw"])
        if kw.get("ret_pad_mask"):
            assert np.array_equal(out[3].cpu().numpy(), gold[name + "/mask"]), name
        # the device-resident frame is accepted too (no host round trip)
        again = (images.read_rgb if color else images.read_grayscale)(torch.from_numpy(frame).to(DEV), **kw)
        again = again[0] if isinstance(again, list) else again
        assert torch.equal(again, out[0])


def test_reader_at_pipeline_size_equals_oracle(built_lib):
    """A 1296x968 frame to the larger side 640 with df 8 (the loftr setting) and to a padded 1200 square with mask
    (the matchformer setting, pad_to = -1): the oracle's tensors exactly."""
    frame = read_image_frame("big", False, 968, 1296)
    img, scales, hw = images.read_grayscale(frame, resize=(640,), df=8, ret_scales=True, device=DEV)
    o = rr.read_image(frame, resize=(640,), df=8)
    assert tuple(img.shape) == (1, 472, 640) and np.array_equal(img.cpu().numpy(), o[0])
    assert np.array_equal(scales.numpy(), o[1]) and hw.tolist() == [968, 1296]
    img, scales, hw, mask = images.read_grayscale(frame, resize=(1200,), df=8, pad_to=-1, ret_scales=True, ret_pad_mask=True, device=DEV)
    o = rr.read_image(frame, resize=(1200,), df=8, pad_to=-1)
    assert tuple(img.shape) == (1, 1200, 1200) and np.array_equal(img.cpu().numpy(), o[0]) and np.array_equal(mask.cpu().numpy(), o[3])


def test_resample_u8_argument_checks(built_lib):
    dev = torch.device(DEV)
    img = torch.zeros((8, 8), dtype=torch.uint8, device=dev)
    bx, kx = images._device_tables(8, 4, dev)
    with pytest.raises(_lib.DfsfmError):
        ops.resample_u8(img, bx, kx, bx, kx)                                   # no output requested
    with pytest.raises(_lib.DfsfmError):
        ops.resample_u8(img.float(), bx, kx, bx, kx, out_u8=True)
    with pytest.raises(_lib.DfsfmError):
        ops.resample_u8(torch.zeros((8, 8, 2), dtype=torch.uint8, device=dev), bx, kx, bx, kx, out_u8=True)   # C = 2
    with pytest.raises(_lib.DfsfmError):
        ops.resample_u8(img, bx, kx, bx, kx, lut=images._lut255(dev), pad_hw=(2, 2))     # padded frame smaller than the image
    with pytest.raises(_lib.DfsfmError):
        ops.resample_u8(img.cpu(), bx, kx, bx, kx, out_u8=True)                          # no CPU fallback
