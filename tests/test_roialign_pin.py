"""RoIAlign is the one stage whose third-party source is not in the reference snapshot.  The committed
recipe oracle/pin_roialign.py builds the upstream CPU kernel into oracle/_ref/ when its source is reachable; this file checks
the recipe's own logic and -- only when that library exists -- pins oracle/restate.py::roi_align_crop against it bit for bit."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import pin_roialign, restate


def test_recipe_extracts_the_kernel_function_and_reports_honestly():
    text = ("#include <TH/TH.h>\nstatic int helper(int a) { if (a) { return 1; } return 0; }\n"
            "void CropAndResizePerBox(\n    const float * image_data, const int batch_size)\n{\n    for (int b = 0; b < 2; ++b) {\n"
            "        if (b) { continue; }\n    }\n}\n\nvoid crop_and_resize_forward(THFloatTensor * image) { }\n")
    core = pin_roialign.extract_core(text)
    assert core.startswith("void CropAndResizePerBox(") and core.rstrip().endswith("}")
    assert "crop_and_resize_forward" not in core and "helper" not in core and core.count("{") == core.count("}")
    with pytest.raises(RuntimeError):
        pin_roialign.extract_core("void something_else(void) { }")
    src, tried = pin_roialign.find_source(allow_clone=False)          # the unit test never touches the network
    assert src is not None or len(tried) >= 2          # unreachable => every location that was looked at is listed


@pytest.mark.skipif(pin_roialign.load() is None,
                    reason="oracle/_ref/libcrop_and_resize.so not built (RoIAlign.pytorch source unreachable): a12 stays parity-unpinned")
def test_restatement_equals_upstream_kernel_bit_for_bit():
    f = pin_roialign.load()
    g = torch.Generator().manual_seed(12)
    N, C, H, W, crop = 3, 5, 48, 64, 35
    feat = torch.randn((N, C, H, W), generator=g)
    pts = torch.rand((40, 2), generator=g) * torch.tensor([W + 10.0, H + 10.0]) - 5.0        # some boxes leave the image
    boxes = torch.cat([pts - crop // 2, pts + crop // 2], -1)                                   # (x1, y1, x2, y2) pixels
    ind = torch.randint(0, N, (40,), generator=g).int()
    ref = restate.roi_align_crop(feat, boxes, ind, crop, crop, 0.0)
    # the normalisation RoIAlign.forward applies before the kernel (transform_fpcoor=False), in fp32 like the restatement
    norm = torch.stack([boxes[:, 1] / float(H - 1), boxes[:, 0] / float(W - 1), boxes[:, 3] / float(H - 1),
                        boxes[:, 2] / float(W - 1)], 1).float().contiguous()
    out = np.zeros((40, C, crop, crop), dtype=np.float32)
    fc = feat.contiguous().numpy()
    f(fc.ctypes.data_as(ctypes.c_void_p), N, C, H, W, norm.numpy().ctypes.data_as(ctypes.c_void_p),
      ind.numpy().ctypes.data_as(ctypes.c_void_p), 0, 40, out.ctypes.data_as(ctypes.c_void_p), crop, crop, ctypes.c_float(0.0))
    assert np.array_equal(out, ref.numpy())
