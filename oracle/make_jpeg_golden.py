"""Golden vectors of the JPEG decode (the reference's only data fixture): the eight frames of
SfM_dataset/example_dataset/example_scene/images copied to tests/golden/example_scene/ with a manifest of what libjpeg-turbo (the
library behind the reference's cv2.imread; here through the installed Pillow) decodes them to -- sha256 of the luma plane
(cv2.IMREAD_GRAYSCALE, src/dataset/utils.py:127, 183) and of the RGB frame (IMREAD_COLOR + BGR2RGB, :86-92).  The files are data
the reference ships for its own demo run; the GPU box has no /root/reference, so tests/test_gpu_jpeg.py decodes these copies on the
device.  Test infrastructure.  Run in the build container:  python oracle/make_jpeg_golden.py"""
import hashlib
import io
import json
import os
import shutil

import numpy as np
from PIL import Image

SRC = "/root/reference/SfM_dataset/example_dataset/example_scene/images"
DST = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "example_scene")


def main():
    os.makedirs(DST, exist_ok=True)
    manifest = {}
    for name in sorted(os.listdir(SRC)):
        buf = open(os.path.join(SRC, name), "rb").read()
        im = Image.open(io.BytesIO(buf))
        im.draft("L", im.size)                       # libjpeg-turbo's JCS_GRAYSCALE output: what IMREAD_GRAYSCALE asks for
        gray = np.asarray(im)
        rgb = np.asarray(Image.open(io.BytesIO(buf)).convert("RGB"))
        shutil.copyfile(os.path.join(SRC, name), os.path.join(DST, name))
        os.chmod(os.path.join(DST, name), 0o644)
        manifest[name] = dict(file_sha256=hashlib.sha256(buf).hexdigest(), height=int(gray.shape[0]), width=int(gray.shape[1]),
                              gray_sha256=hashlib.sha256(gray.tobytes()).hexdigest(), rgb_sha256=hashlib.sha256(rgb.tobytes()).hexdigest())
    import PIL
    from PIL import features
    json.dump(dict(decoder=f"Pillow {PIL.__version__} (libjpeg-turbo {features.version('jpg')})", files=manifest),
              open(os.path.join(DST, "manifest.json"), "w"), indent=1, sort_keys=True)
    print(f"{len(manifest)} files -> {DST}")


if __name__ == "__main__":
    main()
