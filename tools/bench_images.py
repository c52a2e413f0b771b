"""Image feeding: frames/s of images.read_grayscale on the device (decoded 1600x1200 uint8 frame -> [1,480,640] fp32,
the loftr setting resize 640 / df 8) with the frame already in HBM and including the pinned-host -> device copy, next to
the same step on one host core with Pillow (what the reference's reader does per image)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from PIL import Image
from detectorfreesfm_amd import images
dev = torch.device('cuda:0')
rng = np.random.default_rng(0)
H, W = 1200, 1600
frames = [rng.integers(0, 256, (H, W), dtype=np.uint8) for _ in range(8)]
pinned = [torch.from_numpy(f).pin_memory() for f in frames]
resident = [p.to(dev) for p in pinned]
kw = dict(resize=(640,), df=8)
for r in resident: out = images.read_grayscale(r, **kw)
torch.cuda.synchronize()
n = 400
t0 = time.perf_counter()
for i in range(n): out = images.read_grayscale(resident[i % 8], **kw)
torch.cuda.synchronize(); t1 = time.perf_counter()
for i in range(n): out = images.read_grayscale(pinned[i % 8], device=dev, **kw)
torch.cuda.synchronize(); t2 = time.perf_counter()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record()
for i in range(n): out = images.read_grayscale(resident[i % 8], **kw)
ev[1].record(); torch.cuda.synchronize()
t3 = time.perf_counter()
m = 20
for i in range(m):
    a = np.asarray(Image.fromarray(frames[i % 8]).resize((640, 480), resample=Image.LANCZOS)).astype('float32') / 255.
t4 = time.perf_counter()
byt = H * W + H * 640 * 2 + 480 * 640 * 4          # frame in, horizontal pass out + in, fp32 tensor out
print(f"read_grayscale 1600x1200 -> 640x480: resident {n / (t1 - t0):.0f} frames/s ({(t1 - t0) / n * 1e6:.0f} us wall, "
      f"{ev[0].elapsed_time(ev[1]) / n * 1e3:.0f} us device incl. launch gaps, {byt / 1e6:.2f} MB algorithmic), "
      f"with H2D {n / (t2 - t1):.0f} frames/s; Pillow on one host core {m / (t4 - t3):.1f} frames/s")
