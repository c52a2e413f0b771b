"""Device JPEG decode throughput beside libjpeg-turbo (Pillow) on one host core.
python tools/bench_jpeg.py [--sizes 480x640,1200x1600,3000x4000] [--reps 20]
Per frame size: decode-only time with the file bytes already on the host (plan = host marker parse + chunk tables; H2D of the
scan; kernels), split into host plan / device time (events), and the Pillow decode of the same file (draft('L') = the luma plane;
RGB) on one core.  Synthetic frames (tests.test_jpeg_cpu.synth) at quality 90, 4:2:0 -- the layout of the reference's example scene."""
import argparse, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
from PIL import Image
from test_jpeg_cpu import encode, synth
from detectorfreesfm_amd import jpeg, ops

ap = argparse.ArgumentParser()
ap.add_argument("--sizes", default="480x640,1200x1600,3000x4000")
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--chunks", default="128,256")
args = ap.parse_args()
dev = torch.device("cuda:0")
pil_gray_2mp = float("nan")
for size in args.sizes.split(","):
    h, w = (int(v) for v in size.split("x"))
    buf = encode(synth(h, w, True, seed=1), quality=90, subsampling=2)
    t0 = time.perf_counter()
    for _ in range(5):
        im = Image.open(io.BytesIO(buf)); im.draft("L", im.size); np.asarray(im)
    pil_gray = (time.perf_counter() - t0) / 5
    t0 = time.perf_counter()
    for _ in range(5):
        np.asarray(Image.open(io.BytesIO(buf)).convert("RGB"))
    pil_rgb = (time.perf_counter() - t0) / 5
    if (h, w) == (1200, 1600):
        pil_gray_2mp = pil_gray
    print(f"{h}x{w}: file {len(buf) / 1e6:.2f} MB; Pillow (libjpeg-turbo, 1 core): gray {1e3 * pil_gray:.2f} ms, rgb {1e3 * pil_rgb:.2f} ms")
    for cb in (int(v) for v in args.chunks.split(",")):
        for color in (False, True):
            t0 = time.perf_counter()
            for _ in range(args.reps):
                pl = jpeg.plan(buf, cb)
            t_plan = (time.perf_counter() - t0) / args.reps
            out, info = ops.jpeg_decode(pl, 3 if color else 1, dev)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.reps):
                out, info = ops.jpeg_decode(pl, 3 if color else 1, dev)
            torch.cuda.synchronize()
            t_dev = (time.perf_counter() - t0) / args.reps
            mpix = h * w / 1e6
            print(f"   chunk {cb:4d} {'rgb ' if color else 'gray'}: plan {1e3 * t_plan:.2f} ms + upload/kernels/status {1e3 * t_dev:.3f} ms "
                  f"({mpix / t_dev:.0f} MPix/s, {len(buf) / t_dev / 1e6:.0f} MB/s of file; sweeps run {info['sweeps']}, "
                  f"sweeps that decoded something {info['sweeps_used']}; nchunks {pl.frame.nchunks})")

# ---- several decodes in flight: a scene's worth of frames -------------------------------------------------------------------
h, w = 1200, 1600
bufs = [encode(synth(h, w, True, seed=100 + i), quality=90, subsampling=2) for i in range(8)] * 4          # 32 files, 8 distinct
bufs = bufs * 2                                                                                            # 64 files
plans = [jpeg.plan(b) for b in bufs]
for color in (False, True):
    jpeg.decode_batch(bufs, color, dev, plans=plans)          # warm: staging buffer, workspace, side streams
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    outs = jpeg.decode_batch(bufs, color, dev, plans=plans)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    scan = sum(p.scan.size for p in plans)
    px = h * w * len(bufs)
    coef = sum(int(p.frame.nchunks) for p in plans)       # placeholder for the line below (chunks)
    print(f"decode_batch (plans given): {len(bufs)} x {h}x{w} {'rgb' if color else 'gray'}: {1e3 * dt / len(bufs):.3f} ms per frame = "
          f"{len(bufs) / dt:.0f} frames/s from ONE launching thread; {px / dt / 1e9:.2f} GPix/s, scan {scan / dt / 1e9:.2f} GB/s, "
          f"{coef} chunks")
t0 = time.perf_counter()
plans = [jpeg.plan(b) for b in bufs]
print(f"host parse alone (one thread): {1e3 * (time.perf_counter() - t0) / len(bufs):.3f} ms per file")
for k, bsz, nw in ((1, 14, 2), (2, 14, 2), (2, 14, 4), (2, 21, 4), (2, 14, 8), (3, 14, 8)):
    jpeg.decode_many(bufs, False, dev, streams=k, batch=bsz, workers=nw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    outs = jpeg.decode_many(bufs, False, dev, streams=k, batch=bsz, workers=nw)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"decode_many: {len(bufs)} x {h}x{w} gray, {k} stream(s) x batches of {bsz}: {1e3 * dt / len(bufs):.3f} ms per frame = {len(bufs) / dt:.0f} frames/s "
          f"(host parse on {nw} helper threads + launches included; Pillow on one core: {1 / pil_gray_2mp:.0f} frames/s)")
