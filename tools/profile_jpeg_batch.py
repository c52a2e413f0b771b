"""One decode_batch of 64 x 1600x1200 4:2:0 files (plans given), twice: host-side split of the call (staging / launch / finish) and a
target for rocprofv3 --kernel-trace --stats."""
import io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from test_jpeg_cpu import encode, synth
from detectorfreesfm_amd import jpeg, ops
dev = torch.device("cuda:0")
color = len(sys.argv) > 1 and sys.argv[1] == "rgb"
bufs = [encode(synth(1200, 1600, True, seed=100 + i), quality=90, subsampling=2) for i in range(8)] * 8
plans = [jpeg.plan(b) for b in bufs]
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    call = ops.jpeg_decode_batch_launch(plans, 3 if color else 1, dev)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    res = call.finish()
    t3 = time.perf_counter()
    print(f"rep {rep}: staging + uploads + launches {1e3 * (t1 - t0):.2f} ms, device drain {1e3 * (t2 - t1):.2f} ms, finish {1e3 * (t3 - t2):.2f} ms "
          f"-> {1e3 * (t3 - t0) / len(bufs):.3f} ms per file")
