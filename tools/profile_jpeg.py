"""n device decodes of one synthetic 1200x1600 4:2:0 file (for rocprofv3 --kernel-trace --stats): python tools/profile_jpeg.py [n] [gray|rgb]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from test_jpeg_cpu import encode, synth
from detectorfreesfm_amd import jpeg, ops
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
color = (sys.argv[2] if len(sys.argv) > 2 else "gray") == "rgb"
buf = encode(synth(1200, 1600, True, seed=1), quality=90, subsampling=2)
dev = torch.device("cuda:0")
pl = jpeg.plan(buf)
for _ in range(n):
    out, info = ops.jpeg_decode(pl, 3 if color else 1, dev)
torch.cuda.synchronize()
print(len(buf), "bytes,", pl.frame.nchunks, "chunks,", info)
