"""Does v_mfma_f32_32x32x16_f16 keep fp16 SUBNORMAL inputs on gfx950, or flush them?  Feeds subnormal values straight in as
the hi plane of a split activation (lo = 0) through an identity weight matrix: the fp32 output equals the value if they are
kept, 0 if they are flushed.  (split_f32 routes |x| < 2^-14 entirely to the lo plane so that the answer never matters.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from detectorfreesfm_amd import ops
dev = "cuda:0"
for val in (3e-5, 6e-8, 1e-6):
    hi = torch.full((256, 64), val, dtype=torch.float16, device=dev)
    x = ops.SplitAct(hi, torch.zeros_like(hi), 64)
    w = ops.PackedDense(torch.eye(64, device=dev), cin_pad=64)
    y = ops.linear(x, w)
    print(f"fp16 input {float(hi[0, 0]):.6e} (subnormal: {float(hi[0, 0]) < 6.1e-5}) -> MFMA output {float(y[0, 0]):.6e}")
# and subnormal WEIGHTS against normal activations
hi = torch.ones((256, 64), dtype=torch.float16, device=dev)
x = ops.SplitAct(hi, torch.zeros_like(hi), 64)
w = ops.PackedDense(torch.eye(64, device=dev), cin_pad=64)
w.hi[:64, :64] = torch.eye(64, device=dev).half() * 3e-5     # bypass the packing rule: subnormal hi weights
y = ops.linear(x, w)
print(f"subnormal weight {float(w.hi[0, 0]):.6e} x 1.0 -> {float(y[0, 0]):.6e}")
