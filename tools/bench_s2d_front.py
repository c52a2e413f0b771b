"""S2DNet front end (conv1_1 -> conv1_2 -> centre window + max-pool): the fused launch (ops.s2d_front, csrc/s2d_front.hip) beside the
three launches it replaces, 10 000 patches of 35 x 35 (BASELINE configs[2]: 2000 tracks x 5 views).  DFSFM_LIB_PATH selects an
experiment build of the library (the timing-only ablation / priority / skew / one-workgroup-per-CU switches it was written for were
temporary and have left csrc/s2d_front.hip; their numbers are in profiles/r06_s2d_front.txt)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from detectorfreesfm_amd import ops

dev = "cuda:0"
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
g = torch.Generator().manual_seed(1)
x = (torch.randn((n, 35, 35, 3), generator=g) * 1.3).to(dev)
w1 = torch.randn((64, 3, 3, 3), generator=g) * (2.0 / 27) ** 0.5
b1 = torch.randn((64,), generator=g) * 0.1
w2 = torch.randn((64, 64, 3, 3), generator=g) * (2.0 / 576) ** 0.5
b2 = torch.randn((64,), generator=g) * 0.1
fw = ops.S2dFrontWeights(w1.to(dev), b1.to(dev), w2.to(dev), b2.to(dev))
p1 = ops.PackedDense(w1.to(dev), b1.to(dev))
p2 = ops.PackedDense(w2.to(dev), b2.to(dev), cin_pad=64, tap_padded=True)
S = dict(relu=True, out_split=True)


def three():
    y = ops.conv2d_nhwc(ops.conv2d_nhwc(x, p1, 1, 1, **S), p2, 1, 1, **S)
    return y.crop(8, 27, 8, 27), ops.maxpool3x3s2_nhwc(y)


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


t_f = timed(lambda: ops.s2d_front(x, fw, 8, 27))
flop = n * 35 * 35 * 2 * (27 * 64 + 576 * 64)
tag = os.environ.get("DFSFM_LIB_PATH", "product build").split("/")[-1]
print(f"{tag:28s} fused {t_f:.3f} ms ({flop / t_f / 1e9:.0f} TFLOP/s algorithmic, {3 * flop / t_f / 1e9 / 2500:.3f} of the fp16 peak executed)", end="")
if "DFSFM_LIB_PATH" not in os.environ:
    t_3 = timed(three)
    print(f"; three launches {t_3:.3f} ms")
else:
    print()
