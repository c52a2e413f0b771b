"""Debug aid for csrc/s2d_front.hip: conv1_2 = identity (centre tap), so the whole-map 'crop' output IS relu1_1; prints where it
differs from a float64 conv1_1."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from detectorfreesfm_amd import ops
dev = "cuda:0"
g = torch.Generator().manual_seed(3)
n = 3
x = torch.randn((n, 35, 35, 3), generator=g)
w1 = torch.randn((64, 3, 3, 3), generator=g) * 0.3
b1 = torch.rand((64,), generator=g) * 0.5 + 0.5            # positive: relu mostly inactive
w2 = torch.zeros((64, 64, 3, 3)); w2[torch.arange(64), torch.arange(64), 1, 1] = 1.0
b2 = torch.zeros(64)
fw = ops.S2dFrontWeights(w1.to(dev), b1.to(dev), w2.to(dev), b2.to(dev))
crop, pool = ops.s2d_front(x.to(dev), fw, 0, 35)
got = crop.float().cpu().double()
ref = torch.relu(torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), w1.double(), b1.double(), 1, 1)).permute(0, 2, 3, 1)
err = (got - ref).abs()
print("max err", float(err.max()), "scale", float(ref.abs().max()))
print("err by channel block of 4:", [round(float(err[..., c:c + 4].max()), 4) for c in range(0, 64, 4)])
print("err by row:", [round(float(err[:, y].max()), 3) for y in range(35)])
print("err by col:", [round(float(err[:, :, xx].max()), 3) for xx in range(35)])
p = (0, 10, 10)
print("got", got[p][:8].tolist()); print("ref", ref[p][:8].tolist())
# which (pixel, channel) of the reference does got[p][c] equal, if any?
for c in range(4):
    d = (ref - got[p][c]).abs()
    i = int(d.argmin()); print("got[0,10,10,%d]=%.5f closest ref at" % (c, float(got[p][c])), tuple(int(v) for v in torch.unravel_index(torch.tensor(i), ref.shape)), "dist %.2e" % float(d.min()))
# single taps: which k does the kernel see?  weights one-hot in k
for k in (0, 1, 2, 3, 9, 13, 26):
    w1k = torch.zeros((64, 3, 3, 3)); tap, ci = k // 3, k % 3; w1k[:, ci, tap // 3, tap % 3] = 1.0
    fwk = ops.S2dFrontWeights(w1k.to(dev), torch.zeros(64).to(dev), w2.to(dev), b2.to(dev))
    xx = torch.rand((1, 35, 35, 3), generator=g) + 0.1
    c, _ = ops.s2d_front(xx.to(dev), fwk, 0, 35)
    gk = c.float().cpu()[0, 10, 10, 0]
    exp = xx[0, 10 + tap // 3 - 1, 10 + tap % 3 - 1, ci]
    d = (xx[0] - gk).abs(); i = torch.unravel_index(d.argmin(), d.shape)
    print(f"k={k} (ky {tap//3}, kx {tap%3}, ci {ci}): got {float(gk):.4f} expected {float(exp):.4f}; got matches input at (y,x,c)={tuple(int(v) for v in i)} dist {float(d.min()):.1e}")
