"""Why does fine_match's launch time spread 213 - 324 us in the kernels-only profile?  One launch at a time with its
own event pair, in four situations: (a) 200 launches back to back; (b) each launch right after a burst of the power-hungry 3x3 convolution
(the matrix pipe's DVFS state); (c) each launch after a 300-us idle gap; (d) a DIFFERENT 0.93-GB input every launch (4 rotating bags: nothing
of the previous launch's data can still be in the 256-MB MALL or the L2s) against (a)'s single input."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from detectorfreesfm_amd import ops

dev = "cuda:0"
T, Vq, W, C = 2000, 4, 15, 128


def bag(seed):
    g = torch.Generator().manual_seed(seed)
    ref = torch.randn((T, W * W, C), generator=g)
    qry = 0.7 * ref[:, None] + torch.randn((T, Vq, W * W, C), generator=g)
    rs = ops.SplitAct.empty_rows((T, W * W), C, dev)
    qs = ops.SplitAct.empty_rows((T, Vq, W * W), C, dev)
    ops.split_rows(ref.to(dev).view(-1, C), out_split=ops.SplitAct(rs.hi.view(-1, C), rs.lo.view(-1, C), C))
    ops.split_rows(qry.to(dev).view(-1, C), out_split=ops.SplitAct(qs.hi.view(-1, C), qs.lo.view(-1, C), C))
    return rs, qs


bags = [bag(s) for s in range(4)]
mask = torch.ones((T, Vq), dtype=torch.bool, device=dev)
mov = torch.ones((T,), dtype=torch.bool, device=dev)
g = torch.Generator().manual_seed(0)
x = ops.SplitAct.empty(16, 240, 320, 128, dev)
ops.split_rows(torch.randn((16, 240, 320, 128), generator=g).to(dev), None, out_split=x)
pw = ops.PackedDense(torch.randn((128, 128, 3, 3), generator=g).to(dev) * 0.03, torch.zeros(128, device=dev), cin_pad=128, tap_padded=True)


def one(rs, qs, before=None):
    if before is not None:
        before()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    ops.fine_match(rs, qs, mask, mov, W, 7)
    b.record()
    b.synchronize()
    return a.elapsed_time(b) * 1e3


def stats(name, xs):
    xs = sorted(xs)
    n = len(xs)
    mean = sum(xs) / n
    sd = (sum((v - mean) ** 2 for v in xs) / n) ** 0.5
    print(f"{name:58s} n {n:4d}  min {xs[0]:6.1f}  p10 {xs[n // 10]:6.1f}  median {xs[n // 2]:6.1f}  p90 {xs[9 * n // 10]:6.1f}  max {xs[-1]:6.1f}  sd {sd:5.1f} us")


for _ in range(10):
    one(*bags[0])
stats("(a) back to back, one resident input", [one(*bags[0]) for _ in range(200)])
stats("(d) back to back, four rotating inputs (cold caches)", [one(*bags[i % 4]) for i in range(200)])
burst = lambda: [ops.conv2d_nhwc(x, pw, 1, 1, relu=True, out_split=True) for _ in range(4)]
stats("(b) right after 4 x conv3x3 128->128 @240x320 (4.3 ms of MFMA)", [one(*bags[i % 4], before=burst) for i in range(60)])
stats("(c) after a 300-us idle gap", [one(*bags[i % 4], before=lambda: time.sleep(3e-4)) for i in range(100)])
stats("(a') back to back again", [one(*bags[0]) for _ in range(200)])
