"""fine_match timing at BASELINE configs[2] (2000 tracks x 4 query views, W = 15, C = 128): fp32 and split-plane input."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from detectorfreesfm_amd import ops

dev = "cuda:0"
T, Vq, W, C = 2000, 4, 15, 128
g = torch.Generator().manual_seed(3)
ref = torch.randn((T, W * W, C), generator=g).to(dev)
qry = (0.7 * ref[:, None].cpu() + torch.randn((T, Vq, W * W, C), generator=g)).to(dev)
mask = torch.ones((T, Vq), dtype=torch.bool, device=dev)
mov = torch.ones((T,), dtype=torch.bool, device=dev)
rs = ops.SplitAct.empty_rows((T, W * W), C, dev)
qs = ops.SplitAct.empty_rows((T, Vq, W * W), C, dev)
ops.split_rows(ref.view(-1, C), out_split=ops.SplitAct(rs.hi.view(-1, C), rs.lo.view(-1, C), C))
ops.split_rows(qry.view(-1, C), out_split=ops.SplitAct(qs.hi.view(-1, C), qs.lo.view(-1, C), C))
nbytes = (T * Vq * W * W * C + T * 49 * C) * 4


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


for name, fn in (("fp32 input ", lambda: ops.fine_match(ref, qry, mask, mov, W, 7)),
                 ("split input", lambda: ops.fine_match(rs, qs, mask, mov, W, 7))):
    ms = timeit(fn)
    print(f"fine_match {name}: {ms:.3f} ms per {T} tracks x {Vq} views  ({nbytes / ms / 1e9:.2f} TB/s of {nbytes / 1e6:.0f} MB)")
o32, osp = ops.fine_match(rs.float(), qs.float(), mask, mov, W, 7), ops.fine_match(rs, qs, mask, mov, W, 7)
print("index equal:", bool(torch.equal(o32["best_index"], osp["best_index"])), " max |coords diff|:", float((o32["coords"] - osp["coords"]).abs().max()))
