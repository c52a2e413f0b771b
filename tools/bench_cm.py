"""Timing of the coarse-match entry points: 8 pairs of 4800 x 4800 x 256 (the bench shape) and, with --big, 2 pairs of
26 600 x 26 600 (1600 x 1064 frames).  usage: python tools/bench_cm.py [--big] [--split-only]"""
import os, sys, torch
sys.path.insert(0, '.')
from detectorfreesfm_amd import ops, synth
dev = 'cuda:0'
big = "--big" in sys.argv
shapes = [(8, 60, 80)] + ([(2, 133, 200)] if big else [])
def split(f):
    hi = torch.where(f.abs() >= 2.0 ** -14, f, torch.zeros_like(f)).half()
    lo = ((f - hi.float()) * 2048.0).half()
    return ops.SplitAct(hi.to(dev), lo.to(dev), f.shape[-1])
for N, h, w in shapes:
    L = h * w
    f0, f1 = synth.correlated_features(N, L, L, 256, 5, 0.1)
    ins = {"split": (split(f0), split(f1))}
    if "--split-only" not in sys.argv and L < 10000:
        ins["f32"] = (f0.to(dev), f1.to(dev))
    for name, (a, b) in ins.items():
        for _ in range(2):
            out = ops.coarse_match(a, b, (h, w), (h, w), 0.2, 2, 0.1)
        torch.cuda.synchronize()
        best = 1e9
        for rep in range(3):
            s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(5):
                out = ops.coarse_match(a, b, (h, w), (h, w), 0.2, 2, 0.1)
            e.record(); e.synchronize()
            best = min(best, s.elapsed_time(e) / 5)
        print(f"{os.environ.get('DFSFM_LIB_PATH', 'product').split('/')[-1]:16s} {N} x {L}^2 {name:6s} {best:7.3f} ms / call   matches {out['i_ids'].numel()}")
