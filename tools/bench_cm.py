"""Timing of the coarse-match entry points at the bench shape (8 pairs, 4800 x 4800 x 256)."""
import sys, torch
sys.path.insert(0, '.')
from detectorfreesfm_amd import ops, synth
dev = 'cuda:0'
f0, f1 = synth.correlated_features(8, 4800, 4800, 256, 5, 0.1)
def split(f):
    hi = torch.where(f.abs() >= 2.0 ** -14, f, torch.zeros_like(f)).half()
    lo = ((f - hi.float()) * 2048.0).half()
    return ops.SplitAct(hi.to(dev), lo.to(dev), f.shape[-1])
ins = {"f32": (f0.to(dev), f1.to(dev)), "split": (split(f0), split(f1))}
for name, (a, b) in ins.items():
    for _ in range(2):
        out = ops.coarse_match(a, b, (60, 80), (60, 80), 0.2, 2, 0.1)
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5):
        out = ops.coarse_match(a, b, (60, 80), (60, 80), 0.2, 2, 0.1)
    e.record(); e.synchronize()
    print(f"{name:6s} {s.elapsed_time(e) / 5:7.3f} ms / call   matches {out['i_ids'].numel()}")
