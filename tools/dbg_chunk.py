"""Debug aid: chunk invariance / run-to-run determinism of the refinement head (16000-track bag vs its halves)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from detectorfreesfm_amd import HipMultiviewMatcher, synth, ops
from detectorfreesfm_amd.config import multiview_refinement_config
from detectorfreesfm_amd.params import multiview_param_spec, random_state_dict
from test_gpu_e2e import _subset_bag
DEV = "cuda:0"
cfg = multiview_refinement_config()
m = HipMultiviewMatcher(cfg, test=True); m.load_state_dict(random_state_dict(multiview_param_spec(cfg), 1), strict=True); m = m.eval().to(DEV)
T = int(sys.argv[1]) if len(sys.argv) > 1 else 16000
data = synth.refine_bag(T, 5, 480, 640, seed=31)
def run(d):
    h = synth.to_device(d, DEV); m(h)
    return h["query_points_refined"].clone(), h["reference_points_refined"][-1].clone(), h["std"][-1].clone()
def cmp(tag, a, b):
    for n, x, y in zip(("q", "r", "std"), a, b):
        df = (x - y).abs()
        nz = (df > 0)
        print(f"{tag} {n}: equal={torch.equal(x, y)} max|d|={df.max().item():.3e} differing elements={int(nz.sum())} of {df.numel()}")
full1, full2 = run(data), run(data)
cmp("full run-to-run", full1, full2)
half = T // 2
for lo, hi in ((0, half), (half, T)):
    sub = _subset_bag(data, torch.arange(lo, hi))
    h1, h2 = run(sub), run(sub)
    cmp(f"half[{lo}:{hi}] run-to-run", h1, h2)
    ref = (full1[0][:, lo:hi], full1[1][:, :, lo:hi], full1[2][..., lo:hi])
    cmp(f"half[{lo}:{hi}] vs full", h1, ref)
    q, qf = h1[0][0], ref[0][0]
    bad = ((q - qf).abs().sum(-1) > 0).nonzero().flatten()
    print("  first differing tracks:", bad[:10].tolist(), "count", bad.numel())
    ds = (h1[2] - ref[2]).abs().amax(dim=(0, 1))                     # per track
    hist = [int((ds[a:a + 800] > 0).sum()) for a in range(0, ds.numel(), 800)]
    print("  tracks with differing std per block of 800:", hist, "max per block", [f"{ds[a:a + 800].max().item():.1e}" for a in range(0, ds.numel(), 800)])
