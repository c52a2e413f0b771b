"""MatchFormer-LA (large) coarse matcher, 640x480, batch 8, 4 resident batches rotated: pairs/s and the
per-stage split (backbone stages timed by running coarse_features alone)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from detectorfreesfm_amd import synth
from detectorfreesfm_amd.matchformer import HipMatchformer, matchformer_coarse_only_config
from detectorfreesfm_amd.params import matchformer_param_spec, planted_matchformer_state_dict
dev = 'cuda:0'
cfg = matchformer_coarse_only_config(0.4)
m = HipMatchformer(cfg); m.load_state_dict(planted_matchformer_state_dict(matchformer_param_spec(), 0)); m = m.eval().to(dev)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
batches = [synth.to_device(synth.coarse_pair_batch(B, seed=1000 + 10 * i), dev) for i in range(4)]
steps = 12
with torch.no_grad():
    for b in batches: m(dict(b))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 0
    for s in range(steps):
        d = dict(batches[s % 4]); m(d); n += d["mkpts0_c"].shape[0]
    torch.cuda.synchronize(); t1 = time.perf_counter()
    for s in range(steps):
        b = batches[s % 4]
        m.coarse_features(torch.cat([b["image0"], b["image1"]], 0), B)
    torch.cuda.synchronize(); t2 = time.perf_counter()
ms = (t1 - t0) / steps * 1e3; msb = (t2 - t1) / steps * 1e3
print(f"matchformer-LA large {B} pairs 640x480: {ms:.2f} ms/step = {B / ms * 1e3:.1f} pairs/s, backbone(+attention) {msb:.2f} ms, "
      f"coarse matching {ms - msb:.2f} ms, {n / steps / B:.0f} matches/pair")
