"""ASpanFormer coarse matcher (one pair per call, like the reference), 640x480: pairs/s over 4 rotating resident pairs and the
split backbone / span transformer / matching."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from detectorfreesfm_amd import synth
from detectorfreesfm_amd.aspanformer import HipASpanFormer, aspanformer_coarse_only_config
from detectorfreesfm_amd.coarse import backbone_tokens_hip
from detectorfreesfm_amd.params import aspanformer_param_spec, planted_aspanformer_state_dict
dev = 'cuda:0'
cfg = aspanformer_coarse_only_config(0.2)
m = HipASpanFormer(cfg); m.load_state_dict(planted_aspanformer_state_dict(aspanformer_param_spec(cfg), 0)); m = m.eval().to(dev)
pairs = [synth.to_device(synth.coarse_pair_batch(1, seed=1000 + 10 * i), dev) for i in range(4)]
steps = 40
with torch.no_grad():
    for p in pairs: m(dict(p))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 0
    for s in range(steps):
        d = dict(pairs[s % 4]); m(d); n += d["mkpts0_c"].shape[0]
    torch.cuda.synchronize(); t1 = time.perf_counter()
    P = m._packed
    for s in range(steps):
        p = pairs[s % 4]
        backbone_tokens_hip(torch.cat([p["image0"], p["image1"]], 0), P["bb"])
    torch.cuda.synchronize(); t2 = time.perf_counter()
ms, msb = (t1 - t0) / steps * 1e3, (t2 - t1) / steps * 1e3
print(f"aspanformer 1 pair 640x480: {ms:.2f} ms/pair = {1e3 / ms:.1f} pairs/s (backbone {msb:.2f} ms, span transformer + matching "
      f"{ms - msb:.2f} ms), {n / steps:.0f} matches/pair")
