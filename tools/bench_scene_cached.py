"""Scene matching with the backbone evaluated once per image (plugin.match_scene_cached) vs once per pair:
24 images 640x480, all 276 pairs."""
import sys, time, torch
sys.path.insert(0, '.')
from detectorfreesfm_amd import HipLoFTR, plugin, synth
from detectorfreesfm_amd.config import loftr_coarse_only_config
from detectorfreesfm_amd.params import loftr_param_spec, random_state_dict
dev = 'cuda:0'
cfg = loftr_coarse_only_config(0.2)
m = HipLoFTR(cfg); m.load_state_dict(random_state_dict(loftr_param_spec(cfg), 0)); m = m.eval().to(dev)
n = 24
images = torch.rand((n, 1, 480, 640), generator=torch.Generator().manual_seed(0)).to(dev)
pairs = [(i, j) for i in range(n) for j in range(i + 1, n)]
plugin.match_scene_cached(m, images[:4], pairs[:3], batch=8)
torch.cuda.synchronize(); t0 = time.perf_counter()
plugin.match_scene_cached(m, images, pairs, batch=8)
torch.cuda.synchronize(); t1 = time.perf_counter()
with torch.no_grad():
    for lo in range(0, len(pairs), 8):
        ch = pairs[lo:lo + 8]
        d = {"image0": images[[p[0] for p in ch]], "image1": images[[p[1] for p in ch]]}
        m(d)
torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"{len(pairs)} pairs of {n} images: cached tokens {len(pairs) / (t1 - t0):.1f} pairs/s, per-pair backbone {len(pairs) / (t2 - t1):.1f} pairs/s")
