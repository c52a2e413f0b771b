"""Stage timing of the coarse step (backbone / transformer / match) without the full bench."""
import sys, torch
sys.path.insert(0, '.')
from detectorfreesfm_amd import HipLoFTR, synth, ops
from detectorfreesfm_amd.config import loftr_coarse_only_config
from detectorfreesfm_amd.params import loftr_param_spec, random_state_dict
dev = 'cuda:0'
cfg = loftr_coarse_only_config(0.2)
m = HipLoFTR(cfg); m.load_state_dict(random_state_dict(loftr_param_spec(cfg), 0)); m = m.eval().to(dev)
data = synth.to_device(synth.coarse_pair_batch(8, 480, 640, seed=1000), dev)
def t(fn, it=5):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / it
with torch.no_grad():
    P = m._packed or m._pack()
    imgs = torch.cat([data["image0"], data["image1"]], 0)
    c = m._backbone_hip(imgs, P).flatten(1, 2)
    pe = m._pe_tokens((60, 80))
    f0, f1 = c[:8], c[8:]
    print("backbone_ms", round(t(lambda: m._backbone_hip(imgs, P)), 3))
    print("transformer_ms", round(t(lambda: m._transformer(f0, f1, P, pe, pe)), 3))
    print("step_ms", round(t(lambda: m(dict(data))), 3))
