"""A/B: dense_backend 'hip' (fp16x2-split MFMA kernels) vs 'library' (MIOpen / hipBLASLt fp32)."""
import sys, torch
sys.path.insert(0, '.')
from detectorfreesfm_amd import HipLoFTR, HipMultiviewMatcher, synth
from detectorfreesfm_amd.config import loftr_coarse_only_config, multiview_refinement_config
from detectorfreesfm_amd.params import loftr_param_spec, multiview_param_spec, random_state_dict

def t(fn, it=5):
    fn(); fn(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / it

dev = 'cuda:0'
cfg = loftr_coarse_only_config(0.2); sd = random_state_dict(loftr_param_spec(cfg), 0)
rcfg = multiview_refinement_config(); rsd = random_state_dict(multiview_param_spec(rcfg), 1)
data = synth.to_device(synth.coarse_pair_batch(8, 480, 640, seed=1000), dev)
rdata = synth.to_device(synth.refine_bag(2000, 5, 480, 640, seed=2000), dev)
feats = {}
for be in sys.argv[1:] or ['hip', 'library']:
    m = HipLoFTR(cfg, dense_backend=be); m.load_state_dict(sd); m = m.eval().to(dev)
    r = HipMultiviewMatcher(rcfg, dense_backend=be); r.load_state_dict(rsd); r = r.eval().to(dev)
    with torch.no_grad():
        P = m._pack()
        x = torch.cat([data['image0'], data['image1']], 0)
        tb = t(lambda: m._backbone_hip(x, P)) if be == 'hip' else t(lambda: m._backbone(x, P))
        f0, f1, _, _ = m.coarse_features(data['image0'], data['image1'])
        feats[be] = f0
        tt = t(lambda: m._transformer(f0, f1, P))
        tc = t(lambda: m(dict(data)))
        tr = t(lambda: r(dict(rdata)), 3)
    print(f"{be}: backbone16 {tb:.2f} ms  transformer {tt:.2f} ms  coarse step {tc:.2f} ms ({8000/tc:.1f} pairs/s)  refine step {tr:.2f} ms ({2e6/tr:.0f} tracks/s)", flush=True)
if len(feats) == 2:
    a, b = feats['hip'], feats['library']
    print('feature rel diff hip vs library: %.2e' % ((a - b).abs().max() / b.abs().max()).item())
