"""n forwards of one plugin (for rocprofv3 --kernel-trace --stats):
python tools/profile_step.py [coarse|refine|aspan|matchformer] [n]; prints the wall time per forward of the last n-1.
The trace covers the whole process, so what is not part of a warm step is kept out of it (VERDICT r04 #10c): the first-call range
sweep is switched off here (its abs-max reductions and the fused layers' five-GEMM shadow pass are one-time work), and the only
one-time launches left are the weight uploads of the first forward -- 1/n of the copyBuffer rows (n = 10 in tools/gpu_measure.sh)."""
import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from detectorfreesfm_amd import HipLoFTR, HipMultiviewMatcher, ops, synth
ops.RANGE_SWEEP = False
from detectorfreesfm_amd.config import loftr_coarse_only_config, multiview_refinement_config
from detectorfreesfm_amd.params import loftr_param_spec, multiview_param_spec, planted_loftr_state_dict, random_state_dict
which = sys.argv[1] if len(sys.argv) > 1 else 'coarse'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = 'cuda:0'
if which == 'coarse':
    cfg = loftr_coarse_only_config(0.2)
    m = HipLoFTR(cfg); m.load_state_dict(planted_loftr_state_dict(loftr_param_spec(cfg), 0)); m = m.eval().to(dev)
    data = synth.to_device(synth.coarse_pair_batch(8, 480, 640, seed=1000), dev)
elif which == 'aspan':
    from detectorfreesfm_amd.aspanformer import HipASpanFormer, aspanformer_coarse_only_config
    from detectorfreesfm_amd.params import aspanformer_param_spec, planted_aspanformer_state_dict
    cfg = aspanformer_coarse_only_config(0.4)
    m = HipASpanFormer(cfg); m.load_state_dict(planted_aspanformer_state_dict(aspanformer_param_spec(cfg), 0), strict=True)
    m = m.eval().to(dev)
    data = synth.to_device(synth.coarse_pair_batch(1, 480, 640, seed=1000), dev)
elif which == 'aspan_scene':      # the scene path: cached backbone tokens, PAIRS_PER_PASS pairs per transformer pass
    from detectorfreesfm_amd.aspanformer import HipASpanFormer, aspanformer_coarse_only_config
    from detectorfreesfm_amd.params import aspanformer_param_spec, planted_aspanformer_state_dict
    cfg = aspanformer_coarse_only_config(0.4)
    mm = HipASpanFormer(cfg); mm.load_state_dict(planted_aspanformer_state_dict(aspanformer_param_spec(cfg), 0), strict=True)
    mm = mm.eval().to(dev)
    b = synth.to_device(synth.coarse_pair_batch(8, 480, 640, seed=1000), dev)
    with torch.no_grad():
        tk0, hw = mm.image_tokens(b["image0"])
        tk1, _ = mm.image_tokens(b["image1"])
    m = lambda d: mm.match_tokens(tk0, tk1, hw, hw, (480, 640))
    data = {}
elif which == 'matchformer':
    from detectorfreesfm_amd.matchformer import HipMatchformer, matchformer_coarse_only_config
    from detectorfreesfm_amd.params import matchformer_param_spec, planted_matchformer_state_dict
    cfg = matchformer_coarse_only_config(0.4)
    m = HipMatchformer(cfg); m.load_state_dict(planted_matchformer_state_dict(matchformer_param_spec(), 0), strict=True)
    m = m.eval().to(dev)
    data = synth.to_device(synth.coarse_pair_batch(8, 480, 640, seed=1000), dev)
else:
    cfg = multiview_refinement_config()
    m = HipMultiviewMatcher(cfg); m.load_state_dict(random_state_dict(multiview_param_spec(cfg), 1)); m = m.eval().to(dev)
    data = synth.to_device(synth.refine_bag(2000, 5, 480, 640, seed=2000), dev)
import time
m(dict(data))
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(n - 1):
    m(dict(data))
torch.cuda.synchronize()
if n > 1:
    print(f"{which}: {1e3 * (time.perf_counter() - t0) / (n - 1):.3f} ms per forward (wall)")
