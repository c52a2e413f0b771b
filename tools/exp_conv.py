"""Experiment: MIOpen conv behaviour for the two CNNs (memory format x find mode)."""
import sys, time, torch
sys.path.insert(0, '.')
from detectorfreesfm_amd import HipLoFTR, HipMultiviewMatcher
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
cfg = loftr_coarse_only_config(0.2)
m = HipLoFTR(cfg); m.load_state_dict(random_state_dict(loftr_param_spec(cfg), 0)); m = m.eval().to(dev)
rcfg = multiview_refinement_config()
r = HipMultiviewMatcher(rcfg); r.load_state_dict(random_state_dict(multiview_param_spec(rcfg), 1)); r = r.eval().to(dev)
x = torch.rand(16, 1, 480, 640, device=dev)
p = torch.rand(10000, 3, 35, 35, device=dev)
mode = sys.argv[1]
with torch.no_grad():
    if 'bench' in mode:
        torch.backends.cudnn.benchmark = True
    P = m._pack(); RP = r._pack()
    if 'cl' in mode:
        def cl(o):
            if isinstance(o, torch.Tensor): return o.contiguous(memory_format=torch.channels_last) if o.dim() == 4 else o
            if isinstance(o, tuple): return tuple(cl(v) for v in o)
            if isinstance(o, dict): return {k: cl(v) for k, v in o.items()}
            if isinstance(o, list): return o
            return o
        P = cl(P); RP = cl(RP)
        x = x.contiguous(memory_format=torch.channels_last); p = p.contiguous(memory_format=torch.channels_last)
    print(mode, 'backbone16 ms', round(t(lambda: m._backbone(x, P)), 2), 's2dnet10k ms', round(t(lambda: r._s2dnet(p, RP, 15), 3), 2), flush=True)
