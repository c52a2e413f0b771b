"""Per-launch times of HipMultiviewMatcher._s2dnet_hip on the benched bag's 10 000 patches (2000 tracks x 5 views):
every ops.* call of the pass timed with a device sync either side, averaged over n passes.  The kernel-trace stats group launches by
kernel NAME (five different layers share conv_gemm_sf_same_kernel<128,3,4>); this shows each layer by itself.
python tools/bench_s2d_layers.py [n_patches] [passes]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from detectorfreesfm_amd import HipMultiviewMatcher, ops
ops.RANGE_SWEEP = False
from detectorfreesfm_amd.config import multiview_refinement_config
from detectorfreesfm_amd.params import multiview_param_spec, random_state_dict

m_patches = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = 'cuda:0'
cfg = multiview_refinement_config()
model = HipMultiviewMatcher(cfg); model.load_state_dict(random_state_dict(multiview_param_spec(cfg), 1)); model = model.eval().to(dev)
P = model._packed or model._pack()
mt = cfg["multiview_transform"]
W, crop, C = mt["window_size"], mt["crop_size"], mt["d_model"]
g = torch.Generator().manual_seed(3)
x = torch.randn((m_patches, crop, crop, 3), generator=g).to(dev)
dst = torch.empty((m_patches, W * W, C), device=dev)

log = []
def timed(name, fn):
    def w(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = fn(*a, **k)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        t = a[0]
        shp = tuple((t.hi if isinstance(t, ops.SplitAct) else t).shape)
        extra = ""
        if name == "conv2d_nhwc":
            pw = a[1]
            extra = f" {pw.kh}x{pw.kw} {pw.Cin_act}->{pw.Cout}"
            flops = 2.0 * pw.kh * pw.kw * pw.Cin_act * pw.Cout * (r.hi if isinstance(r, ops.SplitAct) else r).numel() / pw.Cout
        else:
            flops = 0.0
        log.append((name + extra, shp, dt, flops))
        return r
    return w
for nm in ("conv2d_nhwc", "maxpool3x3s2_nhwc", "s2d_front", "resample_separable"):
    setattr(ops, nm, timed(nm, getattr(ops, nm)))

with torch.no_grad():
    model._s2dnet_hip(x, P, W, dst)      # warm
    log.clear()
    for _ in range(passes):
        model._s2dnet_hip(x, P, W, dst)
n = len(log) // passes
tot = 0.0
for i in range(n):
    ts = sorted(log[i + p * n][2] for p in range(passes))
    name, shp, _, fl = log[i]
    t = ts[len(ts) // 2]
    tot += t
    print(f"{name:32s} in {str(shp):24s} {1e3 * t:7.3f} ms" + (f"  {fl / t / 1e12:6.1f} TFLOP/s ({fl / t / 2.5e15:.3f} of fp16 MFMA peak)" if fl else ""))
print(f"sum {1e3 * tot:.3f} ms for {m_patches} patches (sync-bracketed launches)")
