"""Where does the coarse step's time go INSIDE the steady-state loop?  bench.py's `breakdown` times each stage alone (5 launches on one
resident batch); the step is 4 - 5 % longer than the sum of those.  This script records events at the stage boundaries of real forwards in
the bench loop (4 rotating batches, 30 steps) and prints the in-loop stage times next to the stand-alone ones: the difference is either
idle time between stages (host launch gaps) or the same kernels running slower in a sustained power-limited loop than in a short burst.
usage: python tools/step_timeline.py"""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from detectorfreesfm_amd import ops, synth
import bench

dev = torch.device("cuda:0")
m = bench.build_coarse(dev)
batches = [synth.to_device(synth.coarse_pair_batch(8, 480, 640, seed=1000 + 10 * k), dev) for k in range(4)]
for k in range(3):
    m(dict(batches[k]))
P = m._packed
mc = m.config["match_coarse"]
pe = m._pe_tokens((60, 80))


def staged(d, ev):
    ev[0].record()
    imgs = torch.cat([d["image0"], d["image1"]], 0)
    c = m._backbone_hip(imgs, P).flatten(1, 2)
    ev[1].record()
    m._transformer(c[:8], c[8:], P, pe, pe)
    g0, g1 = m._feat_split
    ev[2].record()
    out = ops.coarse_match(g0, g1, (60, 80), (60, 80), mc["thr"], mc["border_rm"], mc["dsmax_temperature"], d["scale0"], d["scale1"], 8.0)
    ev[3].record()
    return out


with torch.no_grad():
    n = 30
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(n)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        staged(batches[i % 4], evs[i])
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / n * 1e3
    st = [[evs[i][k].elapsed_time(evs[i][k + 1]) for k in range(3)] for i in range(5, n)]
    gaps = [evs[i][3].elapsed_time(evs[i + 1][0]) for i in range(5, n - 1)]
    mean = lambda xs: sum(xs) / len(xs)
    print(f"in-loop  : backbone {mean([s[0] for s in st]):.3f}  transformer {mean([s[1] for s in st]):.3f}  match {mean([s[2] for s in st]):.3f}  "
          f"between steps {mean(gaps):.3f}  | wall per step {wall:.3f} ms")
    # the same forward through the product entry point, for reference
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        m(dict(batches[i % 4]))
    torch.cuda.synchronize()
    print(f"forward(): wall per step {(time.perf_counter() - t0) / n * 1e3:.3f} ms")
    imgs = torch.cat([batches[0]["image0"], batches[0]["image1"]], 0)
    b = bench.event_time_ms(lambda: m._backbone_hip(imgs, P), 5, 1)
    c = m._backbone_hip(imgs, P).flatten(1, 2)
    t = bench.event_time_ms(lambda: m._transformer(c[:8], c[8:], P, pe, pe), 5, 1)
    m._transformer(c[:8], c[8:], P, pe, pe)
    g0, g1 = m._feat_split
    k3 = bench.event_time_ms(lambda: ops.coarse_match(g0, g1, (60, 80), (60, 80), 0.2, 2, 0.1), 5, 1)
    print(f"alone    : backbone {b:.3f}  transformer {t:.3f}  match {k3:.3f}  (bench.py `breakdown`: 5 launches each, one batch)")
    # sustained single-stage loops: 60 back-to-back launches of the backbone alone
    b60 = bench.event_time_ms(lambda: m._backbone_hip(imgs, P), 60, 1, rounds=1) if 'rounds' in bench.event_time_ms.__code__.co_varnames else None
    print(f"backbone, 60 launches back to back: {b60}")
