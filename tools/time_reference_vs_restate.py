"""BASELINE.md section 2 asks for the REAL reference modules timed on the build box's host cores; ``bench.py``'s ``cpu_baseline`` times
``oracle.restate`` (kind "port") because /root/reference does not travel to the GPU box.  This script runs both side by side where
the reference tree exists (this build container) on the same seeded weights / inputs and prints the ratio, so that the port's
figure can be read as the reference's (VERDICT r04 #10b).  Test infrastructure: imports ``oracle``.

usage: python tools/time_reference_vs_restate.py [threads]   ->  profiles/r05_reference_vs_restate_cpu.txt"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from detectorfreesfm_amd import synth  # noqa: E402
from detectorfreesfm_amd.config import loftr_coarse_only_config, multiview_refinement_config  # noqa: E402
from detectorfreesfm_amd.params import loftr_param_spec, multiview_param_spec, planted_loftr_state_dict, random_state_dict  # noqa: E402
from oracle import ref_import, restate  # noqa: E402

threads = int(sys.argv[1]) if len(sys.argv) > 1 else min(8, torch.get_num_threads())
torch.set_num_threads(threads)


def best_of(fn, n=3):
    fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return min(ts), sorted(ts)[len(ts) // 2]


lines = [f"host: {os.cpu_count()} logical CPUs, torch threads {threads}, torch {torch.__version__}"]
with torch.no_grad():
    cfg = loftr_coarse_only_config(0.2)
    sd = planted_loftr_state_dict(loftr_param_spec(cfg), 0)
    data = synth.coarse_pair_batch(1, 480, 640, seed=1000)
    LoFTR, _ = ref_import.import_loftr()
    ref = LoFTR(cfg).eval()
    ref.load_state_dict(sd, strict=True)
    t_ref = best_of(lambda: ref(dict(data)))
    t_port = best_of(lambda: restate.loftr_coarse_forward(sd, cfg, data))
    d = dict(data)
    ref(d)
    o = restate.loftr_coarse_forward(sd, cfg, data)
    same = torch.equal(d["i_ids"], o["i_ids"]) and torch.equal(d["j_ids"], o["j_ids"]) and torch.equal(d["mconf"], o["mconf"])
    lines.append(f"coarse, one 640x480 pair (planted weights, thr 0.2, {int(o['i_ids'].numel())} matches, tables bit-identical: {same}):")
    lines.append(f"  real LoFTR module            min {t_ref[0]:.3f} s  median {t_ref[1]:.3f} s  -> {1 / t_ref[0]:.3f} pairs/s")
    lines.append(f"  oracle.restate (the port)    min {t_port[0]:.3f} s  median {t_port[1]:.3f} s  -> {1 / t_port[0]:.3f} pairs/s")
    lines.append(f"  port / reference time ratio  {t_port[0] / t_ref[0]:.3f}")

    rcfg = multiview_refinement_config()
    rsd = random_state_dict(multiview_param_spec(rcfg), 1)
    T = 100
    bag = synth.refine_bag(T=T, V=5, H=480, W=640, seed=2000)
    MM = ref_import.import_multiview_matcher()
    mm = MM(rcfg, test=True).eval()
    mm.load_state_dict(rsd, strict=True)

    def run_ref():
        b = {k: (v.clone() if isinstance(v, torch.Tensor) else [x.clone() for x in v] if isinstance(v, list) else v) for k, v in bag.items()}
        mm(b)
        return b
    r_ref = best_of(run_ref, 2)
    r_port = best_of(lambda: restate.multiview_matcher_forward(rsd, rcfg, bag), 2)
    b = run_ref()
    o = restate.multiview_matcher_forward(rsd, rcfg, bag)
    dq = float((b["query_points_refined"] - o["query_points_refined"]).abs().max())
    lines.append(f"refinement, one bag of {T} tracks x 5 views (max |query point diff| real vs port: {dq:.1e} px; RoIAlign is the builder's "
                 "restatement in BOTH runs, SURVEY 8c):")
    lines.append(f"  real MultiviewMatcher module min {r_ref[0]:.3f} s  -> {T / r_ref[0]:.1f} tracks/s")
    lines.append(f"  oracle.restate (the port)    min {r_port[0]:.3f} s  -> {T / r_port[0]:.1f} tracks/s")
    lines.append(f"  port / reference time ratio  {r_port[0] / r_ref[0]:.3f}")
text = "\n".join(lines)
print(text)
with open(os.path.join(ROOT, "profiles", "r05_reference_vs_restate_cpu.txt"), "w") as fh:
    fh.write("# tools/time_reference_vs_restate.py (build container, CPU only): the real reference modules beside the oracle port that\n"
             "# bench.py's cpu_baseline times on the GPU box's host\n" + text + "\n")
