"""How far is the fp32 ORACLE from an exact (float64) evaluation of the same LoFTR at the reference's production frame
sizes?  (tests/test_gpu_e2e.py::test_loftr_production_frame_sizes_vs_oracle, VERDICT r03 "what's missing" #2.)

north_star's 1e-4 on confidences compares two fp32-class evaluations.  At 640x480 they agree to ~1e-5; at 1600x1064 a
confidence is a ratio of sums over 26 600 competitors of exp(sim / 0.1) with |sim| in the hundreds, so the fp32 rounding of the
features (1 ulp = 6e-8 relative) moves a confidence by ~1e-5 ... 1e-4 on its own.  This script measures that on the CPU:
the oracle in fp32 vs the same functions in float64, per matched entry.  Run:  python tools/studies/loftr_hires_noise_study.py H W
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from detectorfreesfm_amd import synth  # noqa: E402
from detectorfreesfm_amd.config import loftr_coarse_only_config  # noqa: E402
from detectorfreesfm_amd.params import loftr_param_spec, planted_loftr_state_dict  # noqa: E402
from oracle import restate  # noqa: E402


def features(sd, cfg, data, dtype):
    sd = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sd.items()}
    img = torch.cat([data["image0"], data["image1"]], 0).to(dtype)
    c, _ = restate.resnet_fpn_8_2(sd, "backbone.", img, False)
    c0, c1 = c.split(1)
    hw = tuple(c0.shape[2:])
    pe = restate.position_encoding_sine(cfg["coarse"]["d_model"], temp_bug_fix=cfg["coarse"]["temp_bug_fix"]).to(dtype)
    f0 = (c0 + pe[:, :, :hw[0], :hw[1]]).flatten(2).transpose(1, 2)
    f1 = (c1 + pe[:, :, :hw[0], :hw[1]]).flatten(2).transpose(1, 2)
    return restate.coarse_transformer(sd, "loftr_coarse.", f0, f1, cfg["coarse"]["layer_names"], cfg["coarse"]["nhead"]), hw


def conf_at(f0, f1, temperature, ii, jj, chunk=2048):
    """dual-softmax confidences at the entries (ii, jj) without the dense matrix: chunked log-sum-exps in f0's dtype."""
    C = f0.shape[-1]
    a, b = f0[0] / C ** 0.5, f1[0] / C ** 0.5
    L, S = a.shape[0], b.shape[0]
    row_lse = torch.empty(L, dtype=a.dtype)
    col_m = torch.full((S,), -float("inf"), dtype=a.dtype)
    col_s = torch.zeros(S, dtype=a.dtype)
    for lo in range(0, L, chunk):
        sim = (a[lo:lo + chunk] @ b.T) / temperature
        row_lse[lo:lo + chunk] = torch.logsumexp(sim, 1)
        m = torch.maximum(col_m, sim.max(0)[0])
        col_s = col_s * torch.exp(col_m - m) + torch.exp(sim - m).sum(0)
        col_m = m
    col_lse = col_m + torch.log(col_s)
    s = (a[ii] * b[jj]).sum(-1) / temperature
    return torch.exp(s - row_lse[ii]) * torch.exp(s - col_lse[jj]), s


def main():
    H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1064, 1600)
    torch.set_num_threads(int(os.environ.get("STUDY_THREADS", "8")))
    cfg = loftr_coarse_only_config(0.2)
    sd = planted_loftr_state_dict(loftr_param_spec(cfg), 0)
    data = synth.coarse_pair_batch(1, H, W, seed=1300)
    with torch.no_grad():
        t0 = time.time()
        o = restate.loftr_coarse_forward(sd, cfg, data, with_fine_backbone=False)
        ii, jj, c32 = o["i_ids"], o["j_ids"], o["mconf"]
        print(f"{W}x{H}: oracle fp32 {len(ii)} matches in {time.time() - t0:.0f} s; |feat| max {o['feat_c0'].abs().max():.1f}")
        (g0, g1), hw = features(sd, cfg, data, torch.float64)
        c64, s64 = conf_at(g0, g1, cfg["match_coarse"]["dsmax_temperature"], ii, jj)
        d = (c32.double() - c64).abs()
        print(f"  fp32 oracle vs float64 evaluation at its own matches: max {d.max():.3e}, mean {d.mean():.3e}, "
              f"> 1e-4: {(d > 1e-4).sum().item()}, > 5e-5: {(d > 5e-5).sum().item()}; signed mean {(c32.double() - c64).mean():+.3e}")
        print(f"  similarity / temperature at the matches: median {s64.median():.1f}, max {s64.max():.1f}")
        # the fp32 FEATURES through an exact dual-softmax: how much of the gap is the matching stage's own fp32 sums
        cmix, _ = conf_at(o["feat_c0"].double(), o["feat_c1"].double(), cfg["match_coarse"]["dsmax_temperature"], ii, jj)
        dm = (c32.double() - cmix).abs()
        print(f"  fp32 oracle vs (its fp32 features -> float64 dual-softmax): max {dm.max():.3e}, mean {dm.mean():.3e}")
        df = (cmix - c64).abs()
        print(f"  (fp32 features vs float64 features) through the same float64 dual-softmax: max {df.max():.3e}, mean {df.mean():.3e}")
        torch.save({"i": ii, "j": jj, "c32": c32, "c64": c64.float(), "c64d": c64}, f"/tmp/hires_noise_{W}x{H}.pt")


if __name__ == "__main__":
    main()
