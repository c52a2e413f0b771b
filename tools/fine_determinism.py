"""Run-to-run reproducibility of fine_match at BASELINE configs[2] (2000 tracks x 4 views): N runs, tracks whose outputs differ
from run 0 (any bit of coords / std / best_index), and the timing.  With DFSFM_LIB_PATH = an experiment build
(the temporary build switches of round 4 have all left csrc/fine_match.hip) this was the experiment matrix of profiles/r04_fine_match_root_cause.txt;
the cause was found in r06 by tools/studies/fine_bisect.py, which builds its variants from the assembly (profiles/r06_fine_match_bisect.txt)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from detectorfreesfm_amd import ops

dev = "cuda:0"
T, Vq, W, C = int(os.environ.get("FINE_T", "2000")), 4, 15, 128
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 40
g = torch.Generator().manual_seed(3)
ref = torch.randn((T, W * W, C), generator=g).to(dev)
qry = (0.7 * ref[:, None].cpu() + torch.randn((T, Vq, W * W, C), generator=g)).to(dev)
mask = torch.ones((T, Vq), dtype=torch.bool, device=dev)
mov = torch.ones((T,), dtype=torch.bool, device=dev)
rs = ops.SplitAct.empty_rows((T, W * W), C, dev)
qs = ops.SplitAct.empty_rows((T, Vq, W * W), C, dev)
ops.split_rows(ref.view(-1, C), out_split=ops.SplitAct(rs.hi.view(-1, C), rs.lo.view(-1, C), C))
ops.split_rows(qry.view(-1, C), out_split=ops.SplitAct(qs.hi.view(-1, C), qs.lo.view(-1, C), C))
first = ops.fine_match(rs, qs, mask, mov, W, 7)
bad_runs, bad_tracks, worst = 0, set(), 0.0
for _ in range(runs):
    o = ops.fine_match(rs, qs, mask, mov, W, 7)
    bits = lambda t: t.contiguous().view(torch.int32)
    d = (bits(o["coords"]) != bits(first["coords"])).flatten(1).any(1) | (bits(o["std"]) != bits(first["std"])).flatten(1).any(1) | \
        (o["best_index"] != first["best_index"])
    n = int(d.sum())
    if n:
        bad_runs += 1
        bad_tracks |= set(d.nonzero()[:, 0].tolist())
        worst = max(worst, float((o["coords"] - first["coords"]).abs().max()))
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20):
    ops.fine_match(rs, qs, mask, mov, W, 7)
b.record()
torch.cuda.synchronize()
def tsub(n):       # residency probe: 256 tracks = one workgroup per CU, 512 = two
    n = min(n, T)
    sub = (ops.SplitAct(rs.hi[:n], rs.lo[:n], C), ops.SplitAct(qs.hi[:n], qs.lo[:n], C), mask[:n], mov[:n])
    for _ in range(3):
        ops.fine_match(*sub, W, 7)
    torch.cuda.synchronize()
    x, y = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    x.record()
    for _ in range(50):
        ops.fine_match(*sub, W, 7)
    y.record()
    torch.cuda.synchronize()
    return x.elapsed_time(y) / 50 * 1000
t256, t512 = tsub(256), tsub(512)
ref32 = ops.fine_match(rs.float(), qs.float(), mask, mov, W, 7)       # the fp32-input entry point on the same values
print(f"{os.environ.get('DFSFM_LIB_PATH', 'product build').split('/')[-1]:32s} finite {bool(torch.isfinite(first['coords']).all())}, vs fp32 entry "
      f"{float((first['coords'] - ref32['coords']).abs().max()):.1e}; {runs} runs: {bad_runs} differ from run 0, {len(bad_tracks)} distinct tracks, "
      f"max |coords diff| {worst:.2e}; {a.elapsed_time(b) / 20:.3f} ms per call; 256 / 512 tracks: {t256:.1f} / {t512:.1f} us "
      f"({'two workgroups per CU' if t512 < 1.6 * t256 else 'one workgroup per CU'})")
