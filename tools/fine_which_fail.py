"""Which tracks / views deviate in a two-workgroups-per-CU experiment build of fine_match (reference: the same library run
in chunks of 250 tracks, i.e. one workgroup per CU at a time)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from detectorfreesfm_amd import ops
dev = "cuda:0"
T, Vq, W, C = 2000, 4, 15, 128
g = torch.Generator().manual_seed(3)
ref = torch.randn((T, W * W, C), generator=g).to(dev)
qry = (0.7 * ref[:, None].cpu() + torch.randn((T, Vq, W * W, C), generator=g)).to(dev)
mask = torch.ones((T, Vq), dtype=torch.bool, device=dev)
mov = torch.ones((T,), dtype=torch.bool, device=dev)
rs = ops.SplitAct.empty_rows((T, W * W), C, dev)
qs = ops.SplitAct.empty_rows((T, Vq, W * W), C, dev)
ops.split_rows(ref.view(-1, C), out_split=ops.SplitAct(rs.hi.view(-1, C), rs.lo.view(-1, C), C))
ops.split_rows(qry.view(-1, C), out_split=ops.SplitAct(qs.hi.view(-1, C), qs.lo.view(-1, C), C))
good = {k: [] for k in ("coords", "std", "best_index")}
for lo in range(0, T, 250):
    o = ops.fine_match(ops.SplitAct(rs.hi[lo:lo + 250], rs.lo[lo:lo + 250], C), ops.SplitAct(qs.hi[lo:lo + 250], qs.lo[lo:lo + 250], C),
                       mask[lo:lo + 250], mov[lo:lo + 250], W, 7)
    torch.cuda.synchronize()
    for k in good:
        good[k].append(o[k].clone())
good = {k: torch.cat(v) for k, v in good.items()}
for run in range(3):
    o = ops.fine_match(rs, qs, mask, mov, W, 7)
    dc = (o["coords"] - good["coords"]).abs().amax(-1)          # [T, Vq]
    bad = (dc > 0).nonzero()
    bi = (o["best_index"] != good["best_index"]).nonzero()[:, 0].tolist()
    print(f"run {run}: {bad.shape[0]} (track, view) pairs deviate in {len(set(bad[:, 0].tolist()))} tracks; best_index differs in {len(bi)} tracks")
    print("   tracks:", sorted(set(bad[:, 0].tolist()))[:40])
    print("   views of the first 12:", [(int(t), int(v), f"{float(dc[t, v]):.1e}") for t, v in bad[:12].tolist()])
    ds = (o['std'] - good['std']).abs()
    bs = (ds > 0).nonzero()
    print("   std deviating:", bs.shape[0], " nan:", int(torch.isnan(o['coords']).sum()),
          [(int(t), int(v), f"{float(o['std'][t, v]):.4f} vs {float(good['std'][t, v]):.4f}") for t, v in bs[:10].tolist()])
    print("   left_norm deviating:", int(((o['left_norm'] - ops.fine_match(rs, qs, mask, mov, W, 7)['left_norm']).abs() > 0).sum()))
