"""Timing of the device keypoint-merge stage at scene scale (300 images, 2000 pairs x 1500 matches = 3 M rows)
next to the CPU restatement of the reference's loops on a 1/10 sample."""
import sys, time, torch, numpy as np
sys.path.insert(0, '.')
from detectorfreesfm_amd import ops
g = torch.Generator().manual_seed(9)
n_img, n_pairs, per = 300, 2000, 1500
a = torch.randint(0, n_img, (n_pairs,), generator=g)
b = (a + 1 + torch.randint(0, n_img - 1, (n_pairs,), generator=g)) % n_img
M = n_pairs * per
rows = torch.empty((M, 5))
rows[:, :4] = (torch.randint(0, 80, (M, 4), generator=g) * 8).float() * 1.31
rows[:, 4] = torch.rand(M, generator=g) * 0.8 + 0.2
i0, i1 = a.repeat_interleave(per).int(), b.repeat_interleave(per).int()
dev = 'cuda:0'
r, x0, x1 = rows.to(dev), i0.to(dev), i1.to(dev)
for _ in range(2):
    out = ops.merge_keypoints(r, x0, x1, n_img)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    out = ops.merge_keypoints(r, x0, x1, n_img)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
print(f"device: {M} match rows, {out[0].shape[0]} keypoints: {dt * 1e3:.2f} ms  ({2 * M / dt / 1e9:.2f} G endpoints/s)")
if len(sys.argv) > 1 and sys.argv[1] == "cpu":
    from oracle import restate_merge as rm
    m = M // 10
    t0 = time.perf_counter()
    rm.merge_keypoints(rows[:m].numpy(), i0[:m].numpy(), i1[:m].numpy(), n_img)
    print(f"cpu restatement (numpy, 1 thread): {m} rows in {time.perf_counter() - t0:.2f} s")
