import sys, torch, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import test_gpu_encoder_fused as T
from detectorfreesfm_amd import ops
DEV='cuda:0'
N, L, S, qg, kg, is_self = T.CASES[6]
sd = T._weights(16); fw = T._fused(sd)
g = torch.Generator().manual_seed(106)
x = torch.randn((N, L, 128), generator=g)
xs = T._to_split(x, pad_cols=128)
x64 = xs.float().double().cpu()
ref = T._stages64(sd, x64, x64, None, None)
state = ops.encoder_kv(xs, fw)
for trial in range(4):
    out_s = ops.SplitAct.empty_rows((N, L), 128, DEV); out32 = torch.full((N, L, 128), 7.0, device=DEV)
    ops.encoder_apply(xs, fw, state, S, out_split=out_s, out=out32, debug_stage=(trial % 2) * 4)
    e = (out32.double().cpu() - ref[5]).abs()[0]
    bad = (e > 1e-3)
    rows = bad.any(1).nonzero().flatten()
    print("trial", trial, "max", e.max().item(), "bad rows", rows.numel(), rows[:20].tolist(), "cols of first bad row", bad[rows[0]].nonzero().flatten()[:16].tolist() if rows.numel() else None)
    if rows.numel():
        r = int(rows[0]); print(" got", out32[0, r, :8].tolist(), "ref", ref[5][0, r, :8].tolist())
