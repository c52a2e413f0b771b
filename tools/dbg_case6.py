import sys, torch, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import test_gpu_encoder_fused as T
from detectorfreesfm_amd import ops
DEV='cuda:0'
for ci in (0, 6):
    N, L, S, qg, kg, is_self = T.CASES[ci]
    sd = T._weights(10 + ci); fw = T._fused(sd)
    g = torch.Generator().manual_seed(100 + ci)
    x = torch.randn((N, L, 128), generator=g)
    xs = T._to_split(x, pad_cols=128)
    state = ops.encoder_kv(xs, fw)
    for trial in range(3):
        out_s = ops.SplitAct.empty_rows((N, L), 128, DEV); out32 = torch.full((N, L, 128), 7.0, device=DEV)
        out_s.hi.fill_(9.0); out_s.lo.fill_(9.0)
        ops.encoder_apply(xs, fw, state, S, out_split=out_s, out=out32, debug_stage=4 if trial else 0)
        d = (out_s.float() - out32).abs()[0]
        bad = d > 0
        rows = bad.any(1).nonzero().flatten()
        print("case", ci, "trial", trial, "mismatch rows", rows.numel(), rows[:12].tolist(), "cols", bad[rows[0]].nonzero().flatten()[:20].tolist() if rows.numel() else None)
        if rows.numel():
            r = int(rows[0]); c = int(bad[r].nonzero()[0])
            print("   hi", out_s.hi[0, r, c:c+4].tolist(), "lo", out_s.lo[0, r, c:c+4].tolist(), "f32", out32[0, r, c:c+4].tolist())
