"""sha1 of the enc_kv apply images at the refinement shapes (reference side 2000 x 225, query side 2000 x 900 with a per-view mask):
run once per library build (DFSFM_LIB_PATH) -- equal hashes = bit-identical images."""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from detectorfreesfm_amd import coarse, ops
DEV = "cuda:0"
C, WW, Vq, T = 128, 225, 4, 2000
g = torch.Generator().manual_seed(0)
sd = {n: torch.randn(s, generator=g) * sc for n, s, sc in (("q_proj.weight", (C, C), .12), ("k_proj.weight", (C, C), .12),
      ("v_proj.weight", (C, C), .12), ("merge.weight", (C, C), .12), ("mlp.0.weight", (2 * C, 2 * C), .09), ("mlp.2.weight", (C, 2 * C), .09))}
for nm in ("norm1", "norm2"):
    sd[nm + ".weight"], sd[nm + ".bias"] = torch.ones(C), torch.zeros(C)
w = coarse.EncoderLayerWeights(lambda n: sd[n].to(DEV), "")
rs = ops.SplitAct.empty_rows((T, WW), 2 * C, DEV)
qs = ops.SplitAct.empty_rows((T, Vq * WW), 2 * C, DEV)
ops.split_rows(torch.randn((T, WW, C), generator=g).to(DEV), None, out_split=rs.cols(0, C))
ops.split_rows(torch.randn((T, Vq * WW, C), generator=g).to(DEV), None, out_split=qs.cols(0, C))
qm = (torch.rand((T, Vq), generator=g) > 0.3).to(DEV)
f = w.fused
import numpy as np
keep = np.zeros(ops.ENC_KV_IMAGE, dtype=bool)          # the image's written bytes: fragment t of pair p comes from lanes with (lane & 31) >> 4 == t
for p_ in range(4):
    for t_ in range(2):
        for hl in range(2):
            for lane in range(64):
                if ((lane & 31) >> 4) == t_:
                    o = ((p_ * 2 + t_) * 2 + hl) * 1024 + lane * 16
                    keep[o:o + 16] = True
keep[16384:16384 + 512] = True
for name, st in (("ref", ops.encoder_kv(rs.cols(0, C), f)), ("query", ops.encoder_kv(qs.cols(0, C), f)),
                 ("query masked", ops.encoder_kv(qs.cols(0, C), f, qm, WW)),
                 ("short S=40", ops.encoder_kv(ops.SplitAct(qs.hi.view(-1, 40, 2 * C)[:1000], qs.lo.view(-1, 40, 2 * C)[:1000], 2 * C).cols(0, C), f))):
    raw = st.cpu().numpy()[:, keep]
    print(name, tuple(st.shape), hashlib.sha1(raw.tobytes()).hexdigest())
