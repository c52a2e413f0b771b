"""Timing of one fused encoder layer (csrc/encoder_fused.hip) at the refinement bench shape (2000 tracks x (1 + 4) views x 225
tokens) next to the five-GEMM path of the same layer: python tools/bench_encoder_fused.py [tracks]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from detectorfreesfm_amd import coarse, ops  # noqa: E402

DEV = "cuda:0"
T = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
C, WW, Vq = 128, 225, 4
g = torch.Generator().manual_seed(0)
sd = {n: torch.randn(s, generator=g) * sc for n, s, sc in (("q_proj.weight", (C, C), .12), ("k_proj.weight", (C, C), .12),
      ("v_proj.weight", (C, C), .12), ("merge.weight", (C, C), .12), ("mlp.0.weight", (2 * C, 2 * C), .09), ("mlp.2.weight", (C, 2 * C), .09))}
for nm in ("norm1", "norm2"):
    sd[nm + ".weight"], sd[nm + ".bias"] = torch.ones(C), torch.zeros(C)
w = coarse.EncoderLayerWeights(lambda n: sd[n].to(DEV), "")


def ev(fn, it=10):
    for _ in range(2):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it):
        fn()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) / it


rs = ops.SplitAct.empty_rows((T, WW), 2 * C, DEV)
qs = ops.SplitAct.empty_rows((T, Vq * WW), 2 * C, DEV)
ops.split_rows(torch.randn((T, WW, C), generator=g).to(DEV), None, out_split=rs.cols(0, C))
ops.split_rows(torch.randn((T, Vq * WW, C), generator=g).to(DEV), None, out_split=qs.cols(0, C))
ro, qo = ops.SplitAct.empty_rows((T, WW), C, DEV), ops.SplitAct.empty_rows((T, Vq * WW), C, DEV)
qm = torch.ones((T, Vq), dtype=torch.bool, device=DEV)
rows = T * WW * (1 + Vq)


def layer(self_):
    if self_:
        coarse.encoder_layer_split(w, rs, rs.cols(0, C), None, ro, 8, is_self=True)
        coarse.encoder_layer_split(w, qs, qs.cols(0, C), None, qo, 8, qm, qm, WW, WW, is_self=True)
    else:
        coarse.encoder_layer_split(w, qs, rs.cols(0, C), None, qo, 8, qm, None, WW, 1)
        coarse.encoder_layer_split(w, rs, qs.cols(0, C), None, ro, 8, None, qm, 1, WW)


for name, fused in (("fused", w.fused), ("five-GEMM", None)):
    keep, w.fused = w.fused, fused
    ts, tc = ev(lambda: layer(True)), ev(lambda: layer(False))
    w.fused = keep
    print(f"{name:10s} self layer {ts:.3f} ms, cross layer {tc:.3f} ms  ({rows} rows; 4 layers: {2 * (ts + tc):.2f} ms)")
f = w.fused
tk = ev(lambda: ops.encoder_kv(qs.cols(0, C), f))
tkm = ev(lambda: ops.encoder_kv(qs.cols(0, C), f, qm, WW))
print(f"enc_kv with the per-view source mask: {tkm:.3f} ms")
st = ops.encoder_kv(qs.cols(0, C), f)
ta = ev(lambda: ops.encoder_apply(qs.cols(0, C), f, st, Vq * WW, out_split=qo))
n = T * Vq * WW
print(f"query rows ({n}): enc_kv {tk:.3f} ms ({n * 512 / tk / 1e6:.0f} GB/s of rows read), enc_apply {ta:.3f} ms "
      f"({n * 816 / 32 * 32768 / ta / 1e9:.0f} TFLOP/s of MFMA issued, {n * 1024 / ta / 1e6:.0f} GB/s of rows)")

# stage profile (s_memtime stamps of wave 0 of every tile): medians over the tiles, in microseconds at 100 MHz
d = ops.encoder_apply(qs.cols(0, C), f, st, Vq * WW, out_split=qo, debug_stage=100)
nt = (n + 127) // 128
t = d.reshape(-1)[:nt * 32].view(nt, 16, 2).double()
t = t[..., 0] + t[..., 1] * 16777216.0
names = ["x tile load", "q GEMM", "phi(q), Z", "attention", "merge GEMM", "LayerNorm1", "MLP (24 slabs)", "LayerNorm2 + staging", "stores"]
dt = (t[:, 1:10] - t[:, 0:9]) / 100.0
med = dt.median(0).values
print("stage medians (us):", ", ".join(f"{nm} {v:.2f}" for nm, v in zip(names, med.tolist())), f"| tile {(t[:, 9] - t[:, 0]).median().item() / 100.0:.2f}")
