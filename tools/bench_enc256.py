"""Times dfsfm_encoder256_apply_f32 (and the state kernels) at the coarse transformer's shapes.  With DFSFM_LIB_PATH pointing at
an ablation build (temporary switches of round 4, since removed from csrc/encoder256.hip) the numbers are the timing breakdown of DESIGN.md section 3 (results are wrong)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import event_time_ms  # noqa: E402
from detectorfreesfm_amd import coarse, ops  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    C = 256
    wsd = {n: torch.randn(sh, generator=g) * sc for n, sh, sc in (("q_proj.weight", (C, C), .09), ("k_proj.weight", (C, C), .09),
           ("v_proj.weight", (C, C), .09), ("merge.weight", (C, C), .09), ("mlp.0.weight", (2 * C, 2 * C), .06),
           ("mlp.2.weight", (C, 2 * C), .06))}
    for nm in ("norm1", "norm2"):
        wsd[nm + ".weight"], wsd[nm + ".bias"] = torch.ones(C), torch.zeros(C)
    lw = coarse.EncoderLayerWeights(lambda n: wsd[n].to(dev), "")
    out = []
    for N in (8, 16):
        xs = ops.SplitAct.empty_rows((N, 4800), 2 * C, dev)
        ops.split_rows(torch.randn((N, 4800, C), generator=g).to(dev), None, out_split=xs.cols(0, C))
        xo = ops.SplitAct.empty_rows((N, 4800), C, dev)
        kv = ops.linear(xs.cols(0, C), lw.pkv).view(N, 4800, 2 * C)
        st = ops.encoder256_state(kv[..., :C], kv[..., C:])
        ms = event_time_ms(lambda: ops.encoder256_apply(xs.cols(0, C), lw.fused256, st, 4800, out_split=xo))
        tiles = N * 4800 // 64
        out.append(f"N={N}: apply {ms * 1000:.1f} us ({tiles} tiles, {ms * 1000 / -(-tiles // 256):.1f} us per round of tiles)")
    print(os.environ.get("DFSFM_LIB_PATH", "product build"), "|", " | ".join(out))
    if len(sys.argv) > 1 and sys.argv[1] == "profile":
        # stage profile: s_memtime stamps of wave 0 of every tile (shader cycles), averaged over the tiles of the N = 16 launch
        d = ops.encoder256_apply(xs.cols(0, C), lw.fused256, st, 4800, out_split=xo, debug_stage=100)
        torch.cuda.synchronize()
        t = d.view(-1)[:tiles * 32].view(tiles, 16, 2).double().cpu()
        stamps = t[..., 0] + t[..., 1] * float(1 << 24)
        names = ["wait for the x tile", "q GEMM (16 slabs)", "phi(q), Z", "attention (KV image loads + 48 MFMAs)", "merge GEMM (16 slabs)",
                 "LayerNorm1", "MLP (96 slabs)", "LayerNorm2 + staging", "row stores + next tile's loads"]
        dt = (stamps[:, 1:10] - stamps[:, 0:9])
        ok = (dt > 0).all(1) & (dt < 1e7).all(1)
        dt = dt[ok]
        print(f"stage profile over {int(ok.sum())} tiles (shader cycles, mean / median):")
        for k, nm in enumerate(names):
            print(f"  {nm:45s} {dt[:, k].mean():9.0f} {dt[:, k].median():9.0f}")
        tot = stamps[ok][:, 9] - stamps[ok][:, 0]
        print(f"  {'tile':45s} {tot.mean():9.0f} {tot.median():9.0f}")


if __name__ == "__main__":
    main()
