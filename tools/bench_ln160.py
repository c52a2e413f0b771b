"""LayerNorm-fused 256-channel linear: 128-row vs 160-row tiles at the coarse transformer's shapes (DFSFM_LN160 = 0 / 1)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from detectorfreesfm_amd import ops
dev = "cuda:0"
g = torch.Generator().manual_seed(0)


def timeit(fn, n=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / n


for rows, K, what in ((38400, 256, "cross merge"), (38400, 512, "cross mlp.2"), (76800, 256, "self merge"), (76800, 512, "self mlp.2"),
                      (4800, 256, "one image"), (19200, 512, "4 images")):
    x = ops.SplitAct.empty_rows((rows,), K, dev)
    ops.split_rows(torch.randn((rows, K), generator=g).to(dev), out_split=x)
    pw = ops.PackedDense((torch.randn((256, K), generator=g) / K ** 0.5).to(dev), None)
    gamma, beta = torch.ones(256, device=dev), torch.zeros(256, device=dev)
    res = ops.SplitAct.empty_rows((rows,), 256, dev)
    ops.split_rows(torch.randn((rows, 256), generator=g).to(dev), out_split=res)
    out = ops.SplitAct.empty_rows((rows,), 256, dev)
    line = f"{what:12s} rows {rows:6d} K {K}:"
    for mode in ("0", "1", None):
        if mode is None:
            os.environ.pop("DFSFM_LN160", None)
        else:
            os.environ["DFSFM_LN160"] = mode
        us = timeit(lambda: ops.linear_ln(x, pw, gamma, beta, residual=res, out_split=out))
        line += f"  {'128-row' if mode == '0' else '160-row' if mode == '1' else 'auto'} {us:6.1f} us"
    print(line)
