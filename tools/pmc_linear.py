"""One linear layer, a handful of launches: target of rocprofv3 --pmc passes.  usage: pmc_linear.py rows K N [relu|f32|ln]"""
import sys, torch
sys.path.insert(0, '.')
from detectorfreesfm_amd import ops
rows, K, N = (int(a) for a in sys.argv[1:4])
mode = sys.argv[4] if len(sys.argv) > 4 else "relu"
dev = 'cuda:0'
g = torch.Generator().manual_seed(0)
xs = ops.SplitAct.empty_rows((rows,), K, dev)
ops.split_rows(torch.randn((rows, K), generator=g).to(dev), None, out_split=xs)
pw = ops.PackedDense((torch.randn((N, K), generator=g) * K ** -0.5).to(dev))
out = torch.empty((rows, N), device=dev)
for _ in range(5):
    if mode == "f32":
        ops.linear(xs, pw, out=out)
    else:
        ops.linear(xs, pw, relu=True, out_split=True)
torch.cuda.synchronize()
