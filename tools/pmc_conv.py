"""One conv layer, a handful of launches: the workload for rocprofv3 --pmc passes on the conv kernels.
usage: pmc_conv.py <same|split> [N H W Cin Cout k]"""
import sys, torch
sys.path.insert(0, '.')
from detectorfreesfm_amd import ops
mode = sys.argv[1] if len(sys.argv) > 1 else "same"
N, H, W, Cin, Cout, k = (int(a) for a in sys.argv[2:8]) if len(sys.argv) >= 8 else (16, 240, 320, 128, 128, 3)
dev = 'cuda:0'
x = torch.randn((N, H, W, Cin), device=dev)
w = torch.randn((Cout, Cin, k, k), device=dev) * 0.05
cp = (Cin + 7) // 8 * 8
hi = torch.zeros((N, H, W, cp), dtype=torch.float16, device=dev); lo = torch.zeros_like(hi)
hi[..., :Cin] = x.half(); lo[..., :Cin] = ((x - x.half().float()) * 2048).half()
xin = ops.SplitAct(hi, lo, Cin)
pw = ops.PackedDense(w, torch.zeros(Cout, device=dev), cin_pad=cp, tap_padded=(mode == "same"))
for _ in range(5):
    y = ops.conv2d_nhwc(xin, pw, 1, k // 2, relu=True, out_split=True)
torch.cuda.synchronize()
