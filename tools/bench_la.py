"""Linear-attention timing at the coarse (16 x 4800 x 8 x 32) and refinement (2000 x 900 / 225 x 8 x 16) shapes."""
import sys, torch
sys.path.insert(0, '.')
from detectorfreesfm_amd import ops
dev = 'cuda:0'
g = torch.Generator().manual_seed(0)
def t(fn, it=20):
    fn(); fn(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / it * 1e3
q, k, v = (torch.randn((16, 4800, 8, 32), generator=g).to(dev) for _ in range(3))
print("coarse D=32 16x4800      : %.1f us" % t(lambda: ops.linear_attention(q, k, v, out_split=True)))
q, k, v = (torch.randn((2000, 900, 8, 16), generator=g).to(dev) for _ in range(3))
print("refine D=16 2000x900 self: %.1f us" % t(lambda: ops.linear_attention(q, k, v, out_split=True)))
k2, v2 = (torch.randn((2000, 225, 8, 16), generator=g).to(dev) for _ in range(2))
print("refine D=16 900<-225     : %.1f us" % t(lambda: ops.linear_attention(q, k2, v2, out_split=True)))
