"""Linear-layer shapes of both transformers on dfsfm_conv2d_nhwc_f32 (1x1 case: the 128x128 two-workgroups-per-CU schedule;
the 512-thread schedule it was A/B'd against in r02 has left the library).  usage: python tools/bench_linear.py"""
import sys, torch
sys.path.insert(0, '.')
from detectorfreesfm_amd import ops
dev = 'cuda:0'
g = torch.Generator().manual_seed(0)
def t_ms(fn, it=20):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / it
shapes = [  # (label, rows, K, N, mode)   mode: f32 out / split relu / ln
    ("coarse qkv", 76800, 256, 768, "f32"), ("coarse mlp.0", 76800, 512, 512, "relu"),
    ("coarse q", 38400, 256, 256, "f32"), ("coarse kv", 38400, 256, 512, "f32"),
    ("refine qkv (qry)", 1800000, 128, 384, "f32"), ("refine merge+LN (qry)", 1800000, 128, 128, "ln"),
    ("refine mlp.0 (qry)", 1800000, 256, 256, "relu"), ("refine mlp.2+LN (qry)", 1800000, 256, 128, "ln"),
    ("refine qkv (ref)", 450000, 128, 384, "f32"), ("refine mlp.0 (ref)", 450000, 256, 256, "relu"),
]
tot = 0
for label, rows, K, N, mode in shapes:
    x = torch.randn((rows, K), generator=g).to(dev)
    xs = ops.SplitAct.empty_rows((rows,), K, dev)
    ops.split_rows(x, None, out_split=xs)
    pw = ops.PackedDense((torch.randn((N, K), generator=g) * K ** -0.5).to(dev))
    if mode == "f32":
        out = torch.empty((rows, N), device=dev)
        fn = lambda: ops.linear(xs, pw, out=out)
        byts = rows * (K + N) * 4
    elif mode == "relu":
        fn = lambda: ops.linear(xs, pw, relu=True, out_split=True)
        byts = rows * (K + N) * 4
    else:
        gam, bet = torch.ones(N, device=dev), torch.zeros(N, device=dev)
        res = ops.SplitAct.empty_rows((rows,), N, dev)
        ops.split_rows(torch.randn((rows, N), generator=g).to(dev), None, out_split=res)
        o = ops.SplitAct.empty_rows((rows,), N, dev)
        fn = lambda: ops.linear_ln(xs, pw, gam, bet, residual=res, out_split=o)
        byts = rows * (K + 2 * N) * 4
    ms = t_ms(fn)
    tot += ms
    print(f"{label:26s} rows {rows:8d} K {K:4d} N {N:4d}: {ms*1e3:8.1f} us  {2.0*rows*K*N/ms/1e9:7.1f} TF-eff  {byts/ms/1e6:7.1f} GB/s")
    del x, xs, pw
print(f"sum {tot*1e3:.1f} us")
