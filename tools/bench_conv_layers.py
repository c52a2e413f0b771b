"""Per-layer timing of dfsfm_conv2d_nhwc_f32 at the shapes of both CNNs and the encoder linears."""
import sys, torch
sys.path.insert(0, '.')
from detectorfreesfm_amd import ops

def t(fn, it=10):
    fn(); fn(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / it

dev = 'cuda:0'
layers = [  # name, N, H, W, Cin, Cout, k, stride, pad
    ("stem 7x7s2 1->128", 16, 480, 640, 1, 128, 7, 2, 3),
    ("l1 3x3 128->128 @240x320", 16, 240, 320, 128, 128, 3, 1, 1),
    ("l2.0 3x3s2 128->196", 16, 240, 320, 128, 196, 3, 2, 1),
    ("l2 3x3 196->196 @120x160", 16, 120, 160, 196, 196, 3, 1, 1),
    ("l2.0 down 1x1s2 128->196", 16, 240, 320, 128, 196, 1, 2, 0),
    ("l3.0 3x3s2 196->256", 16, 120, 160, 196, 256, 3, 2, 1),
    ("l3 3x3 256->256 @60x80", 16, 60, 80, 256, 256, 3, 1, 1),
    ("l3out 1x1 256->256", 16, 60, 80, 256, 256, 1, 1, 0),
    ("s2d conv1_1 3->64 @35", 10000, 35, 35, 3, 64, 3, 1, 1),
    ("s2d conv1_2 64->64 @35", 10000, 35, 35, 64, 64, 3, 1, 1),
    ("s2d conv2_1 64->128 @18", 10000, 18, 18, 64, 128, 3, 1, 1),
    ("s2d conv2_2 128->128 @18", 10000, 18, 18, 128, 128, 3, 1, 1),
    ("s2d conv3_1 128->256 @9", 10000, 9, 9, 128, 256, 3, 1, 1),
    ("s2d conv3_x 256->256 @9", 10000, 9, 9, 256, 256, 3, 1, 1),
    ("s2d adap0 5x5 64->128 19->15", 10000, 19, 19, 64, 128, 5, 1, 0),
    ("s2d adap1 5x5 64->128 @9", 10000, 9, 9, 64, 128, 5, 1, 2),
    ("lin qkv 76800x256->768", 1, 1, 76800, 256, 768, 1, 1, 0),
    ("lin mlp0 76800x512->512", 1, 1, 76800, 512, 512, 1, 1, 0),
    ("lin mlp2 76800x512->256", 1, 1, 76800, 512, 256, 1, 1, 0),
    ("lin fine mlp0 1.8Mx256->256", 1, 1, 1800000, 256, 256, 1, 1, 0),
]
split = len(sys.argv) > 1 and sys.argv[1] in ("split", "same")
same = len(sys.argv) > 1 and sys.argv[1] == "same"     # tap-padded weights -> activation-reuse kernel
for name, N, H, W, Cin, Cout, k, s, p in layers:
    x = torch.randn((N, H, W, Cin), device=dev)
    w = torch.randn((Cout, Cin, k, k), device=dev) * 0.05
    use_split = split and Cin % 4 == 0
    if use_split:      # activations arrive / leave as split fp16 planes (v2 LDS-DMA kernel)
        cp = (Cin + 7) // 8 * 8
        hi = torch.zeros((N, H, W, cp), dtype=torch.float16, device=dev); lo = torch.zeros_like(hi)
        hi[..., :Cin] = x.half(); lo[..., :Cin] = ((x - x.half().float()) * 2048).half()
        xin = ops.SplitAct(hi, lo, Cin)
        tp = same and s == 1 and k > 1 and p == k // 2
        pw = ops.PackedDense(w, torch.zeros(Cout, device=dev), cin_pad=cp, tap_padded=tp)
        name = name + (" [same]" if tp else "")
        ms = t(lambda: ops.conv2d_nhwc(xin, pw, s, p, relu=True, out_split=True))
    else:
        pw = ops.PackedDense(w, torch.zeros(Cout, device=dev))
        out = torch.empty((N, (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1, Cout), device=dev)
        ms = t(lambda: ops.conv2d_nhwc(x, pw, s, p, relu=True, out=out))
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    fl = 2.0 * N * Ho * Wo * Cout * Cin * k * k
    byts = (x.numel() + N * Ho * Wo * Cout) * 4
    print(f"{name:41s} {ms:8.3f} ms  {fl/ms/1e9:7.1f} TF-eff  {byts/ms/1e6:7.0f} GB/s  ({fl/1e9:.1f} GFLOP)", flush=True)
    del x, w, pw
