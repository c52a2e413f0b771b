"""CPU study for the next round: numerical error of Winograd F(2x2,3x3) evaluated with the fp16x2-split
product (3 exact fp16xfp16 products accumulated in fp32), against fp64, next to the direct split convolution.
The transformed tiles / filters are split AFTER the fp32 transforms, as a fused kernel would do."""
import torch
import torch.nn.functional as F
torch.manual_seed(0)


def split(x):
    hi = torch.where(x.abs() >= 2.0 ** -14, x, torch.zeros_like(x)).half()
    lo = ((x - hi.float()) * 2048.0).half()
    return hi.float(), lo.float()


def split_matmul(a, b):          # a [..., M, K], b [..., K, N]; 3-product split GEMM with fp32 accumulation
    ah, al = split(a)
    bh, bl = split(b)
    return ah @ bh + (ah @ bl + al @ bh) / 2048.0


N, C, K, H, W = 2, 128, 128, 32, 32
x = torch.relu(torch.randn(N, C, H, W))
w = torch.randn(K, C, 3, 3) * (2.0 / (C * 9)) ** 0.5
ref = F.conv2d(x.double(), w.double(), padding=1)

# direct split conv (what conv_gemm_sf_same computes)
cols = F.unfold(x, 3, padding=1)                                   # [N, C*9, H*W]
direct = split_matmul(w.reshape(K, -1)[None], cols).reshape(N, K, H, W)

# Winograd F(2x2, 3x3)
Bt = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float32)
At = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)
U = torch.einsum('ij,kcjl,ml->imkc', G, w, G)                      # [4,4,K,C]  fp32 transform of the filters
xp = F.pad(x, (1, 1, 1, 1))
tiles = xp.unfold(2, 4, 2).unfold(3, 4, 2)                         # [N,C,H/2,W/2,4,4]
V = torch.einsum('ij,nchwjl,ml->imnhwc', Bt, tiles, Bt)           # [4,4,N,h,w,C]
Mm = split_matmul(V.reshape(4, 4, -1, C), U.transpose(2, 3))       # [4,4,tiles,K]   16 batched split GEMMs
Mm = Mm.reshape(4, 4, N, H // 2, W // 2, K)
Y = torch.einsum('ij,jlnhwk,ml->nkhiwm', At, Mm, At).reshape(N, K, H, W)

scale = ref.abs().max()
print(f"direct split conv : max err / max|y| = {((direct.double() - ref).abs().max() / scale).item():.3e}")
print(f"winograd F(2,3)   : max err / max|y| = {((Y.double() - ref).abs().max() / scale).item():.3e}")
print(f"fp32 direct conv  : max err / max|y| = {((F.conv2d(x, w, padding=1).double() - ref).abs().max() / scale).item():.3e}")
