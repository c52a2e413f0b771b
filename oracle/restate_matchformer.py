"""TEST INFRASTRUCTURE ONLY -- CPU restatement (oracle) of the MatchFormer-LA coarse matcher.

Functional fp32 torch-CPU restatement on a plain ``state_dict`` of
  Matchformer.forward (fine.enable = False)       third_party/MatchFormer/model/matchformer.py:21-61
  Matchformer_LA_large.forward                      third_party/MatchFormer/model/backbone/match_LA_large.py:176-255
  AttentionBlock / Block / Attention / Mlp / DWConv / PatchEmbed / Positional          ... :15-174
  CoarseMatching (dual_softmax, padding masks)      third_party/MatchFormer/model/backbone/coarse_matching.py:60-228
(paths relative to /root/reference).  Pinned bit-for-bit against the real module (oracle/ref_import.import_matchformer,
tests/test_oracle_golden.py, tests/golden/matchformer_e2e.npz).  Only tests/, smoke() and bench.py's cpu_baseline may
import it.
"""
import torch
import torch.nn.functional as F

from .restate import coarse_match_from_conf

EMBED = (128, 192, 256, 512)
HEADS = 8
CROSS = ((False, False, True), (False, False, True), (False, True, True), (False, True, True))
PATCH = ((7, 2), (3, 2), (3, 2), (3, 2))          # (kernel, stride) of the four patch embeddings


def as_params(sd):
    """The reference runs inference under ``torch.no_grad()`` with parameters that still require grad, and ATen's CPU
    ``linear`` picks its kernel by that flag when the input is a transposed view (``Mlp``: fc2 after the depth-wise
    conv): results differ in the last bit from a plain tensor with the same values.  To stay bit-comparable the
    oracle marks its floating-point weights the same way; nothing else depends on it."""
    return {k: (v.detach().clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items()}


def _ln(sd, p, x, eps):
    return F.layer_norm(x, (x.shape[-1],), sd[p + "weight"], sd[p + "bias"], eps)


def attention(sd, p, x, cross):
    """Attention.forward -- match_LA_large.py:64-89: q / kv projections with bias, elu+1 feature map, and for the
    cross blocks keys / values of the OTHER image (the two halves of the batch swapped)."""
    B, N, C = x.shape
    D = C // HEADS
    q = F.linear(x, sd[p + "q.weight"], sd[p + "q.bias"]).reshape(B, N, HEADS, D)
    kv = F.linear(x, sd[p + "kv.weight"], sd[p + "kv.bias"]).reshape(B, -1, 2, HEADS, D).permute(2, 0, 1, 3, 4)
    k, v = kv[0], kv[1]
    if cross:
        k1, k2 = k.split(B // 2)
        v1, v2 = v.split(B // 2)
        k, v = torch.cat([k2, k1], dim=0), torch.cat([v2, v1], dim=0)
    Q = F.elu(q) + 1
    K = F.elu(k) + 1
    n = v.size(1)
    v = v / n
    KV = torch.einsum("nshd,nshv->nhdv", K, v)
    Z = 1 / (torch.einsum("nlhd,nhd->nlh", Q, K.sum(dim=1)) + 1e-6)
    out = torch.einsum("nlhd,nhdv,nlh->nlhv", Q, KV, Z) * n
    return out.contiguous().view(B, -1, C)


def mlp(sd, p, x, H, W):
    """Mlp.forward -- :29-46: fc1 -> depth-wise 3x3 conv -> GELU -> fc2."""
    x = F.linear(x, sd[p + "fc1.weight"], sd[p + "fc1.bias"])
    B, N, C = x.shape
    y = x.transpose(1, 2).contiguous().view(B, C, H, W)
    y = F.conv2d(y, sd[p + "dwconv.dwconv.weight"], sd[p + "dwconv.dwconv.bias"], 1, 1, 1, C)
    x = F.gelu(y.flatten(2).transpose(1, 2))
    return F.linear(x, sd[p + "fc2.weight"], sd[p + "fc2.bias"])


def attention_block(sd, p, x, stage):
    """AttentionBlock.forward -- :149-174 (PatchEmbed :118-147, Positional :108-116, Block :91-106)."""
    k, s = PATCH[stage]
    x = F.conv2d(x, sd[p + "patch_embed.proj.weight"], sd[p + "patch_embed.proj.bias"], s, k // 2)
    C = x.shape[1]
    x = x * torch.sigmoid(F.conv2d(x, sd[p + "patch_embed.pos.pa_conv.weight"], sd[p + "patch_embed.pos.pa_conv.bias"], 1, 1, 1, C))
    B, _, H, W = x.shape
    x = _ln(sd, p + "patch_embed.norm.", x.flatten(2).transpose(1, 2), 1e-5)
    for i in range(3):
        q = f"{p}block.{i}."
        x = x + attention(sd, q + "attn.", _ln(sd, q + "norm1.", x, 1e-6), CROSS[stage][i])
        x = x + mlp(sd, q + "mlp.", _ln(sd, q + "norm.", x, 1e-6), H, W)
    x = _ln(sd, p + "norm.", x, 1e-6)
    return x.reshape(B, H, W, -1).permute(0, 3, 1, 2).contiguous()


def _outconv2(sd, p, x):
    x = F.conv2d(x, sd[p + "0.weight"], None, 1, 1)
    x = F.batch_norm(x, sd[p + "1.running_mean"], sd[p + "1.running_var"], sd[p + "1.weight"], sd[p + "1.bias"], False, 0.0, 1e-5)
    return F.conv2d(F.leaky_relu(x, 0.01), sd[p + "3.weight"], None, 1, 1)


def backbone(sd, x, with_fine=True, p="backbone."):
    """Matchformer_LA_large.forward -- :220-255.  Returns (c3_out [B,256,H/8,W/8], c1_out [B,128,H/2,W/2] or None)."""
    outs = []
    for s in range(4):
        x = attention_block(sd, f"{p}AttentionBlock{s + 1}.", x, s)
        outs.append(x)
    out1, out2, out3, out4 = outs
    c4 = F.conv2d(out4, sd[p + "layer4_outconv.weight"])
    c4_2x = F.interpolate(c4, size=out3.shape[2:], mode="bilinear", align_corners=True)
    c3 = _outconv2(sd, p + "layer3_outconv2.", F.conv2d(out3, sd[p + "layer3_outconv.weight"]) + c4_2x)
    if not with_fine:
        return c3, None
    c3_2x = F.interpolate(c3, size=out2.shape[2:], mode="bilinear", align_corners=True)
    c2 = _outconv2(sd, p + "layer2_outconv2.", F.conv2d(out2, sd[p + "layer2_outconv.weight"]) + c3_2x)
    c2_2x = F.interpolate(c2, size=out1.shape[2:], mode="bilinear", align_corners=True)
    c1 = _outconv2(sd, p + "layer1_outconv2.", F.conv2d(out1, sd[p + "layer1_outconv.weight"]) + c2_2x)
    return c3, c1


def dual_softmax_conf_masked(feat0, feat1, temperature, mask0=None, mask1=None):
    """CoarseMatching.forward, dual_softmax -- coarse_matching.py:99-117 (padding masks: sim.masked_fill_(-1e9))."""
    C = feat0.shape[-1]
    f0, f1 = feat0 / C ** 0.5, feat1 / C ** 0.5
    sim = torch.einsum("nlc,nsc->nls", f0, f1) / temperature
    if mask0 is not None:
        sim.masked_fill_(~(mask0[..., None] * mask1[:, None]).bool(), -1e9)
    return F.softmax(sim, 1) * F.softmax(sim, 2)


def matchformer_forward(sd, cfg, data, with_fine_backbone=True):
    """Matchformer.forward with fine.enable = False -- matchformer.py:21-52."""
    sd = as_params(sd)
    img0, img1 = data["image0"], data["image1"]
    bs = img0.size(0)
    assert img0.shape[2:] == img1.shape[2:], "the cross blocks pair the two halves of ONE batch: equal frame sizes"
    c, _ = backbone(sd, torch.cat([img0, img1], 0), with_fine_backbone)
    c0, c1 = c.split(bs)
    hw0_c, hw1_c = tuple(c0.shape[2:]), tuple(c1.shape[2:])
    f0 = c0.flatten(2).transpose(1, 2)
    f1 = c1.flatten(2).transpose(1, 2)
    mc = cfg["match_coarse"]
    m0 = data["mask0"].flatten(-2) if "mask0" in data else None
    m1 = data["mask1"].flatten(-2) if "mask1" in data else None
    conf = dual_softmax_conf_masked(f0, f1, mc["dsmax_temperature"], m0, m1)
    assert mc["border_rm"] == 0          # mask_border returns at once for b <= 0 (coarse_matching.py:15-16)
    out = coarse_match_from_conf(conf, hw0_c, hw1_c, tuple(img0.shape[2:]), mc["thr"], 0, data.get("scale0"), data.get("scale1"))
    out.update({"feat_c0": f0, "feat_c1": f1, "hw0_c": hw0_c, "hw1_c": hw1_c, "conf_matrix": conf,
                "mkpts0_f": out["mkpts0_c"], "mkpts1_f": out["mkpts1_c"]})
    return out
