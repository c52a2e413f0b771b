"""MI355X MatchFormer-LA coarse matcher behind the reference's ``NEUSFM_coarse_matcher`` plugin surface
(SURVEY.md 8(f) rank 3).

``HipMatchformer`` is a drop-in for the reference's ``Matchformer`` module with the 'largela' backbone in coarse_only mode
(third_party/MatchFormer/model/matchformer.py:10-64, model/backbone/match_LA_large.py:15-255, config
third_party/MatchFormer/config/matchformer_coarse_only.py): same constructor argument (the lower-cased 'matchformer'
config dict), same 229-tensor ``state_dict`` layout (``matcher.`` prefix stripped, ``strict=True``), same in-place
``forward(data)`` contract (reads image0/1, optional mask0/1 and scale0/1; writes b_ids, i_ids, j_ids, m_bids, mconf,
mkpts0_c/f, mkpts1_c/f, hw*_c ...).

It reuses the LoFTR path's kernels unchanged -- the fp16x2-split MFMA GEMM / convolution kernels (patch embeddings, q, kv,
fc1, fc2 with bias and fused residual; the FPN convs with folded BatchNorm), LayerNorm, linear attention (K1) and the
fused correlation / dual-softmax / mutual-NN stage (K3-K5, here with padding masks) -- and adds what is new in this
architecture: the depth-wise 3x3 convolution with its consumers fused (sigmoid gate of ``Positional``, erf-GELU of ``Mlp``),
K1 at head sizes 24 and 64, LeakyReLU in the conv epilogue and the bilinear 2x of the FPN (csrc/matchformer_ops.hip).
NHWC end to end, so a stage's token matrix [B, L, C] *is* its feature map [B, h, w, C]; both images of every pair share
one batch (the cross blocks pair its two halves, match_LA_large.py:73-78).  Output-identical work the reference wastes
is skipped: FPN levels 2 and 1 (``c1_out`` feeds only the fine matcher, disabled by the shipped config) and the dense
``conf_matrix``.
"""
import torch

from . import ops
from .params import MF_EMBED, MF_PATCH, ParamModule, matchformer_param_spec

MF_HEADS = 8
MF_CROSS = ((False, False, True), (False, False, True), (False, True, True), (False, True, True))   # match_LA_large.py:177-178


def matchformer_coarse_only_config(match_thr: float = 0.4) -> dict:
    """``lower_config(get_cfg_defaults())['matchformer']`` merged with config/matchformer_coarse_only.py
    (src/coarse_match/coarse_match_worker.py:61-71 then overrides ``match_coarse.thr``)."""
    return {
        "backbone_type": "largela", "scens": "outdoor", "resolution": (8, 2), "fine_window_size": 5,
        "fine_concat_coarse_feat": True,
        "coarse": {"d_model": 256, "d_ffn": 256},
        "match_coarse": {"thr": match_thr, "border_rm": 0, "match_type": "dual_softmax", "dsmax_temperature": 0.1,
                         "skh_iters": 3, "skh_init_bin_score": 1.0, "skh_prefilter": False,
                         "train_coarse_percent": 0.2, "train_pad_num_gt_min": 200, "sparse_spvs": True},
        "fine": {"d_model": 128, "d_ffn": 128, "enable": False},
    }


class HipMatchformer(ParamModule):
    def __init__(self, config: dict):
        super().__init__()
        if config["backbone_type"] != "largela":
            raise NotImplementedError("HipMatchformer implements the 'largela' backbone (the one the reference's configs select)")
        if config["match_coarse"]["match_type"] != "dual_softmax" or config["match_coarse"]["border_rm"] != 0:
            raise NotImplementedError("dual_softmax matching with border_rm = 0 (MatchFormer's defaults)")
        if config["fine"]["enable"]:
            raise NotImplementedError("HipMatchformer implements the coarse_only configuration (FINE.ENABLE = False)")
        self.config = config
        self.register_spec(matchformer_param_spec())
        self._packed = None

    # -- checkpoint compatibility (matchformer.py:60-64) ---------------------------------------------------
    def load_state_dict(self, state_dict, *args, **kwargs):
        sd = {(k.replace("matcher.", "", 1) if k.startswith("matcher.") else k): v for k, v in state_dict.items()}
        self._packed = None
        return super().load_state_dict(sd, *args, **kwargs)

    def _apply(self, fn, *a, **kw):
        self._packed = None
        return super()._apply(fn, *a, **kw)

    # -- weight packing ------------------------------------------------------------------------------------
    def _pack(self):
        g = self.p
        P = {"stages": []}
        cin = 1
        for st, C in enumerate(MF_EMBED):
            p = f"backbone.AttentionBlock{st + 1}."
            S = {"proj": ops.PackedDense(g(p + "patch_embed.proj.weight"), g(p + "patch_embed.proj.bias"),
                                         cin_pad=None if st == 0 else cin),          # stage 1 reads the fp32 frame
                 "pos": (g(p + "patch_embed.pos.pa_conv.weight").reshape(C, 9).t().contiguous(), g(p + "patch_embed.pos.pa_conv.bias").contiguous()),
                 "pe_norm": (g(p + "patch_embed.norm.weight").contiguous(), g(p + "patch_embed.norm.bias").contiguous()),
                 "norm": (g(p + "norm.weight").contiguous(), g(p + "norm.bias").contiguous()), "blocks": []}
            for i in range(3):
                q = f"{p}block.{i}."
                S["blocks"].append({
                    "norm1": (g(q + "norm1.weight").contiguous(), g(q + "norm1.bias").contiguous()),
                    "norm": (g(q + "norm.weight").contiguous(), g(q + "norm.bias").contiguous()),
                    "q": ops.PackedDense(g(q + "attn.q.weight"), g(q + "attn.q.bias")),
                    "kv": ops.PackedDense(g(q + "attn.kv.weight"), g(q + "attn.kv.bias")),
                    "fc1": ops.PackedDense(g(q + "mlp.fc1.weight"), g(q + "mlp.fc1.bias")),
                    "dw": (g(q + "mlp.dwconv.dwconv.weight").reshape(4 * C, 9).t().contiguous(), g(q + "mlp.dwconv.dwconv.bias").contiguous()),
                    "fc2": ops.PackedDense(g(q + "mlp.fc2.weight"), g(q + "mlp.fc2.bias")),
                })
            P["stages"].append(S)
            cin = C
        b = "backbone."
        P["l4out"] = ops.PackedDense(g(b + "layer4_outconv.weight"), cin_pad=MF_EMBED[3])
        P["l3out"] = ops.PackedDense(g(b + "layer3_outconv.weight"), cin_pad=MF_EMBED[2])
        q = b + "layer3_outconv2."
        s = g(q + "1.weight") / torch.sqrt(g(q + "1.running_var") + 1e-5)                # conv -> eval BN folded
        P["l3o2_0"] = ops.PackedDense((g(q + "0.weight") * s[:, None, None, None]).contiguous(),
                                      (g(q + "1.bias") - g(q + "1.running_mean") * s).contiguous(), cin_pad=MF_EMBED[3], tap_padded=True)
        P["l3o2_3"] = ops.PackedDense(g(q + "3.weight"), cin_pad=MF_EMBED[3], tap_padded=True)
        self._packed = P
        return P

    # -- one AttentionBlock (match_LA_large.py:149-174) ------------------------------------------------------
    def _stage(self, x, S, st, bs):
        """x: fp32 [B,H,W,1] (stage 1) or SplitAct [B,h,w,Cprev]; returns the stage output as SplitAct [B,h',w',C]."""
        C = MF_EMBED[st]
        D = C // MF_HEADS
        k = MF_PATCH[st]
        y = ops.conv2d_nhwc(x, S["proj"], 2, k // 2)                                  # patch embedding (+ bias)
        B, h, w, _ = y.shape
        L = h * w
        z = ops.dwconv3x3(y, S["pos"][0], S["pos"][1], mode=1)                        # x * sigmoid(pa_conv(x))
        xr = ops.layernorm(z.view(B * L, C), S["pe_norm"][0], S["pe_norm"][1], 1e-5)  # residual stream, fp32 [B*L, C]
        dev = xr.device
        for blk, cross in zip(S["blocks"], MF_CROSS[st]):
            a = ops.SplitAct.empty_rows((B * L,), C, dev)
            ops.layernorm(xr, blk["norm1"][0], blk["norm1"][1], 1e-6, out_split=a, want_f32=False)
            q = ops.linear(a, blk["q"]).view(B, L, MF_HEADS, D)
            kv = ops.linear(a, blk["kv"]).view(B, L, 2 * C)
            kk, vv = kv[..., :C].unflatten(-1, (MF_HEADS, D)), kv[..., C:].unflatten(-1, (MF_HEADS, D))
            msg = torch.empty((B, L, MF_HEADS, D), dtype=torch.float32, device=dev)
            if cross:        # keys / values of the other image of the pair: the two halves of the batch swapped (:73-78)
                ops.linear_attention(q[:bs], kk[bs:], vv[bs:], out=msg[:bs])
                ops.linear_attention(q[bs:], kk[:bs], vv[:bs], out=msg[bs:])
            else:
                ops.linear_attention(q, kk, vv, out=msg)
            x1 = torch.empty_like(xr)
            ops.split_rows(msg.view(B * L, C), xr, out=x1)                            # x + attn(norm1(x))
            m = ops.SplitAct.empty_rows((B * L,), C, dev)
            ops.layernorm(x1, blk["norm"][0], blk["norm"][1], 1e-6, out_split=m, want_f32=False)
            hdn = ops.linear(m, blk["fc1"]).view(B, h, w, 4 * C)
            act = ops.dwconv3x3(hdn, blk["dw"][0], blk["dw"][1], mode=2, out_split=True)     # GELU(dwconv(.))
            xr = ops.linear(ops.SplitAct(act.hi.view(B * L, 4 * C), act.lo.view(B * L, 4 * C), 4 * C), blk["fc2"], residual=x1)
        out = ops.SplitAct.empty(B, h, w, C, dev)
        ops.layernorm(xr, S["norm"][0], S["norm"][1], 1e-6,
                      out_split=ops.SplitAct(out.hi.view(B * L, C), out.lo.view(B * L, C), C), want_f32=False)
        return out

    @torch.no_grad()
    def coarse_features(self, images, bs):
        """images [2*bs,1,H,W] (image0 batch then image1 batch) -> coarse features as SplitAct [2*bs, h/8, w/8, 256]."""
        P = self._packed or self._pack()
        x = images.permute(0, 2, 3, 1)                       # C = 1: NCHW memory is already NHWC
        outs = []
        for st in range(4):
            x = self._stage(x, P["stages"][st], st, bs)
            outs.append(x)
        out3, out4 = outs[2], outs[3]
        c4 = ops.conv2d_nhwc(out4, P["l4out"], 1, 0)                                   # layer4_outconv
        up = ops.bilinear_up(c4, out3.hi.shape[1], out3.hi.shape[2])                    # F.interpolate(..., align_corners=True)
        c3 = ops.conv2d_nhwc(out3, P["l3out"], 1, 0, residual=up, out_split=True)       # layer3_outconv(out3) + c4_out_2x
        t = ops.conv2d_nhwc(c3, P["l3o2_0"], 1, 1, relu=2, out_split=True)              # conv3x3 + BN + LeakyReLU
        return ops.conv2d_nhwc(t, P["l3o2_3"], 1, 1, out_split=True)                    # conv3x3 -> c3_out [B,h,w,256]

    @torch.no_grad()
    @ops.first_call_range_sweep
    def forward(self, data: dict):
        """Updates ``data`` in place like Matchformer.forward (matchformer.py:21-52, fine.enable=False)."""
        img0, img1 = data["image0"], data["image1"]
        if img0.shape[2:] != img1.shape[2:]:
            # the reference would run the backbone per image, whose cross blocks then pair the two halves of ONE image's
            # batch; its data pipeline pads both frames of a pair to one size instead (coarse_match.py:85-87, pad_to = -1)
            raise NotImplementedError("MatchFormer matches two frames of one size (the dataset pads them, masks mark the padding)")
        bs = img0.size(0)
        data.update({"bs": bs, "hw0_i": img0.shape[2:], "hw1_i": img1.shape[2:]})
        feat = self.coarse_features(torch.cat([img0, img1], 0), bs)
        B, h, w, C = feat.hi.shape
        hw = (h, w)
        data.update({"hw0_c": torch.Size(hw), "hw1_c": torch.Size(hw),
                     "hw0_f": torch.Size((img0.shape[2] // 2, img0.shape[3] // 2)),
                     "hw1_f": torch.Size((img1.shape[2] // 2, img1.shape[3] // 2))})
        f = ops.SplitAct(feat.hi.view(B, h * w, C), feat.lo.view(B, h * w, C), C)
        mc = self.config["match_coarse"]
        m = ops.coarse_match(f[:bs], f[bs:], hw, hw, mc["thr"], mc["border_rm"], mc["dsmax_temperature"],
                             data.get("scale0"), data.get("scale1"), data["hw0_i"][0] / h,
                             mask0=data.get("mask0"), mask1=data.get("mask1"))
        data.update({"b_ids": m["b_ids"], "i_ids": m["i_ids"], "j_ids": m["j_ids"], "gt_mask": m["mconf"] == 0,
                     "m_bids": m["b_ids"], "mkpts0_c": m["mkpts0_c"], "mkpts1_c": m["mkpts1_c"], "mconf": m["mconf"],
                     "mkpts0_f": m["mkpts0_c"], "mkpts1_f": m["mkpts1_c"]})
        return None
