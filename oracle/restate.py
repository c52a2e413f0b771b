"""TEST INFRASTRUCTURE ONLY -- CPU restatement (oracle) of the DetectorFreeSfM dense-matching hot path.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this file, and only as the checker / reported baseline.  The product
(``detectorfreesfm_amd``) never imports it and has no CPU fallback.

What it is: a *functional* fp32 torch-CPU restatement of the reference's algorithm for the two
hot-path halves, written against a plain ``state_dict`` (so the same seeded weights feed the real
reference, this oracle and the HIP product):

  coarse   LoFTR.forward, coarse_only          third_party/LoFTR/src/loftr/loftr.py:29-73
  refine   MultiviewMatcher.forward (test)     src/MultiviewMatcher/MultiviewMatcher.py:59-405

Every function cites the reference file:line it follows (paths relative to /root/reference).

How it is pinned: the reference ships NO golden vectors / known-answer tests for this path
(SURVEY.md section 4).  The oracle is therefore pinned against outputs of the reference's own
Python modules executed in the build container (``oracle/ref_import.py`` +
``oracle/make_golden.py`` -> ``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` also
re-checks it live whenever /root/reference is present).

PARITY UNPINNED for one stage: ``roi_align_crop`` restates longcw/RoIAlign.pytorch
(``roi_align`` package; git submodule third_party/RoIAlign.pytorch, .gitmodules:1-3, directory
empty, pinned commit unknown).  Its published algorithm (TensorFlow ``crop_and_resize``:
bilinear, extrapolation_value=0, boxes normalised by (size-1) when transform_fpcoor=False) is
restated from the upstream README/source as recalled in SURVEY.md section 8c; parity is
anchored on the reference's call site src/MultiviewMatcher/matcher_module/fine_preprocess.py:92-106
and on our own known-answer tests (integer-centred boxes reproduce pixels exactly;
out-of-image samples are 0).
"""
import math

import torch
import torch.nn.functional as F

# ----------------------------------------------------------------------------------------------
# K1  linear attention
# ----------------------------------------------------------------------------------------------


def elu1(x):
    """third_party/LoFTR/src/loftr/loftr_module/linear_attention.py:10-11"""
    return F.elu(x) + 1


def linear_attention(q, k, v, q_mask=None, kv_mask=None, eps=1e-6):
    """LinearAttention.forward -- LoFTR linear_attention.py:20-47 and the identical arithmetic in
    src/MultiviewMatcher/matcher_module/linear_attention.py:28-60.
    q [N,L,H,D], k,v [N,S,H,D], masks [N,L]/[N,S] (bool or float) -> [N,L,H,D]."""
    Q = elu1(q)
    K = elu1(k)
    if q_mask is not None:
        Q = Q * q_mask[:, :, None, None]
    if kv_mask is not None:
        K = K * kv_mask[:, :, None, None]
        v = v * kv_mask[:, :, None, None]
    S = v.size(1)
    v = v / S
    KV = torch.einsum("nshd,nshv->nhdv", K, v)
    Z = 1 / (torch.einsum("nlhd,nhd->nlh", Q, K.sum(dim=1)) + eps)
    return (torch.einsum("nlhd,nhdv,nlh->nlhv", Q, KV, Z) * S).contiguous()


# ----------------------------------------------------------------------------------------------
# K2  encoder layer / transformers
# ----------------------------------------------------------------------------------------------


def encoder_layer(sd, p, x, source, nhead, x_mask=None, source_mask=None):
    """LoFTREncoderLayer.forward -- LoFTR transformer.py:35-58 (coarse) and
    src/MultiviewMatcher/matcher_module/transformer.py:66-95 (refine; dropout=0, rezero=None)."""
    bs, C = x.size(0), x.size(2)
    D = C // nhead
    q = F.linear(x, sd[p + "q_proj.weight"]).view(bs, -1, nhead, D)
    k = F.linear(source, sd[p + "k_proj.weight"]).view(bs, -1, nhead, D)
    v = F.linear(source, sd[p + "v_proj.weight"]).view(bs, -1, nhead, D)
    msg = linear_attention(q, k, v, x_mask, source_mask)
    msg = F.linear(msg.view(bs, -1, C), sd[p + "merge.weight"])
    msg = F.layer_norm(msg, (C,), sd[p + "norm1.weight"], sd[p + "norm1.bias"])
    msg = F.linear(torch.cat([x, msg], dim=2), sd[p + "mlp.0.weight"])
    msg = F.linear(F.relu(msg), sd[p + "mlp.2.weight"])
    msg = F.layer_norm(msg, (C,), sd[p + "norm2.weight"], sd[p + "norm2.bias"])
    return x + msg


def coarse_transformer(sd, p, feat0, feat1, layer_names, nhead, mask0=None, mask1=None):
    """LocalFeatureTransformer.forward -- LoFTR transformer.py:80-101.
    NOTE the 'cross' order: feat1 attends to the ALREADY UPDATED feat0 (:96-97)."""
    for i, name in enumerate(layer_names):
        lp = f"{p}layers.{i}."
        if name == "self":
            feat0 = encoder_layer(sd, lp, feat0, feat0, nhead, mask0, mask0)
            feat1 = encoder_layer(sd, lp, feat1, feat1, nhead, mask1, mask1)
        elif name == "cross":
            feat0 = encoder_layer(sd, lp, feat0, feat1, nhead, mask0, mask1)
            feat1 = encoder_layer(sd, lp, feat1, feat0, nhead, mask1, mask0)
        else:
            raise KeyError(name)
    return feat0, feat1


def multiview_transformer(sd, p, ref, qry, layer_names, nhead, query_mask=None):
    """multiview LocalFeatureTransformer.forward -- src/MultiviewMatcher/matcher_module/transformer.py:132-178.
    ref [T,WW,C]; qry [T,Vq,WW,C]; query_mask [T,Vq] -> same shapes.
    NOTE the 'cross' order differs from the coarse matcher: BOTH sides are updated from the
    pre-update tensors (src0, src1 captured at :163)."""
    T, Vq, WW, C = qry.shape
    q = qry.reshape(T, Vq * WW, C)
    qm = None
    if query_mask is not None:
        qm = query_mask[:, :, None].expand(T, Vq, WW).reshape(T, Vq * WW)
    for i, name in enumerate(layer_names):
        lp = f"{p}layers.{i}."
        if name == "self":
            ref, q = (encoder_layer(sd, lp, ref, ref, nhead, None, None),
                      encoder_layer(sd, lp, q, q, nhead, qm, qm))
        elif name == "cross":
            s0, s1 = ref, q
            q, ref = (encoder_layer(sd, lp, q, s0, nhead, qm, None),
                      encoder_layer(sd, lp, ref, s1, nhead, None, qm))
        else:
            raise NotImplementedError(name)
    return ref, q.reshape(T, Vq, WW, C)


# ----------------------------------------------------------------------------------------------
# K7  positional encoding, K6 ResNetFPN_8_2
# ----------------------------------------------------------------------------------------------


def position_encoding_sine(d_model, max_shape=(256, 256), temp_bug_fix=False):
    """PositionEncodingSine.__init__ -- LoFTR utils/position_encoding.py:22-35.
    temp_bug_fix=False reproduces the operator-precedence quirk at :28:
    ``-ln(1e4) / d_model // 2`` == floor(-ln(1e4)/d_model / 2) == -1.0 for d_model=256."""
    pe = torch.zeros((d_model, *max_shape))
    y = torch.ones(max_shape).cumsum(0).float().unsqueeze(0)
    x = torch.ones(max_shape).cumsum(1).float().unsqueeze(0)
    ar = torch.arange(0, d_model // 2, 2).float()
    if temp_bug_fix:
        div = torch.exp(ar * (-math.log(10000.0) / (d_model // 2)))
    else:
        div = torch.exp(ar * (-math.log(10000.0) / d_model // 2))
    div = div[:, None, None]
    pe[0::4] = torch.sin(x * div)
    pe[1::4] = torch.cos(x * div)
    pe[2::4] = torch.sin(y * div)
    pe[3::4] = torch.cos(y * div)
    return pe.unsqueeze(0)


def _bn(sd, p, x, eps=1e-5):
    return F.batch_norm(x, sd[p + "running_mean"], sd[p + "running_var"], sd[p + "weight"],
                        sd[p + "bias"], False, 0.0, eps)


def _basic_block(sd, p, x, stride):
    """BasicBlock.forward -- LoFTR backbone/resnet_fpn.py:31-40."""
    y = F.relu(_bn(sd, p + "bn1.", F.conv2d(x, sd[p + "conv1.weight"], None, stride, 1)))
    y = _bn(sd, p + "bn2.", F.conv2d(y, sd[p + "conv2.weight"], None, 1, 1))
    if stride != 1:
        x = _bn(sd, p + "downsample.1.", F.conv2d(x, sd[p + "downsample.0.weight"], None, stride, 0))
    return F.relu(x + y)


def resnet_fpn_8_2(sd, p, x, with_fine=True):
    """ResNetFPN_8_2.forward -- LoFTR backbone/resnet_fpn.py:100-118.
    Returns (x3_out [N,256,H/8,W/8], x1_out [N,128,H/2,W/2] or None)."""
    x0 = F.relu(_bn(sd, p + "bn1.", F.conv2d(x, sd[p + "conv1.weight"], None, 2, 3)))
    x1 = _basic_block(sd, p + "layer1.1.", _basic_block(sd, p + "layer1.0.", x0, 1), 1)
    x2 = _basic_block(sd, p + "layer2.1.", _basic_block(sd, p + "layer2.0.", x1, 2), 1)
    x3 = _basic_block(sd, p + "layer3.1.", _basic_block(sd, p + "layer3.0.", x2, 2), 1)
    x3_out = F.conv2d(x3, sd[p + "layer3_outconv.weight"])
    if not with_fine:
        return x3_out, None

    def outconv2(q, t):
        t = F.conv2d(t, sd[q + "0.weight"], None, 1, 1)
        t = F.leaky_relu(_bn(sd, q + "1.", t), 0.01)
        return F.conv2d(t, sd[q + "3.weight"], None, 1, 1)
    x3_2x = F.interpolate(x3_out, scale_factor=2.0, mode="bilinear", align_corners=True)
    x2_out = outconv2(p + "layer2_outconv2.", F.conv2d(x2, sd[p + "layer2_outconv.weight"]) + x3_2x)
    x2_2x = F.interpolate(x2_out, scale_factor=2.0, mode="bilinear", align_corners=True)
    x1_out = outconv2(p + "layer1_outconv2.", F.conv2d(x1, sd[p + "layer1_outconv.weight"]) + x2_2x)
    return x3_out, x1_out


# ----------------------------------------------------------------------------------------------
# K3 / K4 / K5  coarse matching
# ----------------------------------------------------------------------------------------------


def dual_softmax_conf(feat0, feat1, temperature, mask0=None, mask1=None):
    """CoarseMatching.forward -- LoFTR utils/coarse_matching.py:103-116 (dual_softmax; padding masks [N,L] / [N,S]:
    ``sim_matrix.masked_fill_(~(mask_c0[..., None] * mask_c1[:, None]).bool(), -INF)`` with INF = 1e9, :108-112)."""
    C = feat0.shape[-1]
    f0, f1 = feat0 / C ** 0.5, feat1 / C ** 0.5
    sim = torch.einsum("nlc,nsc->nls", f0, f1) / temperature
    if mask0 is not None:
        sim.masked_fill_(~(mask0[..., None] * mask1[:, None]).bool(), -1e9)
    return F.softmax(sim, 1) * F.softmax(sim, 2)


def coarse_match_from_conf(conf, hw0_c, hw1_c, hw0_i, thr, border_rm, scale0=None, scale1=None, mask0=None, mask1=None):
    """CoarseMatching.get_coarse_match (eval) -- coarse_matching.py:148-258, mask_border :8-22.
    The high-side border slices ``m[:, -b:0]`` are EMPTY, so only the first ``b`` rows/cols of
    each grid are removed; reproduced here for index parity.  With padding masks (``'mask0' in data``, [N,h,w]) the
    border is ``mask_border_with_padding`` (:25-41) instead: the low side as above, and per pair the last ``b`` rows /
    columns of each frame's VALID extent (h = max over columns of the column sums, w = max over rows of the row sums) and
    everything beyond -- Python slices ``m[b_idx, h0 - bd:]``, so a negative start counts from the end."""
    N = conf.shape[0]
    h0, w0 = hw0_c
    h1, w1 = hw1_c
    mask = (conf > thr).view(N, h0, w0, h1, w1).clone()
    b = border_rm
    if b > 0:
        mask[:, :b] = False
        mask[:, :, :b] = False
        mask[:, :, :, :b] = False
        mask[:, :, :, :, :b] = False
        if mask0 is not None:
            h0s, w0s = mask0.sum(1).max(-1)[0].int(), mask0.sum(-1).max(-1)[0].int()
            h1s, w1s = mask1.sum(1).max(-1)[0].int(), mask1.sum(-1).max(-1)[0].int()
            for n, (a0, b0, a1, b1) in enumerate(zip(h0s, w0s, h1s, w1s)):
                mask[n, a0 - b:] = False
                mask[n, :, b0 - b:] = False
                mask[n, :, :, a1 - b:] = False
                mask[n, :, :, :, b1 - b:] = False
    mask = mask.view(N, h0 * w0, h1 * w1)
    mask = mask * (conf == conf.max(dim=2, keepdim=True)[0]) * (conf == conf.max(dim=1, keepdim=True)[0])
    mask_v, all_j = mask.max(dim=2)
    b_ids, i_ids = torch.where(mask_v)
    j_ids = all_j[b_ids, i_ids]
    mconf = conf[b_ids, i_ids, j_ids]
    scale = hw0_i[0] / hw0_c[0]
    s0 = scale * scale0[b_ids][:, [1, 0]] if scale0 is not None else scale
    s1 = scale * scale1[b_ids][:, [1, 0]] if scale1 is not None else scale
    mk0 = torch.stack([i_ids % w0, i_ids // w0], dim=1) * s0
    mk1 = torch.stack([j_ids % w1, j_ids // w1], dim=1) * s1
    keep = mconf != 0
    return {"b_ids": b_ids, "i_ids": i_ids, "j_ids": j_ids, "m_bids": b_ids[keep],
            "mkpts0_c": mk0[keep], "mkpts1_c": mk1[keep], "mconf": mconf[keep]}


def coarse_matching(feat0, feat1, hw0_c, hw1_c, hw0_i, thr=0.2, border_rm=2, temperature=0.1,
                    scale0=None, scale1=None, return_conf=False, mask0=None, mask1=None):
    """mask0 / mask1: padding masks [N,h0c,w0c] / [N,h1c,w1c] (1 = valid), as ``data['mask0']`` (loftr.py:61-65)."""
    N = feat0.shape[0]
    m0 = None if mask0 is None else mask0.reshape(N, -1)
    m1 = None if mask1 is None else mask1.reshape(N, -1)
    conf = dual_softmax_conf(feat0, feat1, temperature, m0, m1)
    out = coarse_match_from_conf(conf, hw0_c, hw1_c, hw0_i, thr, border_rm, scale0, scale1, mask0, mask1)
    if return_conf:
        out["conf_matrix"] = conf
    return out


def loftr_coarse_forward(sd, cfg, data, with_fine_backbone=True):
    """LoFTR.forward with fine.enable=False -- LoFTR loftr.py:29-73.  ``cfg`` is the lower-cased
    LOFTR config dict.  ``with_fine_backbone=True`` also evaluates the FPN top-down branch the
    reference computes and discards in coarse_only mode (resnet_fpn.py:110-116)."""
    img0, img1 = data["image0"], data["image1"]
    bs = img0.size(0)
    hw0_i, hw1_i = tuple(img0.shape[2:]), tuple(img1.shape[2:])
    p = "backbone."
    if hw0_i == hw1_i:
        c, _ = resnet_fpn_8_2(sd, p, torch.cat([img0, img1], 0), with_fine_backbone)
        c0, c1 = c.split(bs)
    else:
        c0, _ = resnet_fpn_8_2(sd, p, img0, with_fine_backbone)
        c1, _ = resnet_fpn_8_2(sd, p, img1, with_fine_backbone)
    hw0_c, hw1_c = tuple(c0.shape[2:]), tuple(c1.shape[2:])
    pe = position_encoding_sine(cfg["coarse"]["d_model"], temp_bug_fix=cfg["coarse"]["temp_bug_fix"])
    f0 = (c0 + pe[:, :, :hw0_c[0], :hw0_c[1]]).flatten(2).transpose(1, 2)
    f1 = (c1 + pe[:, :, :hw1_c[0], :hw1_c[1]]).flatten(2).transpose(1, 2)
    m0 = m1 = None                      # loftr.py:61-63: padding masks at the coarse resolution, flattened
    if "mask0" in data:
        m0, m1 = data["mask0"].flatten(-2), data["mask1"].flatten(-2)
    f0, f1 = coarse_transformer(sd, "loftr_coarse.", f0, f1, cfg["coarse"]["layer_names"],
                                cfg["coarse"]["nhead"], m0, m1)
    mc = cfg["match_coarse"]
    out = coarse_matching(f0, f1, hw0_c, hw1_c, hw0_i, mc["thr"], mc["border_rm"],
                          mc["dsmax_temperature"], data.get("scale0"), data.get("scale1"),
                          mask0=data.get("mask0"), mask1=data.get("mask1"))
    out.update({"feat_c0": f0, "feat_c1": f1, "hw0_c": hw0_c, "hw1_c": hw1_c,
                "mkpts0_f": out["mkpts0_c"], "mkpts1_f": out["mkpts1_c"]})
    return out


# ----------------------------------------------------------------------------------------------
# K8  RoIAlign (restated third-party; PARITY UNPINNED, see header)
# ----------------------------------------------------------------------------------------------


def roi_align_crop(featuremap, boxes, box_ind, crop_h, crop_w, extrapolation_value=0.0):
    """roi_align.RoIAlign(crop_h, crop_w, transform_fpcoor=False).forward as called from
    src/MultiviewMatcher/matcher_module/fine_preprocess.py:92-106.
    featuremap [N,C,H,W]; boxes [M,4]=(x1,y1,x2,y2) pixels; box_ind [M] -> [M,C,crop_h,crop_w].
    fp32 operation order follows upstream: normalise by (size-1), then TensorFlow
    crop_and_resize: in = y1n*(H-1) + iy*((y2n-y1n)*(H-1)/(crop_h-1)); sample outside
    [0,size-1] -> extrapolation_value; bilinear between floor/ceil neighbours with
    top = tl+(tr-tl)*xl ; bot = bl+(br-bl)*xl ; out = top+(bot-top)*yl."""
    N, C, H, W = featuremap.shape
    M = boxes.shape[0]
    f32 = torch.float32
    boxes = boxes.to(f32)
    x1 = boxes[:, 0] / float(W - 1)
    y1 = boxes[:, 1] / float(H - 1)
    x2 = boxes[:, 2] / float(W - 1)
    y2 = boxes[:, 3] / float(H - 1)
    hs = (y2 - y1) * (H - 1) / (crop_h - 1) if crop_h > 1 else torch.zeros_like(y1)
    ws = (x2 - x1) * (W - 1) / (crop_w - 1) if crop_w > 1 else torch.zeros_like(x1)
    iy = torch.arange(crop_h, dtype=f32)
    ix = torch.arange(crop_w, dtype=f32)
    if crop_h > 1:
        in_y = (y1 * (H - 1))[:, None] + iy[None] * hs[:, None]
    else:
        in_y = (0.5 * (y1 + y2) * (H - 1))[:, None].expand(M, 1)
    if crop_w > 1:
        in_x = (x1 * (W - 1))[:, None] + ix[None] * ws[:, None]
    else:
        in_x = (0.5 * (x1 + x2) * (W - 1))[:, None].expand(M, 1)
    vy = (in_y >= 0) & (in_y <= H - 1)
    vx = (in_x >= 0) & (in_x <= W - 1)
    ty = torch.floor(in_y)
    by = torch.ceil(in_y)
    yl = (in_y - ty)[:, None, :, None]
    lx = torch.floor(in_x)
    rx = torch.ceil(in_x)
    xl = (in_x - lx)[:, None, None, :]
    tyi = ty.clamp(0, H - 1).long()
    byi = by.clamp(0, H - 1).long()
    lxi = lx.clamp(0, W - 1).long()
    rxi = rx.clamp(0, W - 1).long()
    out = torch.empty(M, C, crop_h, crop_w, dtype=featuremap.dtype)
    bi = box_ind.long()
    for n in torch.unique(bi).tolist():
        sel = bi == n
        fm = featuremap[n]                                   # [C,H,W]

        def g(yi, xi):
            return fm[:, yi[sel][:, :, None], xi[sel][:, None, :]].permute(1, 0, 2, 3)
        tl, tr = g(tyi, lxi), g(tyi, rxi)
        bl, br = g(byi, lxi), g(byi, rxi)
        top = tl + (tr - tl) * xl[sel]
        bot = bl + (br - bl) * xl[sel]
        val = top + (bot - top) * yl[sel]
        valid = (vy[sel][:, None, :, None] & vx[sel][:, None, None, :]).expand_as(val)
        out[sel] = torch.where(valid, val, torch.full_like(val, extrapolation_value))
    return out


def extract_local_patches(image, points_xy, crop_size):
    """FinePreprocess._extract_local_patches (scales=None) -- fine_preprocess.py:92-106:
    boxes = [pt - crop//2, pt + crop//2]; image [1,C,H,W]; points [M,2] -> [M,C,crop,crop]."""
    r = crop_size // 2
    boxes = torch.cat([points_xy - r, points_xy + r], dim=-1).to(torch.float32)
    bid = torch.zeros(points_xy.shape[0], dtype=torch.int32)
    return roi_align_crop(image, boxes, bid, crop_size, crop_size, 0.0)


# ----------------------------------------------------------------------------------------------
# K9  S2DNet
# ----------------------------------------------------------------------------------------------

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)
# encoder indices of the conv layers in torchvision vgg16.features[:16] (s2dnet.py:80-92)
_VGG_CONVS = (0, 2, 5, 7, 10, 12, 14)


def s2dnet_forward(sd, p, patches, window_size):
    """S2DNet._forward with num_layers=2, combine=True, substitute_pooling_layers=True,
    zoomin_strategy='post', scales=None, sparse=True -- s2dnet.py:127-193 (AdapLayers :24-52).
    patches [M,3,crop,crop] -> [M,128,window,window]."""
    mean = patches.new_tensor(IMAGENET_MEAN)[:, None, None]
    std = patches.new_tensor(IMAGENET_STD)[:, None, None]
    x = (patches - mean) / std

    def conv(i, t):
        return F.relu(F.conv2d(t, sd[f"{p}encoder.{i}.weight"], sd[f"{p}encoder.{i}.bias"], 1, 1))
    x = conv(2, conv(0, x))
    f0 = x                                                     # relu1_2, full res
    x = F.max_pool2d(x, 3, 2, 1)
    x = conv(7, conv(5, x))
    x = F.max_pool2d(x, 3, 2, 1)
    x = conv(14, conv(12, conv(10, x)))
    f1 = x                                                     # relu3_3, 1/4 res

    def adap(i, t):
        q = f"{p}adaptation_layers.adap_layer_{i}."
        t = F.relu(F.conv2d(t, sd[q + "0.weight"], sd[q + "0.bias"]))
        t = F.conv2d(t, sd[q + "2.weight"], sd[q + "2.bias"], 1, 2)
        return _bn(sd, q + "3.", t)
    fmap = adap(0, f0)
    fmap = fmap + F.interpolate(adap(1, f1), size=fmap.shape[2:], mode="bicubic", align_corners=True)
    crop = fmap.shape[-1]
    r = window_size // 2
    c = crop // 2
    return fmap[..., c - r:c + r + 1, c - r:c + r + 1]


# ----------------------------------------------------------------------------------------------
# K11 / K12  fine matching
# ----------------------------------------------------------------------------------------------


def _grid_normalized(W):
    """kornia.utils.create_meshgrid(W, W, True) flattened: [WW,2], (x,y), linspace(-1,1)."""
    lin = (torch.linspace(0, W - 1, W) / (W - 1) - 0.5) * 2
    gy, gx = torch.meshgrid(lin, lin, indexing="ij")
    return torch.stack([gx, gy], -1).reshape(-1, 2)


def fine_matching(ref, qry, W, left_win, track_mask, movable_mask):
    """FineMatching.forward core (test config: left_point_movement=left_win,
    best_left_strategy='smallest_mean_std', s2d heatmap + argsoftmax) --
    src/MultiviewMatcher/utils/fine_matching.py:36-98, select_left_point :100-119,
    _s2d_heatmap :195-219, argsoftmax :258-285, _obtain_left_normalized_offset :129-179.
    ref [T,WW,C]; qry [T,Vq,WW,C]; track_mask [T,Vq] bool; movable_mask [T] bool.
    Returns left_offset_norm [T,2], coords_normed [T,Vq,2], std [T,Vq], best_index [T]."""
    T, Vq, WW, C = qry.shape
    r = left_win // 2
    c = W // 2
    rw = ref.view(T, W, W, C)[:, c - r:c + r + 1, c - r:c + r + 1].reshape(T, left_win * left_win, C)
    sim = torch.einsum("mlc,mnrc->mlnr", rw, qry)
    heat = torch.softmax((1.0 / C ** 0.5) * sim, dim=-1)        # [T,L,Vq,WW]
    grid = _grid_normalized(W)
    L = left_win * left_win
    h = heat.reshape(T, L * Vq, WW)
    ex = torch.sum(grid[:, 0] * h, -1, keepdim=True)
    ey = torch.sum(grid[:, 1] * h, -1, keepdim=True)
    coords = torch.cat([ex, ey], -1)                            # [T,L*Vq,2]
    var = torch.sum(grid.reshape(1, 1, WW, 2) ** 2 * h.reshape(T, L * Vq, WW, 1), dim=-2) - coords ** 2
    std = torch.sum(torch.sqrt(torch.clamp(var, min=1e-10)), -1)
    coords = coords.view(T, L, Vq, 2)
    std = std.view(T, L, Vq)
    tm = track_mask[:, None, :].expand(T, L, Vq).float()
    score = (tm * std).sum(-1) / tm.sum(-1).clamp(min=1)
    best = torch.min(score, dim=-1)[1]
    best = torch.where(movable_mask, best, torch.full_like(best, L // 2))
    off = torch.stack([best % left_win, best // left_win], -1)
    left_norm = (off / (left_win - 1)) * 2 - 1
    ids = torch.arange(T)
    fine_matching.last_score = score          # [T,L] candidate scores, kept for the parity tests' tie analysis
    return left_norm, coords[ids, best], std[ids, best], best


def multiview_matcher_forward(sd, cfg, data, chunk_track=1000):
    """MultiviewMatcher.forward(data, chunk_track=1000, chunk_backbone_img=True), eval/test mode,
    n_matching_steps=1, enable_multiview_scale_align=False --
    src/MultiviewMatcher/MultiviewMatcher.py:59-405.  ``cfg`` = model.multiview_refinement dict.
    ``data['images']`` is a list of [1,3,h,w].  Returns a dict with
    query_points_refined [1,T,2], reference_points_refined [1,V-1,T,2], std [1,V-1,T]."""
    images = data["images"]
    n_img = len(images)
    mt = cfg["multiview_transform"]
    W, crop = mt["window_size"], mt["crop_size"]
    left = cfg["multiview_matching_test"]["left_point_movement_window_size"]
    fine_res = cfg["backbone"]["resolution"][-1]
    fine_scale = torch.full((1, n_img, 2), float(fine_res))
    scales = fine_scale * data["scales"][:, :, [1, 0]] if "scales" in data else fine_scale   # :68-78
    cur_ref = data["reference_points_coarse"]
    pts = torch.cat([data["query_points"][:, None], cur_ref], dim=1)                           # [1,V,T,2]
    img_idxs = torch.cat([data["query_img_idxs"][:, None], data["reference_img_idxs"]], dim=1)
    B, V, T = img_idxs.shape
    pt_scales = scales.view(-1, 2)[img_idxs.view(-1)].view(B, V, T, 2)                         # :103 (-1 -> last)
    pts = pts / pt_scales

    # view-count grouping :117-133
    max_view_tracks = 16 * chunk_track
    keys, counts = torch.unique(data["track_valid_mask"].sum(-2).max(0)[0], sorted=True, return_counts=True)
    keys, counts = keys.flip(0), counts.flip(0).clone()
    chunk_views, num_tracks = [], []
    i = 0
    while i < keys.shape[0]:
        vv = int(keys[i]) + 1
        chunk_views.append(vv)
        if vv * int(counts[i]) <= max_view_tracks:
            num_tracks.append(int(counts[i]))
            i += 1
        else:
            counts[i] -= max_view_tracks // vv
            num_tracks.append(max_view_tracks // vv)

    # per-image crop + backbone :188-279
    flat_idx = img_idxs.reshape(-1)
    flat_pts = pts.reshape(-1, 2)
    order = torch.full((B * V * T,), -1, dtype=torch.long)
    feats, n_done = [], 0
    for ii in range(n_img):
        m = flat_idx == ii
        if int(m.sum()) == 0:
            continue
        patches = extract_local_patches(images[ii], flat_pts[m], crop)
        f = s2dnet_forward(sd, "backbone.", patches, W)                                         # [M,128,W,W]
        f = f.flatten(2).transpose(1, 2)                                                        # m (h w) c
        order[m] = torch.arange(n_done, n_done + f.shape[0])
        feats.append(f)
        n_done += f.shape[0]
    feats = torch.cat(feats, 0)[order]                                                          # -1 -> last row
    feats = feats.view(B, V, T, W * W, -1).permute(0, 2, 1, 3, 4)[0]                            # [T,V,WW,C]
    f_ref, f_qry = feats[:, 0], feats[:, 1:]

    tvm = data["track_valid_mask"].transpose(1, 2)[0]                                           # [T,V-1]
    movable = data["query_movable_mask"][0] if "query_movable_mask" in data else torch.ones(T, dtype=torch.bool)
    q_ref_out, r_ref_out, std_out, score_out, best_out = [], [], [], [], []
    i = 0
    layer_names = list(mt["layer_names"]) * mt["layer_iter_n"]
    for cv, nt in zip(chunk_views, num_tracks):
        sl = slice(i, i + nt)
        fr, fq = f_ref[sl], f_qry[sl, :cv - 1]
        tm = tvm[sl, :cv - 1]
        if mt["enable"]:
            fr, fq = multiview_transformer(sd, "fine_transformer.", fr, fq, layer_names, mt["nhead"], tm)
        left_norm, coords, std, best = fine_matching(fr, fq, W, left, tm, movable[sl])
        score_out.append(fine_matching.last_score)
        best_out.append(best)
        s_q = pt_scales[0, 0, sl]                                                               # [nt,2]
        s_r = pt_scales[0, 1:cv, sl].transpose(0, 1)                                            # [nt,cv-1,2]
        q_ref = data["query_points"][0, sl] + left_norm * (left // 2) * s_q                     # fine_matching.py:221-232
        r_ref = cur_ref[0, :cv - 1, sl].transpose(0, 1) + coords * (W // 2) * s_r               # :234-252
        q_ref_out.append(q_ref)
        r_ref_out.append(F.pad(r_ref.transpose(0, 1), (0, 0, 0, 0, 0, V - cv)))
        std_out.append(F.pad(std.transpose(0, 1), (0, 0, 0, V - cv)))
        i += nt
    return {"query_points_refined": torch.cat(q_ref_out, 0)[None],
            "reference_points_refined": torch.cat(r_ref_out, 1)[None],
            "std": torch.cat(std_out, 1)[None],
            "cand_score": torch.cat(score_out, 0), "best_index": torch.cat(best_out, 0),
            "features_ref": f_ref, "features_qry": f_qry}
