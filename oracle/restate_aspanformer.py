"""TEST INFRASTRUCTURE ONLY -- functional fp32 torch-CPU restatement of the ASpanFormer coarse matcher (SURVEY.md 8(f)
rank 4) on a plain ``state_dict``: third_party/aspantransformer/src/ASpanFormer/aspanformer.py:31-111 with
``fine.enable = False`` and ``online_resize = True`` (how src/coarse_match/coarse_match_worker.py:45-60 builds it), for
any frame size: the online resize of sides that are not multiples of 32 (aspanformer.py:119-139) is restated from
torchvision 0.9.1's source (``resize_df`` below) because torchvision is not installed here -- PARITY UNPINNED for that one
step; everything else is pinned.

Each function cites the reference lines it follows (paths relative to third_party/aspantransformer/src/ASpanFormer/).
PINNED: tests/test_oracle_golden.py compares it bit for bit with the real module (imported unchanged through
oracle/ref_import.py) via tests/golden/aspanformer_e2e.npz and live when /root/reference exists.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline may import this file.
"""
import math

import torch
import torch.nn.functional as F

from . import restate

INF = 1e9


def position_encoding(d_model, scaling, max_shape=(256, 256)):
    """PositionEncodingSine.forward with an online ``scaling`` -- utils/position_encoding.py:44-60."""
    pe = torch.zeros((d_model, *max_shape))
    y_position = torch.ones(max_shape).cumsum(0).float().unsqueeze(0) * scaling[0]
    x_position = torch.ones(max_shape).cumsum(1).float().unsqueeze(0) * scaling[1]
    div_term = torch.exp(torch.arange(0, d_model // 2, 2).float() * (-math.log(10000.0) / (d_model // 2)))
    div_term = div_term[:, None, None]
    pe[0::4] = torch.sin(x_position * div_term)
    pe[1::4] = torch.cos(x_position * div_term)
    pe[2::4] = torch.sin(y_position * div_term)
    pe[3::4] = torch.cos(y_position * div_term)
    return pe.unsqueeze(0)


def resize_df(image, df=32):
    """ASpanFormer.resize_df (aspanformer.py:131-139).  ``transforms.Resize([h, w]).forward(tensor)`` of the pinned
    torchvision 0.9.1 (environment.yaml:9) is functional_tensor.resize: ``interpolate(img, size=[h, w], mode='bilinear',
    align_corners=False)`` (the antialias option only exists from 0.10).  torchvision is not installed here, so this line is a
    restatement of that published source, NOT checked against torchvision itself."""
    h, w = image.shape[2], image.shape[3]
    h_new, w_new = h // df * df, w // df * df
    if h != h_new or w != w_new:
        return F.interpolate(image, size=[h_new, w_new], mode="bilinear", align_corners=False)
    return image


def layernorm2d(sd, p, x):
    """aspan_module/attention.py:7-19: channel-wise, unbiased std, eps added to the std."""
    mean, std = x.mean(dim=1, keepdim=True), x.std(dim=1, keepdim=True)
    return sd[p + "affine"][None, :, None, None] * (x - mean) / (std + 1e-6) + sd[p + "bias"][None, :, None, None]


def full_attention(q, k, v, nhead, mask0=None, mask1=None, temp=1):
    """FullAttention.forward -- aspan_module/attention.py:141-165.  q, k, v [N, D, L]."""
    bs, d_model = q.shape[0], q.shape[1]
    q, k, v = (t.view(bs, nhead, d_model // nhead, -1) for t in (q, k, v))
    QK = torch.einsum("nhdl,nhds->nhls", q, k)
    if mask0 is not None:
        QK.masked_fill_(~(mask0[:, None, :, None] * mask1[:, None, None]).bool(), float(-1e8))
    softmax_temp = temp / q.size(2) ** .5
    A = torch.softmax(softmax_temp * QK, dim=-1)
    return torch.einsum("nhls,nhds->nhdl", A, v).contiguous().view(bs, d_model, -1)


def _conv1d(sd, key, x):
    return F.conv1d(x, sd[key])


def message_layer_ini(sd, p, f0, f1, pos1, nhead, mask0, mask1):
    """messageLayer_ini.update -- aspan_module/transformer.py:43-65."""
    bs, d_model, h, w = f0.shape
    f0_flat, f1_flat = f0.view(bs, d_model, -1), f1.view(bs, d_model, -1)
    f1_v = torch.cat([f1_flat, pos1.view(bs, pos1.shape[1], -1)], dim=1)
    q, k = _conv1d(sd, p + "q_proj.weight", f0_flat), _conv1d(sd, p + "k_proj.weight", f1_flat)
    v = _conv1d(sd, p + "v_proj.weight", f1_v)
    msg = _conv1d(sd, p + "merge_head.weight", full_attention(q, k, v, nhead, mask0, mask1)).view(bs, -1, h, w)
    x = torch.cat([f0, layernorm2d(sd, p + "norm1.", msg)], dim=1)
    x = F.conv2d(F.relu(F.conv2d(x, sd[p + "merge_f.0.weight"])), sd[p + "merge_f.2.weight"])
    return f0 + layernorm2d(sd, p + "norm2.", x)


def flow_initializer(sd, p, feat0, feat1, pos0, pos1, cfg, mask0, mask1, ds0, ds1):
    """flow_initializer.forward -- aspan_module/transformer.py:151-187."""
    bs, dim = feat0.size(0), cfg["d_model"]
    h0, w0, h1, w1 = feat0.shape[2], feat0.shape[3], feat1.shape[2], feat1.shape[3]
    sub0, sub1 = F.avg_pool2d(feat0, ds0, stride=ds0), F.avg_pool2d(feat1, ds1, stride=ds1)
    spos0, spos1 = F.avg_pool2d(pos0, ds0, stride=ds0), F.avg_pool2d(pos1, ds1, stride=ds1)
    if mask0 is not None:
        mask0 = -F.max_pool2d(-mask0.view(bs, 1, h0, w0), ds0, stride=ds0).view(bs, -1)
        mask1 = -F.max_pool2d(-mask1.view(bs, 1, h1, w1), ds1, stride=ds1).view(bs, -1)
    for i in range(cfg["ini_layer_num"]):
        lp = f"{p}layers_coarse.{i}."
        sub0, sub1 = (message_layer_ini(sd, lp, sub0, sub1, spos1, cfg["nhead"], mask0, mask1),
                      message_layer_ini(sd, lp, sub1, sub0, spos0, cfg["nhead"], mask1, mask0))
    dec0 = F.conv2d(sub0, sd[p + "decoupler.weight"], sd[p + "decoupler.bias"])
    dec1 = F.conv2d(sub1, sd[p + "decoupler.weight"], sd[p + "decoupler.bias"])
    up = lambda t, ds: F.interpolate(t, scale_factor=ds, mode="bilinear")          # F.upsample(..., mode='bilinear')
    upd0, flow0 = up(dec0[:, :dim], ds0), up(dec0[:, dim:], ds0)
    upd1, flow1 = up(dec1[:, :dim], ds1), up(dec1[:, dim:], ds1)
    feat0 = feat0 + F.conv2d(torch.cat([feat0, upd0], dim=1), sd[p + "up_merge.weight"], sd[p + "up_merge.bias"])
    feat1 = feat1 + F.conv2d(torch.cat([feat1, upd1], dim=1), sd[p + "up_merge.weight"], sd[p + "up_merge.bias"])
    return feat0, feat1, flow0, flow1


def decode_flow(sd, p, flow_feature, kshape):
    """messageLayer_gla.decode_flow -- aspan_module/transformer.py:125-133."""
    bs, _, h, w = flow_feature.shape
    scale_factor = torch.tensor([kshape[1], kshape[0]])[None, None, None]
    x = F.conv1d(F.relu(F.conv1d(flow_feature.view(bs, -1, h * w), sd[p + "flow_decoder.0.weight"])), sd[p + "flow_decoder.2.weight"])
    flow = x.permute(0, 2, 1).view(bs, h, w, 4)
    return torch.cat([torch.sigmoid(flow[:, :, :, :2]) * scale_factor, flow[:, :, :, 2:]], dim=-1)


def partition_token(sd, p, q, k, v, offset, span_scale, maskv, nhead, nsample):
    """HierachicalAttention.partition_token -- aspan_module/attention.py:92-117."""
    bs, d_model, h, w = q.shape
    hk, wk = k.shape[2], k.shape[3]
    offset = offset.view(bs, -1, 2)
    span_scale = span_scale.view(bs, -1, 1, 2)
    offset_sample = sd[p + "sample_offset"][None, None] * span_scale
    sample_pixel = offset[:, :, None] + offset_sample
    sample_norm = sample_pixel / torch.tensor([wk / 2, hk / 2])[None, None, None] - 1
    n0 = nsample[0]
    q = q.view(bs, -1, h // n0, n0, w // n0, n0).permute(0, 1, 2, 4, 3, 5).contiguous().view(bs, nhead, d_model // nhead, -1, n0 ** 2)
    k = F.grid_sample(k, grid=sample_norm, align_corners=False).view(bs, nhead, d_model // nhead, -1, nsample[1] ** 2)
    v = F.grid_sample(v, grid=sample_norm, align_corners=False).view(bs, nhead, d_model // nhead, -1, nsample[1] ** 2)
    mask_sample = None
    if maskv is not None:
        mask_sample = F.grid_sample(maskv.view(bs, -1, h, w).float(), grid=sample_norm, mode="nearest", align_corners=False) == 1
    return q, k, v, mask_sample


def group_attention(query, key, value, temp, d_model, mask_sample=None):
    """HierachicalAttention.group_attention -- aspan_module/attention.py:120-133."""
    bs = query.shape[0]
    QK = torch.einsum("bhdgn,bhdgm->bhgnm", query, key)
    if mask_sample is not None:
        num_head, number_n = QK.shape[1], QK.shape[3]
        QK.masked_fill_(~(mask_sample[:, :, :, None]).expand(-1, num_head, -1, number_n, -1).bool(), float(-1e8))
    softmax_temp = temp / query.size(2) ** .5
    A = torch.softmax(softmax_temp * QK, dim=-1)
    return torch.einsum("bhgnm,bhdgm->bhdgn", A, value).contiguous().view(bs, d_model, -1)


def hierarchical_attention(sd, p, query, key, value, flow, size_q, size_kv, cfg, mask0, mask1, ds0, ds1):
    """HierachicalAttention.forward -- aspan_module/attention.py:42-90."""
    nsample, nhead, d_model, nlevel = cfg["nsample"], cfg["nhead"], cfg["d_model"], 3
    variance, offset = flow[:, :, :, 2:], flow[:, :, :, :2]
    bs = query.shape[0]
    h0, w0, h1, w1 = size_q[0], size_q[1], size_kv[0], size_kv[1]
    variance = torch.exp(0.5 * variance) * cfg["radius_scale"]
    span_scale = torch.clamp((variance * 2 / nsample[1]), min=1)
    sub0, sub1 = [ds0, 2, 1], [ds1, 2, 1]
    q_list = [F.avg_pool2d(query.view(bs, -1, h0, w0), kernel_size=s, stride=s) for s in sub0]
    k_list = [F.avg_pool2d(key.view(bs, -1, h1, w1), kernel_size=s, stride=s) for s in sub1]
    v_list = [F.avg_pool2d(value.view(bs, -1, h1, w1), kernel_size=s, stride=s) for s in sub1]
    offset_list = [F.avg_pool2d(offset.permute(0, 3, 1, 2), kernel_size=s * nsample[0], stride=s * nsample[0]).permute(0, 2, 3, 1) / s
                   for s in sub0[1:]]
    span_list = [F.avg_pool2d(span_scale.permute(0, 3, 1, 2), kernel_size=s * nsample[0], stride=s * nsample[0]).permute(0, 2, 3, 1)
                 for s in sub0[1:]]
    if mask0 is not None:
        mask0, mask1 = mask0.view(bs, 1, h0, w0), mask1.view(bs, 1, h1, w1)
        mask0_list = [-F.max_pool2d(-mask0, kernel_size=s, stride=s) for s in sub0]
        mask1_list = [-F.max_pool2d(-mask1, kernel_size=s, stride=s) for s in sub1]
    else:
        mask0_list = mask1_list = [None, None, None]
    m0f = mask0_list[0].view(bs, -1) if mask0 is not None else None
    m1f = mask1_list[0].view(bs, -1) if mask1 is not None else None
    messages = [full_attention(q_list[0].flatten(2), k_list[0].flatten(2), v_list[0].flatten(2), nhead, m0f, m1f,
                               sd[p + "temp"]).view(bs, d_model, h0 // ds0[0], w0 // ds0[1])]
    for index in range(1, nlevel):
        q, k, v = q_list[index], k_list[index], v_list[index]
        q, k, v, mask_sample = partition_token(sd, p, q, k, v, offset_list[index - 1], span_list[index - 1],
                                               mask0_list[index], nhead, nsample)
        messages.append(group_attention(q, k, v, 1, d_model, mask_sample).view(bs, d_model, h0 // sub0[index], w0 // sub0[index]))
    all_message = torch.cat([F.interpolate(messages[i], scale_factor=sub0[i], mode="nearest") for i in range(nlevel)],
                            dim=1).view(bs, -1, h0 * w0)
    x = F.conv1d(F.relu(F.conv1d(all_message, sd[p + "merge_head.0.weight"])), sd[p + "merge_head.2.weight"])
    return x.view(bs, -1, h0, w0)


def message_layer_gla(sd, p, x0, x1, ff0, ff1, pos0, pos1, cfg, update_flow, mask0, mask1, ds0, ds1):
    """messageLayer_gla.forward / update -- aspan_module/transformer.py:96-123."""
    d_model = cfg["d_model"]

    def update(xa, xb, flow, ffa, posb, ma, mb, dsa, dsb):
        bs = xa.shape[0]
        q = _conv1d(sd, p + "q_proj.weight", xa.view(bs, d_model, -1))
        k = _conv1d(sd, p + "k_proj.weight", xb.view(bs, d_model, -1))
        xb_pos = torch.cat([xb, posb], dim=1)
        v = _conv1d(sd, p + "v_proj.weight", xb_pos.view(bs, xb_pos.shape[1], -1))
        msg = hierarchical_attention(sd, p + "attention.", q, k, v, flow, xa.shape[2:], xb.shape[2:], cfg, ma, mb, dsa, dsb)
        feat = torch.cat([xa, ffa], dim=1) if update_flow else xa
        y = torch.cat([feat, layernorm2d(sd, p + "norm1.", msg)], dim=1)
        y = F.conv2d(F.relu(F.conv2d(y, sd[p + "merge_f.0.weight"])), sd[p + "merge_f.2.weight"], padding=1)
        feat = feat + layernorm2d(sd, p + "norm2.", y)
        return feat[:, :d_model], feat[:, d_model:]
    flow0, flow1 = decode_flow(sd, p, ff0, ff1.shape[2:]), decode_flow(sd, p, ff1, ff0.shape[2:])
    x0n, ff0n = update(x0, x1, flow0, ff0, pos1, mask0, mask1, ds0, ds1)
    x1n, ff1n = update(x1, x0, flow1, ff1, pos0, mask1, mask0, ds1, ds0)
    return x0n, x1n, ff0n, ff1n, flow0, flow1


def coarse_transformer(sd, p, feat0, feat1, pos0, pos1, cfg, mask0=None, mask1=None, ds0=(4, 4), ds1=(4, 4)):
    """LocalFeatureTransformer_Flow.forward -- aspan_module/transformer.py:214-243."""
    bs, d_model = feat0.size(0), cfg["d_model"]
    ds0, ds1 = list(ds0), list(ds1)
    pos0, pos1 = F.conv2d(pos0, sd[p + "pos_transform.weight"]), F.conv2d(pos1, sd[p + "pos_transform.weight"])
    pos0, pos1 = pos0.expand(bs, -1, -1, -1), pos1.expand(bs, -1, -1, -1)
    if mask0 is not None:
        mask0, mask1 = mask0[:, None].float(), mask1[:, None].float()
    feat0, feat1, ff0, ff1 = flow_initializer(sd, p + "ini_layer.", feat0, feat1, pos0, pos1, cfg, mask0, mask1, ds0, ds1)
    flows = [[], []]
    for i in range(cfg["layer_num"]):
        feat0, feat1, ff0, ff1, fl0, fl1 = message_layer_gla(sd, f"{p}layers.{i}.", feat0, feat1, ff0, ff1, pos0, pos1, cfg,
                                                             i < cfg["layer_num"] - 1, mask0, mask1, ds0, ds1)
        flows[0].append(fl0)
        flows[1].append(fl1)
    flows = [torch.stack(flows[0], dim=0), torch.stack(flows[1], dim=0)]
    feat0 = feat0.permute(0, 2, 3, 1).reshape(bs, -1, d_model)
    feat1 = feat1.permute(0, 2, 3, 1).reshape(bs, -1, d_model)
    return feat0, feat1, flows


def offset_matches(flow, conf_mask, hw_c, hw_i0, side):
    """CoarseMatching.get_offset_match / get_offset_match_work -- utils/coarse_matching.py:266-328 (one side)."""
    layer_num, bs = flow.shape[0], flow.shape[1]
    off = flow.view(layer_num, bs, -1, 4)
    conf = off[:, :, :, 2:].mean(dim=-1)
    if conf_mask is not None:
        conf.masked_fill_(~conf_mask.bool()[None].expand(layer_num, -1, -1), 100)
    off = off[:, :, :, :2]
    mask_conf = conf < 2
    for index in range(bs):
        mask_conf[:, index, 0] = True
    scale = hw_i0[0] / hw_c[0]
    l_ids, b_ids, i_ids = torch.where(mask_conf)
    j_coor = off[l_ids, b_ids, i_ids, :2] * scale
    i_coor = torch.stack([i_ids % hw_c[1], i_ids // hw_c[1]], dim=1) * scale
    out = {"offset_bids_" + side: b_ids, "offset_lids_" + side: l_ids, "conf" + side: conf[mask_conf]}
    if side == "right":
        out.update({"offset_kpts0_f_" + side: j_coor, "offset_kpts1_f_" + side: i_coor})
    else:
        out.update({"offset_kpts0_f_" + side: i_coor, "offset_kpts1_f_" + side: j_coor})
    return out


def aspanformer_forward(sd, cfg, data, with_fine_backbone=True):
    """ASpanFormer.forward, fine disabled, online_resize=True -- aspanformer.py:31-111.
    Returns the keys the reference writes into ``data`` (+ the transformer outputs for kernel-level checks)."""
    img0, img1 = data["image0"], data["image1"]
    assert img0.shape[0] == 1 and img1.shape[1] == 1                                  # aspanformer.py:43
    orig = [(im.shape[2], im.shape[3]) for im in (img0, img1)]
    img0, img1 = resize_df(img0), resize_df(img1)                                          # aspanformer.py:119-121
    tr = cfg["coarse"]["train_res"]
    tr_h, tr_w = (tr, tr) if len(tr) == 1 else (tr[0], tr[1])
    pos_scale0 = [tr_h / img0.shape[2], tr_w / img0.shape[3]]
    pos_scale1 = [tr_h / img1.shape[2], tr_w / img1.shape[3]]
    rs0 = torch.tensor([orig[0][1] / img0.shape[3], orig[0][0] / img0.shape[2]])[None]     # online_resize_scale (:128-129)
    rs1 = torch.tensor([orig[1][1] / img1.shape[3], orig[1][0] / img1.shape[2]])[None]
    bs = img0.size(0)
    hw0_i, hw1_i = tuple(img0.shape[2:]), tuple(img1.shape[2:])
    if hw0_i == hw1_i:
        c, _ = restate.resnet_fpn_8_2(sd, "backbone.", torch.cat([img0, img1], 0), with_fine_backbone)
        c0, c1 = c.split(bs)
    else:
        c0, _ = restate.resnet_fpn_8_2(sd, "backbone.", img0, with_fine_backbone)
        c1, _ = restate.resnet_fpn_8_2(sd, "backbone.", img1, with_fine_backbone)
    hw0_c, hw1_c = tuple(c0.shape[2:]), tuple(c1.shape[2:])
    d = cfg["coarse"]["d_model"]
    pe0 = position_encoding(d, pos_scale0)[:, :, :hw0_c[0], :hw0_c[1]]
    pe1 = position_encoding(d, pos_scale1)[:, :, :hw1_c[0], :hw1_c[1]]
    mask0 = mask1 = None
    if "mask0" in data:
        mask0, mask1 = data["mask0"].flatten(-2), data["mask1"].flatten(-2)
    f0, f1, flows = coarse_transformer(sd, "loftr_coarse.", c0 + pe0, c1 + pe1, pe0, pe1, cfg["coarse"], mask0, mask1)
    mc = cfg["match_coarse"]
    f0n, f1n = f0 / f0.shape[-1] ** .5, f1 / f1.shape[-1] ** .5                           # coarse_matching.py:107-108
    sim = torch.einsum("nlc,nsc->nls", f0n, f1n) * sd["coarse_matching.temperature"]
    if mask0 is not None:
        sim.masked_fill_(~(mask0[..., None] * mask1[:, None]).bool(), -INF)
    conf = F.softmax(sim, 1) * F.softmax(sim, 2)
    if "mask0" in data:
        raise NotImplementedError("mask_border_with_padding is exercised through the MatchFormer oracle")
    out = restate.coarse_match_from_conf(conf, hw0_c, hw1_c, hw0_i, mc["thr"], mc["border_rm"], data.get("scale0"),
                                         data.get("scale1"))
    out["conf_matrix"] = conf
    if flows[0].shape[2:4] == flows[1].shape[2:4]:
        out["predict_flow"] = torch.stack(flows, dim=0)
    else:
        out["predict_flow"] = flows
    out.update(offset_matches(flows[0], mask0, hw0_c, hw0_i, "left"))
    out.update(offset_matches(flows[1], mask1, hw0_c, hw0_i, "right"))
    out["mkpts0_c"], out["mkpts1_c"] = out["mkpts0_c"] * rs0, out["mkpts1_c"] * rs1     # in place on the shared tensors (:104-108)
    out.update({"feat_c0": f0, "feat_c1": f1, "hw0_c": hw0_c, "hw1_c": hw1_c, "mkpts0_f": out["mkpts0_c"],
                "mkpts1_f": out["mkpts1_c"], "image0": img0, "image1": img1})
    return out
