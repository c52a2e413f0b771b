"""Test-only stand-ins: the oracle's CPU restatements behind the ``ops`` signatures, so that the
HOST logic of the plugins (weight packing, BN folding, grouping, scatter order, dead-work
skipping) can be exercised by ``-m "not gpu"`` tests.  The product never uses these."""
import contextlib

import torch
import torch.nn.functional as F

from oracle import restate


def _rows_split(y):
    """fp32 rows [..., C] -> SplitAct with planes of the same shape."""
    from detectorfreesfm_amd.ops import SplitAct
    hi, lo = _split(y)
    return SplitAct(hi.half(), lo.half(), y.shape[-1])


def _put_split(dst, y):
    hi, lo = _split(y)
    dst.hi.copy_(hi.half().reshape(dst.hi.shape))
    dst.lo.copy_(lo.half().reshape(dst.lo.shape))


def _la(q, k, v, q_mask=None, kv_mask=None, q_group=1, kv_group=1, eps=1e-6, out=None, out_split=False):
    qm = None if q_mask is None else q_mask.repeat_interleave(q_group, dim=1)[:, :q.shape[1]]
    km = None if kv_mask is None else kv_mask.repeat_interleave(kv_group, dim=1)[:, :k.shape[1]]
    y = restate.linear_attention(q, k, v, qm, km, eps)
    if out_split:
        N, L, H, D = y.shape
        return _rows_split(y.reshape(N * L, H * D))
    if out is not None:
        out.copy_(y)
        return out
    return y


def _cm(feat0, feat1, hw0_c, hw1_c, thr, border, temperature, scale0=None, scale1=None, coarse_scale=8.0,
        mask0=None, mask1=None, defer=False):
    if defer:       # ops.PendingMatches: the result is asked for later
        out = _cm(feat0, feat1, hw0_c, hw1_c, thr, border, temperature, scale0, scale1, coarse_scale, mask0, mask1)
        import types
        return types.SimpleNamespace(result=lambda: out)
    from detectorfreesfm_amd.ops import SplitAct
    if isinstance(feat0, SplitAct):       # split planes carry the fp32 value to 2^-22: correlate what they hold
        feat0, feat1 = feat0.float(), feat1.float()
    hw0_i = (hw0_c[0] * coarse_scale, hw0_c[1] * coarse_scale)
    if mask0 is not None:
        N = feat0.shape[0]
        return restate.coarse_matching(feat0, feat1, hw0_c, hw1_c, hw0_i, thr, border, temperature, scale0, scale1,
                                       mask0=mask0.reshape(N, *hw0_c).bool(), mask1=mask1.reshape(N, *hw1_c).bool())
    return restate.coarse_matching(feat0, feat1, hw0_c, hw1_c, hw0_i, thr, border, temperature, scale0, scale1)


def _roi(feat, boxes, crop_h, crop_w, box_ind=None, out_slot=None, extrapolation_value=0.0, mean=None,
         std=None, out=None, channels_last=False):
    bi = torch.zeros(boxes.shape[0], dtype=torch.int32) if box_ind is None else box_ind
    p = restate.roi_align_crop(feat, boxes, bi, crop_h, crop_w, extrapolation_value)
    if mean is not None:
        p = (p - mean[:, None, None]) / std[:, None, None]
    if channels_last:
        p = p.permute(0, 2, 3, 1).contiguous()
    if out is None:
        return p
    if out_slot is None:
        out.copy_(p)
    else:
        out[out_slot] = p
    return out


def _fm(ref, qry, track_mask, movable, W, left, query_pts=None, scale_q=None, ref_pts=None, scale_r=None,
        rs_t=0, rs_n=0):
    from detectorfreesfm_amd.ops import SplitAct
    if isinstance(ref, SplitAct):            # the split-plane entry point: the planes' exact values
        ref, qry = ref.float(), qry.float()
    T, Vq = qry.shape[:2]
    mv = torch.ones(T, dtype=torch.bool) if movable is None else movable.bool()
    left_norm, coords, std, best = restate.fine_matching(ref, qry, W, left, track_mask.bool(), mv)
    out = {"best_index": best.int(), "left_norm": left_norm, "coords": coords, "std": std}
    if query_pts is not None:
        out["query_refined"] = query_pts + left_norm * (left // 2) * scale_q
    if ref_pts is not None:   # strided [Vq(+), T(+), 2] views, (rs_t, rs_n) = (1, Tfull)
        rp = ref_pts[:Vq, :T].transpose(0, 1)
        sr = scale_r[:Vq, :T].transpose(0, 1)
        out["ref_refined"] = rp + coords * (W // 2) * sr
    return out


def _ln(x, gamma, beta, eps=1e-5, residual=None, out=None, out_split=None, want_f32=True):
    from detectorfreesfm_amd.ops import SplitAct
    y = torch.nn.functional.layer_norm(x, (x.shape[-1],), gamma, beta, eps)
    if residual is not None:
        if isinstance(residual, SplitAct):
            residual = residual.float()
        y = residual.reshape(y.shape) + y
    if out_split is not None:
        _put_split(out_split, y)
    if out is None:
        return y if want_f32 else None
    out.copy_(y.reshape(out.shape))
    return out


def _linear_ln(x, pw, gamma, beta, eps=1e-5, residual=None, out=None, out_split=None):
    y = _linear(x, pw)
    return _ln(y, gamma, beta, eps, residual, out, out_split, want_f32=False)


def _split_rows(x, add=None, out=None, out_split=None):
    y = x if add is None else (x.reshape(-1, add.shape[0], x.shape[-1]) + add).reshape(x.shape)
    if out is not None:
        out.copy_(y.reshape(out.shape))
    if out_split is not None:
        _put_split(out_split, y)


def _scatter(a, b, slot, dst):
    v = a if b is None else a + b
    v = v.transpose(1, 2)
    if slot is None:
        dst[:v.shape[0]] = v
    else:
        dst[slot] = v
    return dst


def _split(x):
    hi = torch.where(x.abs() >= 2.0 ** -14, x, torch.zeros_like(x)).half().float()
    return hi, ((x - hi) * 2048.0).half().float()


def _to_split(y):
    from detectorfreesfm_amd.ops import SplitAct
    C = y.shape[-1]
    cp = (C + 7) // 8 * 8
    hi, lo = _split(y)
    H = torch.zeros((*y.shape[:3], cp), dtype=torch.float16)
    L = torch.zeros((*y.shape[:3], cp), dtype=torch.float16)
    H[..., :C], L[..., :C] = hi.half(), lo.half()
    return SplitAct(H, L, C)


def _conv(x, pw, stride=1, pad=0, residual=None, relu=False, out=None, out_split=False):
    """Exact algebra of dfsfm_conv2d_nhwc_f32: fp16x2 split of both operands, three products."""
    import torch.nn.functional as F
    from detectorfreesfm_amd.ops import SplitAct
    K = pw.kh * pw.kw * pw.Cin

    def unpack(t):      # tap-padded weights carry zero channels beyond the activation's
        w = t[:pw.Cout, :K].float().reshape(pw.Cout, pw.kh, pw.kw, pw.Cin)[..., :pw.Cin_act]
        return w.permute(0, 3, 1, 2).contiguous()
    wh, wl = unpack(pw.hi), unpack(pw.lo)
    if isinstance(x, SplitAct):
        xh, xl = x.hi.float().permute(0, 3, 1, 2), x.lo.float().permute(0, 3, 1, 2)
    else:
        xh, xl = _split(x.permute(0, 3, 1, 2))
    y = F.conv2d(xh, wh, None, stride, pad) + (F.conv2d(xh, wl, None, stride, pad) + F.conv2d(xl, wh, None, stride, pad)) / 2048.0
    y = y.permute(0, 2, 3, 1)
    if pw.bias is not None:
        y = y + pw.bias
    if residual is not None:
        y = y + (residual.float() if isinstance(residual, SplitAct) else residual).reshape(y.shape)
    if relu == 2:
        y = torch.nn.functional.leaky_relu(y, 0.01)
    elif relu:
        y = torch.relu(y)
    if isinstance(out_split, SplitAct):
        _put_split(out_split, y.reshape(out_split.hi.shape))
        return out_split
    if out_split:
        return _to_split(y)
    if out is None:
        return y.contiguous()
    out.copy_(y.reshape(out.shape))
    return out


def _linear(x, pw, residual=None, relu=False, out=None, out_split=False):
    from detectorfreesfm_amd.ops import SplitAct
    if isinstance(x, SplitAct):
        K = x.hi.shape[-1]
        x4 = SplitAct(x.hi.reshape(1, 1, -1, K), x.lo.reshape(1, 1, -1, K), K)
        nrows = x4.hi.shape[2]
    else:
        x4 = x.reshape(1, 1, -1, x.shape[-1])
        nrows = x4.shape[2]
    y = _conv(x4, pw, 1, 0, None if residual is None else residual.reshape(1, 1, nrows, -1), relu)
    y = y.reshape(nrows, pw.Cout)
    if out_split:
        return _rows_split(y)
    if out is None:
        return y
    out.copy_(y.reshape(out.shape))
    return out


def _maxpool(x):
    import torch.nn.functional as F
    from detectorfreesfm_amd.ops import SplitAct
    if isinstance(x, SplitAct):
        return _to_split(F.max_pool2d(x.float().permute(0, 3, 1, 2), 3, 2, 1).permute(0, 2, 3, 1))
    return F.max_pool2d(x.permute(0, 3, 1, 2), 3, 2, 1).permute(0, 2, 3, 1).contiguous()


def _s2d_front(patches, fw, c0, c1):
    """ops.s2d_front: both layers with the split algebra of ``_conv`` (conv1_1's weights unpacked from its MFMA fragments), then
    the centre window and the max-pool."""
    import torch.nn.functional as F
    pl = fw.w1f.reshape(2, 2, 2, 4, 16, 8).permute(2, 0, 1, 4, 3, 5).reshape(2, 64, 32).float()     # [plane][co][k]
    wh, wl = (pl[i][:, :27].reshape(64, 3, 3, 3).permute(0, 3, 1, 2).contiguous() for i in (0, 1))   # [co, ci, ky, kx]
    xh, xl = _split(patches.permute(0, 3, 1, 2))
    r1 = F.conv2d(xh, wh, None, 1, 1) + (F.conv2d(xh, wl, None, 1, 1) + F.conv2d(xl, wh, None, 1, 1)) / 2048.0
    r1 = torch.relu(r1 + fw.b1[None, :, None, None]).permute(0, 2, 3, 1)
    y = _conv(_to_split(r1), fw.conv2, 1, 1, relu=True, out_split=True)
    return y.crop(c0, c1, c0, c1), _maxpool(y)


def _resample(y, By, Bx, out=None):
    r = torch.einsum("ab,cd,mbdk->mack", By, Bx, y).reshape(y.shape[0], By.shape[0] * Bx.shape[0], y.shape[-1])
    if out is None:
        return r
    out.copy_(r)
    return out


def _dwconv(x, w, bias, mode=0, out_split=False):
    import torch.nn.functional as F
    C = x.shape[-1]
    w4 = w.reshape(C, 1, 3, 3) if w.dim() == 4 else w.t().reshape(C, 1, 3, 3)
    xx = x.permute(0, 3, 1, 2)
    y = F.conv2d(xx, w4, bias, 1, 1, 1, C)
    y = (y, xx * torch.sigmoid(y), F.gelu(y))[mode].permute(0, 2, 3, 1).contiguous()
    return _to_split(y) if out_split else y


def _bilinear(x, hout, wout):
    import torch.nn.functional as F
    return F.interpolate(x.permute(0, 3, 1, 2), size=(hout, wout), mode="bilinear", align_corners=True).permute(0, 2, 3, 1).contiguous()


def _resample_u8(src, bounds_x, kk_x, bounds_y, kk_y, out_u8=False, lut=None, pad_hw=None, want_mask=False):
    """ops.resample_u8 from the int32 tables alone (the arithmetic of csrc/image_resize.hip, in numpy)."""
    import numpy as np
    a = src.numpy()
    a = a[:, :, None] if a.ndim == 2 else a

    def one_pass(img, bounds, kk, axis):
        s = np.moveaxis(img, axis, 0).astype(np.int64)
        d = np.empty((bounds.shape[0],) + s.shape[1:], dtype=np.uint8)
        for i in range(bounds.shape[0]):
            lo, n = int(bounds[i, 0]), int(bounds[i, 1])
            ss = (1 << 21) + np.tensordot(kk[i, :n].astype(np.int64), s[lo:lo + n], axes=(0, 0))
            d[i] = np.clip(ss >> 22, 0, 255)
        return np.moveaxis(d, 0, axis)
    r = one_pass(one_pass(a, bounds_x.numpy(), kk_x.numpy(), 1), bounds_y.numpy(), kk_y.numpy(), 0)
    o8 = torch.from_numpy(r[:, :, 0] if src.dim() == 2 else r) if out_u8 else None
    of = mk = None
    if lut is not None:
        Hn, Wn, C = r.shape
        ph, pw = (Hn, Wn) if pad_hw is None else pad_hw
        of = torch.zeros((C, ph, pw), dtype=torch.float32)
        of[:, :Hn, :Wn] = lut[torch.from_numpy(r.astype(np.int64))].permute(2, 0, 1)
        if want_mask:
            mk = torch.zeros((ph, pw), dtype=torch.float32)
            mk[:Hn, :Wn] = 1
    return o8, of, mk


# ---------------------------------------------------------------- ASpanFormer pieces (csrc/aspan_ops.hip)
def _avgpool(x, k, out=None):
    import torch.nn.functional as F
    y = F.avg_pool2d(x.permute(0, 3, 1, 2).contiguous(), k, stride=k).permute(0, 2, 3, 1).contiguous()
    if out is not None:
        out.copy_(y.view(out.shape))
        return out
    return y


def _full_attention(q, k, v, nhead, scale, kv_swap=False):
    from oracle import restate_aspanformer as ra
    N = q.shape[0]
    outs = []
    for n in range(N):
        o = n ^ 1 if kv_swap else n
        y = ra.full_attention(q[n:n + 1].transpose(1, 2).contiguous(), k[o:o + 1].transpose(1, 2).contiguous(),
                              v[o:o + 1].transpose(1, 2).contiguous(), nhead, temp=scale * (q.shape[2] // nhead) ** .5)
        outs.append(y.transpose(1, 2))
    return torch.cat(outs, 0).contiguous()


def _span_attention(q, hw, k, v, hw_k, flow, hw0, sample_offset, nhead, nsample, radius_scale, temp=1.0, kv_swap=False):
    """One level of HierachicalAttention with the reference's own operations (aspan_module/attention.py:49-66, 92-133);
    batched [N, rows, C] inputs pair image n with the keys / values of image n ^ kv_swap."""
    import torch.nn.functional as F
    from oracle import restate_aspanformer as ra
    if q.dim() == 3:
        N = q.shape[0]
        fl = flow.reshape(N, -1, 4)
        return torch.stack([_span_attention(q[n], hw, k[n ^ 1 if kv_swap else n], v[n ^ 1 if kv_swap else n], hw_k, fl[n], hw0,
                                            sample_offset, nhead, nsample, radius_scale, temp) for n in range(N)], 0)
    (h, w), (hk, wk), (H0, W0) = hw, hw_k, hw0
    s = H0 // h
    C = q.shape[-1]
    fl = flow.reshape(1, H0, W0, 4)
    variance = torch.exp(0.5 * fl[..., 2:]) * radius_scale
    span_scale = torch.clamp(variance * 2 / nsample[1], min=1)
    ker = s * nsample[0]
    off = F.avg_pool2d(fl[..., :2].permute(0, 3, 1, 2), kernel_size=ker, stride=ker).permute(0, 2, 3, 1) / s
    span = F.avg_pool2d(span_scale.permute(0, 3, 1, 2), kernel_size=ker, stride=ker).permute(0, 2, 3, 1)
    nchw = lambda t, hh, ww: t.reshape(1, hh, ww, C).permute(0, 3, 1, 2).contiguous()
    sd = {"sample_offset": sample_offset}
    qq, kk, vv, _ = ra.partition_token(sd, "", nchw(q, h, w), nchw(k, hk, wk), nchw(v, hk, wk), off, span, None, nhead, nsample)
    y = ra.group_attention(qq, kk, vv, temp, C)                               # [1, C, h*w] in (group, member) order
    return y[0].transpose(0, 1).contiguous()


def _layernorm2d(x, affine, bias, residual=None, out=None, out_split=None, want_f32=True):
    mean, std = x.mean(dim=-1, keepdim=True), x.std(dim=-1, keepdim=True)
    y = affine * (x - mean) / (std + 1e-6) + bias
    if residual is not None:
        y = residual.float() + y
    if out_split is not None:
        _put_split(out_split, y)
    if out is not None:
        out.copy_(y.view(out.shape))
        return out
    return y.contiguous() if want_f32 else None


def _upsample(x, scale, bilinear, out=None, out_split=None, want_f32=True):
    import torch.nn.functional as F
    xc = x.permute(0, 3, 1, 2).contiguous()
    y = F.interpolate(xc, scale_factor=scale, mode="bilinear") if bilinear else F.interpolate(xc, scale_factor=scale, mode="nearest")
    y = y.permute(0, 2, 3, 1).contiguous()
    if out_split is not None:
        _put_split(out_split, y.reshape(-1, y.shape[-1]))
    if out is not None:
        out.copy_(y.view(out.shape))
        return out
    return y if want_f32 else None


def _resize_bilinear(x, hout, wout):
    import torch.nn.functional as F
    return F.interpolate(x.reshape(-1, 1, *x.shape[-2:]), size=[hout, wout], mode="bilinear", align_corners=False).reshape(
        *x.shape[:-2], hout, wout)


def _flow_decode(x, wk, hk):
    return torch.cat([torch.sigmoid(x[:, :2]) * torch.tensor([float(wk), float(hk)]), x[:, 2:4]], dim=1).contiguous()


def _merge(rows, img0, img1, n_images):
    """ops.merge_keypoints on the numpy restatement of the reference's consumer stage (oracle/restate_merge.py)."""
    from oracle import restate_merge as rm
    kp, sc, off, ids = rm.merge_keypoints(rows.numpy().astype("float32").reshape(-1, 5), img0.numpy().astype("int32"),
                                          img1.numpy().astype("int32"), int(n_images))
    return torch.from_numpy(kp), torch.from_numpy(sc), torch.from_numpy(off), torch.from_numpy(ids)


def _enc_kv(src, fw, kv_mask=None, kv_group=1):
    """Stand-in for ops.encoder_kv: the attention state is simply the source tokens and their mask (the layer is evaluated
    in _enc_apply with the oracle's restatement on the values the split planes hold)."""
    return {"src": src.float().clone(), "mask": kv_mask, "group": kv_group}


def _enc_apply(x, fw, state, S, q_mask=None, q_group=1, out_split=None, out=None, eps=1e-5, attn_eps=1e-6, debug_stage=0):
    xv = x.float()
    N, L, _ = xv.shape
    xm = None if q_mask is None else q_mask.repeat_interleave(q_group, dim=1)[:, :L]
    sm = None if state["mask"] is None else state["mask"].repeat_interleave(state["group"], dim=1)[:, :state["src"].shape[1]]
    y = restate.encoder_layer(fw.values, "", xv, state["src"], 8, xm, sm)
    if out_split is not None:
        _put_split(out_split, y)
    if out is not None:
        hi, lo = _split(y)
        out.copy_((hi + lo / 2048.0).reshape(out.shape))          # the kernel's fp32 form is the exact value of the planes
    return None


def _enc256_state(k, v, kv_mask=None, kv_group=1):
    """Stand-in for ops.encoder256_state: the attention state is the projected k, v and their mask."""
    return {"k": k.float().clone(), "v": v.float().clone(), "mask": kv_mask, "group": kv_group}


def _enc256_kv(src, fw, kv_mask=None, kv_group=1):
    """Stand-in for ops.encoder256_kv: k | v projection of the source tokens on the values the kv stream holds."""
    sv = src.float()
    return {"k": F.linear(sv, fw.values["k_proj.weight"]), "v": F.linear(sv, fw.values["v_proj.weight"]), "mask": kv_mask,
            "group": kv_group}


def _enc256_apply(x, fw, state, S, q_mask=None, q_group=1, out_split=None, out=None, eps=1e-5, attn_eps=1e-6, debug_stage=0):
    """Stand-in for ops.encoder256_apply: the query side of LoFTREncoderLayer.forward (transformer.py:35-58) on the values the
    fragment stream holds (fw.values), attention through the oracle's linear_attention."""
    xv = x.float()
    N, L, C = xv.shape
    W = fw.values
    Sk = state["k"].shape[1]
    xm = None if q_mask is None else q_mask.repeat_interleave(q_group, dim=1)[:, :L]
    sm = None if state["mask"] is None else state["mask"].repeat_interleave(state["group"], dim=1)[:, :Sk]
    q = F.linear(xv, W["q_proj.weight"]).view(N, L, 8, C // 8)
    msg = restate.linear_attention(q, state["k"].reshape(N, Sk, 8, C // 8), state["v"].reshape(N, Sk, 8, C // 8), xm, sm)
    msg = F.linear(msg.reshape(N, L, C), W["merge.weight"])
    msg = F.layer_norm(msg, (C,), W["norm1.weight"], W["norm1.bias"], eps)
    msg = F.linear(F.relu(F.linear(torch.cat([xv, msg], 2), W["mlp.0.weight"])), W["mlp.2.weight"])
    y = xv + F.layer_norm(msg, (C,), W["norm2.weight"], W["norm2.bias"], eps)
    if out_split is not None:
        _put_split(out_split, y)
    if out is not None:
        hi, lo = _split(y)
        out.copy_((hi + lo / 2048.0).reshape(out.shape))
    return None


def _jpeg_decode(pl, out_channels, device, sweeps=4, max_calls=8):
    """ops.jpeg_decode on the CPU lane model of the decoder (tests/jpeg_emul.cpp: the device thread functions compiled with g++),
    thread order reversed so that every sweep sees only the previous sweep's states, like a launch whose threads all start
    together."""
    import jpeg_emul
    out, info = jpeg_emul.decode(pl, out_channels == 3, sweeps=sweeps, order=1, max_calls=max_calls)
    st = info["status"]
    from detectorfreesfm_amd import jpeg
    if st[0] != 0:
        raise jpeg.CorruptJpeg("entropy decode did not reach its fixed point")
    if st[1] or st[2]:
        raise jpeg.CorruptJpeg(f"corrupt scan: {st[1]} invalid codes, {st[2]} restart intervals with a wrong block count")
    return torch.from_numpy(out), dict(sweeps=info["sweeps"], calls=info["calls"])


@contextlib.contextmanager
def cpu_ops():
    from detectorfreesfm_amd import ops
    saved = {n: getattr(ops, n) for n in ("linear_attention", "coarse_match", "roi_align", "fine_match",
                                          "layernorm", "add_scatter_tokens", "conv2d_nhwc", "linear",
                                          "maxpool3x3s2_nhwc", "split_rows", "linear_ln", "merge_keypoints", "resample_separable", "dwconv3x3",
                                          "bilinear_up", "resample_u8", "avgpool", "full_attention",
                                          "span_attention", "layernorm2d", "upsample", "flow_decode", "resize_bilinear",
                                          "encoder_kv", "encoder_apply", "encoder256_state", "encoder256_apply", "encoder256_kv",
                                          "jpeg_decode", "s2d_front")}
    ops.jpeg_decode = _jpeg_decode
    ops.s2d_front = _s2d_front
    ops.linear_attention, ops.coarse_match, ops.roi_align, ops.fine_match = _la, _cm, _roi, _fm
    ops.layernorm, ops.add_scatter_tokens = _ln, _scatter
    ops.conv2d_nhwc, ops.linear, ops.maxpool3x3s2_nhwc = _conv, _linear, _maxpool
    ops.split_rows, ops.linear_ln, ops.merge_keypoints = _split_rows, _linear_ln, _merge
    ops.resample_separable = _resample
    ops.dwconv3x3, ops.bilinear_up = _dwconv, _bilinear
    ops.resample_u8 = _resample_u8
    ops.avgpool, ops.full_attention, ops.span_attention = _avgpool, _full_attention, _span_attention
    ops.layernorm2d, ops.upsample, ops.flow_decode = _layernorm2d, _upsample, _flow_decode
    ops.resize_bilinear = _resize_bilinear
    ops.encoder_kv, ops.encoder_apply = _enc_kv, _enc_apply
    ops.encoder256_state, ops.encoder256_apply, ops.encoder256_kv = _enc256_state, _enc256_apply, _enc256_kv
    try:
        yield
    finally:
        for n, f in saved.items():
            setattr(ops, n, f)
