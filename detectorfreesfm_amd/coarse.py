"""MI355X coarse matcher behind the reference's ``NEUSFM_coarse_matcher`` plugin surface.

``HipLoFTR`` is a drop-in for the reference's ``LoFTR`` module in coarse_only mode
(third_party/LoFTR/src/loftr/loftr.py:12-87): same constructor argument (the lower-cased config
dict), same ``state_dict`` layout (``load_state_dict(strict=True)`` of the official checkpoints,
``matcher.`` prefix stripped), same in-place ``forward(data)`` contract
(src/coarse_match/coarse_match_worker.py:83-99 reads ``m_bids, mkpts0_f, mkpts1_f, mconf``).

Every kernel of the path is hand-written HIP (``libdfsfm_hip.so``): the ResNet convolutions and all ``nn.Linear``
GEMMs on the fp16x2-split MFMA implicit-GEMM kernels (folded BN, ReLU, residual, LayerNorm in the epilogue), linear
attention (K1) inside every encoder layer, and the fused correlation / dual-softmax / mutual-NN / keypoint stage
(K3-K5).  There is no library (MIOpen / hipBLASLt) path in this package.  Output-identical work the reference wastes is skipped:
the FPN top-down branch (dead when fine.enable=False, resnet_fpn.py:110-116) and the dense
``conf_matrix`` (never read by an inference caller).
"""
import math



import torch

from . import ops
from .params import ParamModule, loftr_param_spec


def position_encoding_sine(d_model: int, max_shape=(256, 256), temp_bug_fix: bool = False) -> torch.Tensor:
    """2-D sinusoidal encoding buffer [1,d_model,H,W] with the semantics of PositionEncodingSine
    (third_party/LoFTR/src/loftr/utils/position_encoding.py:22-35).  With temp_bug_fix=False the
    exponent uses ``-ln(1e4) / d_model // 2`` (floor division binds last), which the released
    weights were trained with and the plugin builder selects (coarse_match_worker.py:35)."""
    ys = torch.arange(1, max_shape[0] + 1, dtype=torch.float32)[:, None].expand(*max_shape)
    xs = torch.arange(1, max_shape[1] + 1, dtype=torch.float32)[None, :].expand(*max_shape)
    k = torch.arange(0, d_model // 2, 2, dtype=torch.float32)
    rate = (-math.log(10000.0) / (d_model // 2)) if temp_bug_fix else (-math.log(10000.0) / d_model // 2)
    div = torch.exp(k * rate)[:, None, None]
    pe = torch.zeros((d_model, *max_shape))
    pe[0::4] = torch.sin(xs * div)
    pe[1::4] = torch.cos(xs * div)
    pe[2::4] = torch.sin(ys * div)
    pe[3::4] = torch.cos(ys * div)
    return pe[None]


def _fold_bn(w, bn_w, bn_b, mean, var, eps=1e-5):
    """conv -> eval BatchNorm == conv with scaled weights + bias.  Folded in float64: the products w*s are then split
    into the fp16 hi/lo planes (ops.PackedDense) from their exact values instead of from an fp32 rounding of them -- the
    fold adds no rounding of its own to the 22-bit weight representation (tools/studies/aspan_noise_study.py: every
    backbone layer's rounding is amplified ~500x by ASpanFormer's transformer)."""
    s = bn_w.double() / torch.sqrt(var.double() + eps)
    return (w.double() * s[:, None, None, None]).contiguous(), (bn_b.double() - mean.double() * s).contiguous()


# Module flags (no environment reads): False keeps d_model-128 layers on the five-GEMM path that MatchFormer / ASpanFormer use
# for their other widths; FUSED_ENCODER256 the same for the d_model-256 layers of the coarse transformer (csrc/encoder256.hip);
# FUSED_KV256 = False keeps only their source side on the k | v projection GEMM + K1's partial sums (encoder256_state).  Every
# setting is a tested path: tests/test_gpu_encoder256.py and tests/test_gpu_encoder_fused.py run the layers both ways.
FUSED_ENCODER = True
FUSED_ENCODER256 = True
FUSED_KV256 = True


class EncoderLayerWeights:
    """Packed weights of one LoFTREncoderLayer (transformer.py:7-33): fp16x2-split operands of dfsfm_conv2d_nhwc_f32 (1x1 case)."""

    def __init__(self, get, prefix):
        wq, wk, wv = get(prefix + "q_proj.weight"), get(prefix + "k_proj.weight"), get(prefix + "v_proj.weight")
        self.pq, self.pkv = ops.PackedDense(wq), ops.PackedDense(torch.cat([wk, wv], 0))
        self.pqkv = ops.PackedDense(torch.cat([wq, wk, wv], 0))
        self.pmerge, self.p2 = ops.PackedDense(get(prefix + "merge.weight")), ops.PackedDense(get(prefix + "mlp.2.weight"))
        self.p1 = ops.PackedDense(get(prefix + "mlp.0.weight"))
        self.n1 = (get(prefix + "norm1.weight").contiguous(), get(prefix + "norm1.bias").contiguous())
        self.n2 = (get(prefix + "norm2.weight").contiguous(), get(prefix + "norm2.bias").contiguous())
        # d_model 128 (the refinement head): the whole layer runs as two fused kernels (csrc/encoder_fused.hip)
        self.fused = None
        if wq.shape == (ops.ENC_C, ops.ENC_C) and FUSED_ENCODER:
            self.fused = ops.EncoderFusedWeights(wq, wk, wv, get(prefix + "merge.weight"), get(prefix + "mlp.0.weight"),
                                                 get(prefix + "mlp.2.weight"), self.n1, self.n2)
        # d_model 256 (the coarse transformer): the query side of the layer is ONE fused kernel (csrc/encoder256.hip); the
        # k | v projection stays a split-plane GEMM (pkv)
        self.fused256 = None
        if wq.shape == (ops.ENC256_C, ops.ENC256_C) and FUSED_ENCODER256:
            self.fused256 = ops.Encoder256Weights(wq, get(prefix + "merge.weight"), get(prefix + "mlp.0.weight"),
                                                  get(prefix + "mlp.2.weight"), self.n1, self.n2, wk=wk, wv=wv)


def encoder_layer_split(w: EncoderLayerWeights, xs, src, out_x, out_xs, nhead, x_mask=None, source_mask=None,
                        q_group=1, kv_group=1, is_self=False):
    """LoFTREncoderLayer.forward (LoFTR transformer.py:35-58; multiview copy
    src/MultiviewMatcher/matcher_module/transformer.py:66-95) with every GEMM on the split-plane LDS-DMA kernels, concat-free
    (``torch.cat([x, message])`` never happens: norm1(message) lands in the second half of the [x | message] buffer the MLP
    GEMM reads), q|k|v fused for self attention, k|v for cross, ReLU in the GEMM epilogue: the token state lives ONLY as split
    fp16 planes (ops.SplitAct, value = hi + lo/2048), written by the epilogue of whichever kernel computes it
    (LayerNorm, attention apply, GEMM) -- no conversion pass, no concat, no fp32 copy of the residual chain.

    xs  SplitAct [N,L,2C]       first half = x (also the residual); second half receives norm1(message)
    src SplitAct [N,S,C] view   source tokens (== xs.cols(0,C) for self-attention)
    out_x  fp32 [N,L,C] view or None: receives x + norm2(mlp(...)) (the last layer's features)
    out_xs SplitAct [N,L,C] view or None: the same values as split planes for the next layer"""
    N, L, C2 = xs.hi.shape
    C = C2 // 2
    D = C // nhead
    S = src.hi.shape[1]
    xs_x = xs.cols(0, C)
    fused128 = w.fused is not None and nhead == 8 and L >= 32 and (out_x is not None or out_xs is not None)
    fused256 = w.fused256 is not None and nhead == 8 and L >= 16 and (out_x is not None or out_xs is not None)
    if (fused128 or fused256) and ops.range_check_active():
        # Range guard of the fused kernels: they split q, k, v, the message, norm1(merge) and the ReLU hidden layer in
        # REGISTERS with a saturating clamp -- nothing a producer-side check can see.  While a range sweep is open (the first call
        # after load_state_dict) or DFSFM_DEBUG_RANGE is on, the same layer ALSO runs on the five-GEMM path below, whose every
        # intermediate is a checked producer; its results are discarded (scratch output; the unused [norm1] half of ``xs``).
        scratch = ops.SplitAct.empty_rows((N, L), C, xs.hi.device)
        _encoder_layer_unfused(w, xs, src, None, scratch, nhead, x_mask, source_mask, q_group, kv_group, is_self)
    if fused128:
        # two launches: source tokens -> per-sequence attention state; x tokens -> layer output.  q, k, v, the message, the
        # merged message and the MLP's hidden layer never reach memory (the second half of ``xs`` stays unused)
        state = ops.encoder_kv(src, w.fused, source_mask, kv_group)
        ops.encoder_apply(xs_x, w.fused, state, S, x_mask, q_group, out_split=out_xs, out=out_x)
        return out_x
    if fused256:
        # source side: k | v projection fused with K1's partial sums (+ the small image kernel); query side: q, attention,
        # merge, LayerNorm, MLP, LayerNorm and the residual in one kernel.  q, k, v, the message, [x | norm1] and the hidden
        # layer never reach memory; the second half of ``xs`` stays unused
        if FUSED_KV256 and w.fused256.kv_stream is not None:
            state = ops.encoder256_kv(src, w.fused256, source_mask, kv_group)        # k, v never reach memory either
        else:
            kv = ops.linear(src, w.pkv).view(N, S, 2 * C)
            state = ops.encoder256_state(kv[..., :C], kv[..., C:], source_mask, kv_group)
        ops.encoder256_apply(xs_x, w.fused256, state, S, x_mask, q_group, out_split=out_xs, out=out_x)
        return out_x
    return _encoder_layer_unfused(w, xs, src, out_x, out_xs, nhead, x_mask, source_mask, q_group, kv_group, is_self)


def encoder_layer_fused128(w: EncoderLayerWeights, x, src, out_xs, x_mask=None, source_mask=None, q_group=1, kv_group=1):
    """The fused d_model-128 form of ``encoder_layer_split`` on token planes of ANY row stride (``x`` [N,L,C], ``src`` [N,S,C] SplitAct
    views): what a caller uses whose tokens do not sit in an [x | message] buffer -- the refinement head's first layer reads the
    backbone's split-plane output directly.  No range-guard shadow pass: callers take ``encoder_layer_split`` while a sweep is open."""
    state = ops.encoder_kv(src, w.fused, source_mask, kv_group)
    ops.encoder_apply(x, w.fused, state, src.hi.shape[1], x_mask, q_group, out_split=out_xs)
    return out_xs


def _encoder_layer_unfused(w, xs, src, out_x, out_xs, nhead, x_mask, source_mask, q_group, kv_group, is_self):
    """The five-GEMM + K1 form of ``encoder_layer_split`` (every intermediate reaches memory through a range-checked producer)."""
    N, L, C2 = xs.hi.shape
    C = C2 // 2
    D = C // nhead
    S = src.hi.shape[1]
    xs_x = xs.cols(0, C)
    if is_self:
        qkv = ops.linear(xs_x, w.pqkv).view(N, L, 3 * C)
        q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
    else:
        q = ops.linear(xs_x, w.pq).view(N, L, C)
        kv = ops.linear(src, w.pkv).view(N, S, 2 * C)
        k, v = kv[..., :C], kv[..., C:]
    if ops.range_check_active() and (w.fused is not None or w.fused256 is not None):
        # what only the fused kernels hold as split planes: phi(q) = elu(q) + 1 <= |q| + 1, phi(k), v / S, and the per-head state
        # KV = sum_s phi(k_s) v_s / S, a mean over the source tokens and therefore bounded by max phi(k) * max |v|
        qa, ka, va = q.abs().max(), k.abs().max(), v.abs().max()
        ops.range_note("fused encoder layer: phi(q)", qa + 1.0)
        ops.range_note("fused encoder layer: phi(k)", ka + 1.0)
        ops.range_note("fused encoder layer: v", va)
        ops.range_note("fused encoder layer: KV state bound max phi(k) * max |v|", (ka + 1.0) * va)
    msg = ops.linear_attention(q.unflatten(-1, (nhead, D)), k.unflatten(-1, (nhead, D)),
                               v.unflatten(-1, (nhead, D)), x_mask, source_mask, q_group, kv_group, out_split=True)
    fuse_ln = C in (64, 128, 256)    # a row fits one workgroup tile: LayerNorm runs in the GEMM epilogue
    if fuse_ln:
        ops.linear_ln(msg, w.pmerge, w.n1[0], w.n1[1], out_split=xs.cols(C, 2 * C))        # norm1 -> [x | message]
    else:
        merged = ops.linear(msg, w.pmerge)
        ops.layernorm(merged, w.n1[0], w.n1[1], out_split=xs.cols(C, 2 * C), want_f32=False)
    h = ops.linear(xs, w.p1, relu=True, out_split=True)                                   # relu(mlp.0([x|message]))
    if fuse_ln:
        ops.linear_ln(h, w.p2, w.n2[0], w.n2[1], residual=xs_x, out=out_x, out_split=out_xs)
    else:
        o = ops.linear(h, w.p2)
        ops.layernorm(o.view(N, L, C), w.n2[0], w.n2[1], residual=xs_x, out=out_x, out_split=out_xs,
                      want_f32=False)
    return out_x


def fold_backbone(g):
    """ResNetFPN_8_2 weights (backbone/resnet_fpn.py:43-118) with eval-mode BatchNorm folded into (weight, bias) pairs;
    ``g(name)`` fetches a tensor by its state_dict name.  Shared by HipLoFTR and HipASpanFormer (ASpanFormer's backbone file
    is identical)."""
    P = {}

    def conv_bn(conv, bn):
        return _fold_bn(g(conv + ".weight"), g(bn + ".weight"), g(bn + ".bias"),
                        g(bn + ".running_mean"), g(bn + ".running_var"))
    P["stem"] = conv_bn("backbone.conv1", "backbone.bn1")
    for li in (1, 2, 3):
        for bi in (0, 1):
            q = f"backbone.layer{li}.{bi}"
            blk = {"c1": conv_bn(q + ".conv1", q + ".bn1"), "c2": conv_bn(q + ".conv2", q + ".bn2"),
                   "stride": 2 if (li > 1 and bi == 0) else 1}
            if blk["stride"] != 1:
                blk["down"] = conv_bn(q + ".downsample.0", q + ".downsample.1")
            P[f"l{li}b{bi}"] = blk
    P["l3out"] = g("backbone.layer3_outconv.weight")
    return P


def pack_backbone_hip(P, same_conv=True):
    """The folded backbone weights as split-plane operands of the conv kernels."""
    def pk(wb, split_in=True, same=False):    # weights for a conv whose input arrives as a SplitAct
        w, b = wb if isinstance(wb, tuple) else (wb, None)   # same: stride-1 3x3 -> activation-reuse kernel
        return ops.PackedDense(w, b, cin_pad=(w.shape[1] + 7) // 8 * 8 if split_in else None,
                               tap_padded=same and same_conv)
    H = {"stem": pk(P["stem"], split_in=False), "l3out": pk(P["l3out"])}
    for li in (1, 2, 3):
        for bi in (0, 1):
            b = P[f"l{li}b{bi}"]
            hb = {"c1": pk(b["c1"], same=b["stride"] == 1), "c2": pk(b["c2"], same=True), "stride": b["stride"]}
            if "down" in b:
                hb["down"] = pk(b["down"])
            H[f"l{li}b{bi}"] = hb
    return H


def backbone_tokens_hip(x, H):
    """x [N,1,H,W] -> coarse feature map as fp32 tokens [N, H/8, W/8, C] (NHWC == token-major): every conv is one launch
    with folded BN, ReLU and the residual fused; activations between convolutions travel as split fp16 planes."""
    t = x.permute(0, 2, 3, 1)                       # C=1: NCHW memory is already NHWC
    t = ops.conv2d_nhwc(t, H["stem"], 2, 3, relu=True, out_split=True)
    for li in (1, 2, 3):
        for bi in (0, 1):
            b = H[f"l{li}b{bi}"]
            y = ops.conv2d_nhwc(t, b["c1"], b["stride"], 1, relu=True, out_split=True)
            sc = ops.conv2d_nhwc(t, b["down"], b["stride"], 0, out_split=True) if "down" in b else t
            t = ops.conv2d_nhwc(y, b["c2"], 1, 1, residual=sc, relu=True, out_split=True)
    return ops.conv2d_nhwc(t, H["l3out"], 1, 0)     # fp32 tokens for the transformer


class HipLoFTR(ParamModule):
    def __init__(self, config: dict):
        super().__init__()
        # tap-reuse ("same") schedule for the stride-1 3x3 convs; False packs them for the flattened-K kernel (set before the
        # first forward; tests/test_gpu_e2e.py::test_flattened_k_conv_schedule_equals_same_schedule)
        self.same_conv = True
        if config["match_coarse"]["match_type"] != "dual_softmax":
            raise NotImplementedError("only the dual_softmax coarse matcher is on the hot path")
        if config["fine"]["enable"]:
            raise NotImplementedError("HipLoFTR implements the coarse_only configuration "
                                      "(loftr_ds_coarse_only.py: LOFTR.FINE.ENABLE=False)")
        if config["coarse"]["attention"] != "linear":
            raise NotImplementedError("only linear attention")
        self.config = config
        self.register_spec(loftr_param_spec(config))
        self.register_buffer("pe", position_encoding_sine(config["coarse"]["d_model"],
                                                          temp_bug_fix=config["coarse"]["temp_bug_fix"]),
                             persistent=False)
        self._packed = None

    # -- checkpoint compatibility (loftr.py:83-87) -------------------------------------------
    def load_state_dict(self, state_dict, *args, **kwargs):
        sd = {}
        for k, v in state_dict.items():
            sd[k.replace("matcher.", "", 1) if k.startswith("matcher.") else k] = v
        self._packed = None
        return super().load_state_dict(sd, *args, **kwargs)

    def _apply(self, fn, *a, **kw):
        self._packed = None
        return super()._apply(fn, *a, **kw)

    # -- weight packing -----------------------------------------------------------------------
    def _pack(self):
        g = self.p
        P = fold_backbone(g)
        n_layers = len(self.config["coarse"]["layer_names"])
        P["enc"] = [EncoderLayerWeights(g, f"loftr_coarse.layers.{i}.") for i in range(n_layers)]
        P["hip"] = pack_backbone_hip(P, self.same_conv)
        self._packed = P
        return P

    # -- K6 on the hand-written NHWC implicit-GEMM kernel (conv + folded BN + ReLU + residual fused) --
    def _backbone_hip(self, x, P):
        """x [N,1,H,W] -> coarse feature map as tokens [N, H/8, W/8, C] (NHWC == token-major).
        Activations between convolutions travel as split fp16 planes (ops.SplitAct): the producer's
        epilogue splits once, consumers DMA the planes straight into LDS."""
        return backbone_tokens_hip(x, P["hip"])

    # -- K2/K1: LocalFeatureTransformer.forward (transformer.py:80-101) -------------------------
    def _transformer(self, f0, f1, P, pe0=None, pe1=None, mask0=None, mask1=None, want_f32=True):
        """f0 [N,L,C], f1 [N,S,C] (+ optional positional-encoding tables added on the way in) -> updated
        features (contiguous fp32).  When both images have the same grid they share buffers so that
        self layers run as ONE batch of 2N sequences.  Split planes [.,.,2C] = [x | norm1(message)] ping-pong, fp32 only
        for the result (encoder_layer_split).  mask0 [N,L] / mask1 [N,S] (uint8 / bool, 1 = valid): the padding masks of
        transformer.py:80-97 -- query mask and source mask of every layer's linear attention (K1).
        want_f32=False: the last layer writes only the split planes the correlation reads (``self._feat_split``); the fp32 copy of the
        final features (79 MB per 8 pairs) is for callers of ``coarse_features`` and is not produced; returns (None, None)."""
        nhead = self.config["coarse"]["nhead"]
        names = self.config["coarse"]["layer_names"]
        N, L, C = f0.shape
        S = f1.shape[1]
        same = L == S
        dev = f0.device

        def new_f32(width):
            if same:
                big = torch.empty((2 * N, L, width), dtype=torch.float32, device=dev)
                return big, big[:N], big[N:]
            return None, torch.empty((N, L, width), dtype=torch.float32, device=dev), \
                torch.empty((N, S, width), dtype=torch.float32, device=dev)

        def new_split(width):
            if same:
                big = ops.SplitAct.empty_rows((2 * N, L), width, dev)
                return big, big[:N], big[N:]
            return None, ops.SplitAct.empty_rows((N, L), width, dev), ops.SplitAct.empty_rows((N, S), width, dev)

        XS, XSn = new_split(2 * C), new_split(2 * C)
        ops.split_rows(f0, pe0, out_split=XS[1].cols(0, C))
        ops.split_rows(f1, pe1, out_split=XS[2].cols(0, C))
        fin = new_split(C)      # the final features as contiguous split planes: operands of the correlation
        out = (None, None, None)
        m01 = torch.cat([mask0, mask1], 0) if (mask0 is not None and same) else None
        for li, (w, name) in enumerate(zip(P["enc"], names)):
            last = li == len(names) - 1
            oxs = fin if last else tuple(None if b is None else b.cols(0, C) for b in XSn)
            if last and want_f32:
                out = new_f32(C)            # fp32 copy of the final features only
            if name == "self":
                if same:   # both images through one batched call
                    encoder_layer_split(w, XS[0], XS[0].cols(0, C), out[0], oxs[0], nhead, m01, m01, is_self=True)
                else:
                    for i, mk in ((1, mask0), (2, mask1)):
                        encoder_layer_split(w, XS[i], XS[i].cols(0, C), out[i], oxs[i], nhead, mk, mk, is_self=True)
            elif name == "cross":
                encoder_layer_split(w, XS[1], XS[2].cols(0, C), out[1], oxs[1], nhead, mask0, mask1)
                encoder_layer_split(w, XS[2], oxs[1], out[2], oxs[2], nhead, mask1, mask0)   # sees the updated feat0 (:96-97)
            else:
                raise KeyError(name)
            XS, XSn = XSn, XS
        self._feat_split = (fin[1], fin[2])
        return out[1], out[2]

    def coarse_features(self, image0, image1, mask0=None, mask1=None, want_f32=True):
        """Backbone + positional encoding + transformer -> (feat_c0 [N,L,C], feat_c1 [N,S,C], hw0_c, hw1_c); with ``want_f32=False``
        the features exist only as the split planes in ``self._feat_split`` (what ``forward`` correlates) and the first two are None."""
        P = self._packed or self._pack()
        bs = image0.size(0)
        same = image0.shape[2:] == image1.shape[2:]
        if same:
            c = self._backbone_hip(torch.cat([image0, image1], 0), P)
            c0, c1 = c[:bs], c[bs:]
        else:
            c0, c1 = self._backbone_hip(image0, P), self._backbone_hip(image1, P)
        hw0_c, hw1_c = tuple(c0.shape[1:3]), tuple(c1.shape[1:3])
        f0, f1 = self._transformer(c0.flatten(1, 2), c1.flatten(1, 2), P, self._pe_tokens(hw0_c),
                                   self._pe_tokens(hw1_c), mask0, mask1, want_f32)     # pos-enc added while splitting
        return f0, f1, hw0_c, hw1_c

    # -- "backbone once per image" (the reference re-runs the CNN for every pair an image is in)
    @torch.no_grad()
    @ops.first_call_range_sweep
    def image_tokens(self, images):
        """[B,1,H,W] -> (coarse backbone tokens [B, h*w, C] fp32, (h, w)).  Per-image results do not depend on what
        else is in the batch, so they can be cached and paired freely (``match_tokens``)."""
        P = self._packed or self._pack()
        c = self._backbone_hip(images, P)
        return c.flatten(1, 2), tuple(c.shape[1:3])

    @torch.no_grad()
    @ops.first_call_range_sweep
    def match_tokens(self, tok0, tok1, hw0_c, hw1_c, hw0_i, scale0=None, scale1=None, mask0=None, mask1=None, defer=False):
        """Positional encoding + transformer + coarse matching on cached backbone tokens of N pairs
        (tok0 [N,L,C], tok1 [N,S,C]; optional padding masks [N,h0c,w0c] / [N,h1c,w1c]); the same dict of matches
        ``forward`` leaves in ``data`` -- or, with ``defer=True``, an ``ops.PendingMatches`` whose ``result()`` is that dict
        (no host read inside the call: the scene loop launches the next batch first)."""
        P = self._packed or self._pack()
        self._feat_split = None
        m0, m1 = self._flat_masks(mask0, mask1, tok0.shape[0], tuple(hw0_c), tuple(hw1_c), tok0.device)
        self._transformer(tok0, tok1, P, self._pe_tokens(tuple(hw0_c)), self._pe_tokens(tuple(hw1_c)), m0, m1, want_f32=False)
        f0, f1 = self._feat_split
        self._feat_split = None
        mc = self.config["match_coarse"]
        return ops.coarse_match(f0, f1, tuple(hw0_c), tuple(hw1_c), mc["thr"], mc["border_rm"], mc["dsmax_temperature"],
                                scale0, scale1, hw0_i[0] / hw0_c[0], mask0=m0, mask1=m1, defer=defer)

    supports_defer = True      # plugin.match_scene_cached pipelines the batches of a scene through ``match_tokens(defer=True)``

    @staticmethod
    def _flat_masks(mask0, mask1, N, hw0_c, hw1_c, dev):
        """data['mask0'] / data['mask1'] ([N,h,w] at the coarse resolution, '0' = padded position; loftr.py:35-36, 61-63)
        as contiguous uint8 [N,L] / [N,S] on the device, or (None, None)."""
        if (mask0 is None) != (mask1 is None):
            raise ValueError("mask0 and mask1 come together (loftr.py:62-63 reads both)")
        if mask0 is None:
            return None, None
        if tuple(mask0.shape) != (N, *hw0_c) or tuple(mask1.shape) != (N, *hw1_c):
            raise ValueError(f"mask0 / mask1 must be [N,h,w] at the coarse resolution {hw0_c} / {hw1_c}, "
                             f"got {tuple(mask0.shape)} / {tuple(mask1.shape)}")
        return ((mask0.to(dev) != 0).to(torch.uint8).flatten(1).contiguous(),
                (mask1.to(dev) != 0).to(torch.uint8).flatten(1).contiguous())

    def _pe_tokens(self, hw):
        """Positional encoding in token-major layout [h*w, C] (cached per grid size)."""
        cache = self.__dict__.setdefault("_pe_cache", {})
        key = (hw, self.pe.device)
        if key not in cache:
            cache[key] = self.pe[0, :, :hw[0], :hw[1]].permute(1, 2, 0).reshape(hw[0] * hw[1], -1).contiguous()
        return cache[key]

    @torch.no_grad()
    @ops.first_call_range_sweep
    def forward(self, data: dict, defer: bool = False):
        """Updates ``data`` in place like LoFTR.forward (loftr.py:29-73, fine.enable=False) and returns None.
        ``defer=True`` (an extension for callers that loop over batches: ``plugin.match_scene_cached``, ``bench.py``'s pipelined
        rate) queues the whole forward WITHOUT the host read of the match count and returns ``finish``: calling it later -- after
        the next batch has been launched -- fills the match keys of ``data``."""
        img0, img1 = data["image0"], data["image1"]
        data.update({"bs": img0.size(0), "hw0_i": img0.shape[2:], "hw1_i": img1.shape[2:]})
        self._feat_split = None
        m0 = m1 = None
        if "mask0" in data:     # padded frames (loftr.py:61-65): masks through every attention, the dual-softmax, the border
            hw0, hw1 = (img0.shape[2] // 8, img0.shape[3] // 8), (img1.shape[2] // 8, img1.shape[3] // 8)
            m0, m1 = self._flat_masks(data["mask0"], data.get("mask1"), img0.size(0), hw0, hw1, img0.device)
        _, _, hw0_c, hw1_c = self.coarse_features(img0, img1, m0, m1, want_f32=False)
        f0, f1 = self._feat_split              # correlate the split planes the last LayerNorm wrote
        self._feat_split = None
        data.update({"hw0_c": torch.Size(hw0_c), "hw1_c": torch.Size(hw1_c),
                     "hw0_f": torch.Size((img0.shape[2] // 2, img0.shape[3] // 2)),
                     "hw1_f": torch.Size((img1.shape[2] // 2, img1.shape[3] // 2))})
        mc = self.config["match_coarse"]
        scale = data["hw0_i"][0] / hw0_c[0]
        pending = ops.coarse_match(f0, f1, hw0_c, hw1_c, mc["thr"], mc["border_rm"], mc["dsmax_temperature"],
                                   data.get("scale0"), data.get("scale1"), scale, mask0=m0, mask1=m1, defer=True)

        def finish():
            m = pending.result()
            data.update({"b_ids": m["b_ids"], "i_ids": m["i_ids"], "j_ids": m["j_ids"],
                         "gt_mask": m["mconf"] == 0, "m_bids": m["b_ids"], "mkpts0_c": m["mkpts0_c"],
                         "mkpts1_c": m["mkpts1_c"], "mconf": m["mconf"],
                         "mkpts0_f": m["mkpts0_c"], "mkpts1_f": m["mkpts1_c"]})
            return data
        if defer:
            return finish
        finish()
        return None
