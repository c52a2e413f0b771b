"""Parameter layouts of the two hot-path models, identical (names, shapes) to the reference's
``state_dict`` so real checkpoints load with ``strict=True``:

* LoFTR (211 tensors): third_party/LoFTR/src/loftr/loftr.py:12-27 and its sub-modules; the
  ``matcher.`` key prefix is stripped like loftr.py:83-87.
* MultiviewMatcher (72 tensors): src/MultiviewMatcher/MultiviewMatcher.py:17-42; checkpoint
  keys ``matcher.*`` -> strip, ``loftr_fine`` -> ``fine_transformer``, ``loftr_coarse`` dropped
  (src/post_optimization/matcher_model/multiview_match_worker.py:40-53).

Also: seeded synthetic weights (no checkpoints are available offline).  The distributions follow
the reference's initialisers (kaiming_normal fan_out for ResNet convs, xavier_uniform for
transformer matrices, torch defaults for VGG / adaptation convs) with randomised BatchNorm
running statistics so that eval-mode BN is exercised.  The generator is deterministic for a
given seed on any machine (torch CPU generator), so fixtures only need to store the seed.
"""
import math
from typing import Dict, List, Tuple

import torch
import torch.nn as nn

Spec = List[Tuple[str, Tuple[int, ...], str]]   # (name, shape, kind)


def _bn(spec: Spec, p: str, c: int):
    spec += [(p + "weight", (c,), "bn_w"), (p + "bias", (c,), "bn_b"),
             (p + "running_mean", (c,), "bn_mean"), (p + "running_var", (c,), "bn_var"),
             (p + "num_batches_tracked", (), "counter")]


def _encoder_layers(spec: Spec, p: str, n_layers: int, d: int):
    for i in range(n_layers):
        q = f"{p}layers.{i}."
        for nm in ("q_proj", "k_proj", "v_proj", "merge"):
            spec.append((q + nm + ".weight", (d, d), "xavier"))
        spec.append((q + "mlp.0.weight", (2 * d, 2 * d), "xavier"))
        spec.append((q + "mlp.2.weight", (d, 2 * d), "xavier"))
        for nm in ("norm1", "norm2"):
            spec += [(q + nm + ".weight", (d,), "ones"), (q + nm + ".bias", (d,), "zeros")]


def _resnet_fpn_spec(s: Spec, cfg: dict):
    """ResNetFPN_8_2 (backbone/resnet_fpn.py:43-98; the same file in third_party/LoFTR and third_party/aspantransformer)."""
    d0 = cfg["resnetfpn"]["initial_dim"]
    b1, b2, b3 = cfg["resnetfpn"]["block_dims"]
    p = "backbone."
    s.append((p + "conv1.weight", (d0, 1, 7, 7), "kaiming_out"))
    _bn(s, p + "bn1.", d0)
    cin = d0
    for li, (dim, stride) in enumerate(((b1, 1), (b2, 2), (b3, 2)), start=1):
        for bi in range(2):
            q = f"{p}layer{li}.{bi}."
            st = stride if bi == 0 else 1
            s.append((q + "conv1.weight", (dim, cin, 3, 3), "kaiming_out"))
            s.append((q + "conv2.weight", (dim, dim, 3, 3), "kaiming_out"))
            _bn(s, q + "bn1.", dim)
            _bn(s, q + "bn2.", dim)
            if st != 1:
                s.append((q + "downsample.0.weight", (dim, cin, 1, 1), "kaiming_out"))
                _bn(s, q + "downsample.1.", dim)
            cin = dim
    s.append((p + "layer3_outconv.weight", (b3, b3, 1, 1), "kaiming_out"))
    s.append((p + "layer2_outconv.weight", (b3, b2, 1, 1), "kaiming_out"))
    s.append((p + "layer2_outconv2.0.weight", (b3, b3, 3, 3), "kaiming_out"))
    _bn(s, p + "layer2_outconv2.1.", b3)
    s.append((p + "layer2_outconv2.3.weight", (b2, b3, 3, 3), "kaiming_out"))
    s.append((p + "layer1_outconv.weight", (b2, b1, 1, 1), "kaiming_out"))
    s.append((p + "layer1_outconv2.0.weight", (b2, b2, 3, 3), "kaiming_out"))
    _bn(s, p + "layer1_outconv2.1.", b2)
    s.append((p + "layer1_outconv2.3.weight", (b1, b2, 3, 3), "kaiming_out"))


def loftr_param_spec(cfg: dict) -> Spec:
    s: Spec = []
    _resnet_fpn_spec(s, cfg)
    dc, df = cfg["coarse"]["d_model"], cfg["fine"]["d_model"]
    _encoder_layers(s, "loftr_coarse.", len(cfg["coarse"]["layer_names"]), dc)
    # fine-level modules: present in every LoFTR checkpoint, unused when fine.enable=False
    s += [("fine_preprocess.down_proj.weight", (df, dc), "kaiming_out"),
          ("fine_preprocess.down_proj.bias", (df,), "zeros"),
          ("fine_preprocess.merge_feat.weight", (df, 2 * df), "kaiming_out"),
          ("fine_preprocess.merge_feat.bias", (df,), "zeros")]
    _encoder_layers(s, "loftr_fine.", len(cfg["fine"]["layer_names"]), df)
    return s


def aspanformer_param_spec(cfg: dict) -> Spec:
    """ASpanFormer (217 tensors): third_party/aspantransformer/src/ASpanFormer/aspanformer.py:14-29 and its sub-modules
    (aspan_module/transformer.py:8-32, 68-93, 138-149, 192-206; attention.py:21-39; utils/coarse_matching.py:72).  Transformer
    matrices are xavier_uniform (transformer.py:208-212), ``temp`` 1, the dual-softmax temperature 10, ``sample_offset`` the
    fixed 8x8 sampling pattern; the fine-level modules exist in every checkpoint and are unused in coarse_only mode."""
    s: Spec = []
    _resnet_fpn_spec(s, cfg)
    c = cfg["coarse"]
    d, df_, nl = c["d_model"], c["d_flow"], c["layer_num"]
    p = "loftr_coarse."
    s.append((p + "pos_transform.weight", (df_, d, 1, 1), "xavier"))
    for i in range(c["ini_layer_num"]):
        q = f"{p}ini_layer.layers_coarse.{i}."
        s += [(q + "q_proj.weight", (d, d, 1), "xavier"), (q + "k_proj.weight", (d, d, 1), "xavier"),
              (q + "v_proj.weight", (d, d + df_, 1), "xavier"), (q + "merge_head.weight", (d, d, 1), "xavier"),
              (q + "merge_f.0.weight", (2 * d, 2 * d, 1, 1), "xavier"), (q + "merge_f.2.weight", (d, 2 * d, 1, 1), "xavier"),
              (q + "norm1.affine", (d,), "ones"), (q + "norm1.bias", (d,), "zeros"),
              (q + "norm2.affine", (d,), "ones"), (q + "norm2.bias", (d,), "zeros")]
    s += [(p + "ini_layer.decoupler.weight", (d + df_, d, 1, 1), "xavier"),
          (p + "ini_layer.decoupler.bias", (d + df_,), "torch_conv_b:%d" % d),
          (p + "ini_layer.up_merge.weight", (d, 2 * d, 1, 1), "xavier"),
          (p + "ini_layer.up_merge.bias", (d,), "torch_conv_b:%d" % (2 * d))]
    for i in range(nl):
        q = f"{p}layers.{i}."
        extra = df_ if i < nl - 1 else 0                    # the last layer does not update the flow feature
        s += [(q + "flow_decoder.0.weight", (df_ // 2, df_, 1), "xavier"), (q + "flow_decoder.2.weight", (4, df_ // 2, 1), "xavier"),
              (q + "attention.temp", (), "const:1.0"), (q + "attention.sample_offset", (c["nsample"][1] ** 2, 2), "sample_offset"),
              (q + "attention.merge_head.0.weight", (d, 3 * d, 1), "xavier"), (q + "attention.merge_head.2.weight", (d, d, 1), "xavier"),
              (q + "q_proj.weight", (d, d, 1), "xavier"), (q + "k_proj.weight", (d, d, 1), "xavier"),
              (q + "v_proj.weight", (d, d + df_, 1), "xavier"),
              (q + "merge_f.0.weight", (d + df_, 2 * d + extra, 1, 1), "xavier"),
              (q + "merge_f.2.weight", (d + extra, d + df_, 3, 3), "xavier"),
              (q + "norm1.affine", (d,), "ones"), (q + "norm1.bias", (d,), "zeros"),
              (q + "norm2.affine", (d + extra,), "ones"), (q + "norm2.bias", (d + extra,), "zeros")]
    s.append(("coarse_matching.temperature", (), "const:10.0"))
    dfine = cfg["fine"]["d_model"]
    s += [("fine_preprocess.down_proj.weight", (dfine, d), "kaiming_out"), ("fine_preprocess.down_proj.bias", (dfine,), "zeros"),
          ("fine_preprocess.merge_feat.weight", (dfine, 2 * dfine), "kaiming_out"), ("fine_preprocess.merge_feat.bias", (dfine,), "zeros")]
    _encoder_layers(s, "loftr_fine.", len(cfg["fine"]["layer_names"]), dfine)
    return s


_VGG_CONVS = ((0, 3, 64), (2, 64, 64), (5, 64, 128), (7, 128, 128), (10, 128, 256), (12, 256, 256), (14, 256, 256))
_ADAP_IN = (64, 256)   # channels of relu1_2 / relu3_3 (vgg16_layers, backbone/S2DNet/vggnet.py:12-44)


def multiview_param_spec(cfg: dict) -> Spec:
    s: Spec = []
    od = cfg["backbone"]["s2dnet"]["output_dim"]
    for idx, cin, cout in _VGG_CONVS:
        s += [(f"backbone.encoder.{idx}.weight", (cout, cin, 3, 3), "torch_conv_w"),
              (f"backbone.encoder.{idx}.bias", (cout,), "torch_conv_b:%d" % (cin * 9))]
    for i in range(cfg["backbone"]["s2dnet"]["num_layers"]):
        q = f"backbone.adaptation_layers.adap_layer_{i}."
        s += [(q + "0.weight", (64, _ADAP_IN[i], 1, 1), "torch_conv_w"),
              (q + "0.bias", (64,), "torch_conv_b:%d" % _ADAP_IN[i]),
              (q + "2.weight", (od, 64, 5, 5), "torch_conv_w"),
              (q + "2.bias", (od,), "torch_conv_b:%d" % (64 * 25))]
        _bn(s, q + "3.", od)
    mt = cfg["multiview_transform"]
    _encoder_layers(s, "fine_transformer.", len(mt["layer_names"]) * mt["layer_iter_n"], mt["d_model"])
    return s


MF_EMBED = (128, 192, 256, 512)        # Matchformer_LA_large: embed_dims, 3 blocks per stage, mlp_ratio 4, 8 heads
MF_PATCH = (7, 3, 3, 3)


def matchformer_param_spec() -> Spec:
    """Matchformer (backbone 'largela') -- third_party/MatchFormer/model/matchformer.py:10-19 and
    model/backbone/match_LA_large.py:118-218: 229 tensors, in the reference's ``state_dict`` order."""
    s: Spec = []
    cin = 1
    for st, C in enumerate(MF_EMBED):
        p = f"backbone.AttentionBlock{st + 1}."
        k = MF_PATCH[st]
        s += [(p + "patch_embed.proj.weight", (C, cin, k, k), "conv_fan_out:1"), (p + "patch_embed.proj.bias", (C,), "bn_b"),
              (p + "patch_embed.pos.pa_conv.weight", (C, 1, 3, 3), f"conv_fan_out:{C}"),
              (p + "patch_embed.pos.pa_conv.bias", (C,), "bn_b"),
              (p + "patch_embed.norm.weight", (C,), "bn_w"), (p + "patch_embed.norm.bias", (C,), "bn_b")]
        for i in range(3):
            q = f"{p}block.{i}."
            s += [(q + "norm1.weight", (C,), "bn_w"), (q + "norm1.bias", (C,), "bn_b"),
                  (q + "attn.q.weight", (C, C), "tn02"), (q + "attn.q.bias", (C,), "bn_b"),
                  (q + "attn.kv.weight", (2 * C, C), "tn02"), (q + "attn.kv.bias", (2 * C,), "bn_b"),
                  (q + "norm.weight", (C,), "bn_w"), (q + "norm.bias", (C,), "bn_b"),
                  (q + "mlp.fc1.weight", (4 * C, C), "tn02"), (q + "mlp.fc1.bias", (4 * C,), "bn_b"),
                  (q + "mlp.dwconv.dwconv.weight", (4 * C, 1, 3, 3), f"conv_fan_out:{4 * C}"),
                  (q + "mlp.dwconv.dwconv.bias", (4 * C,), "bn_b"),
                  (q + "mlp.fc2.weight", (C, 4 * C), "tn02"), (q + "mlp.fc2.bias", (C,), "bn_b")]
        s += [(p + "norm.weight", (C,), "bn_w"), (p + "norm.bias", (C,), "bn_b")]
        cin = C
    e = MF_EMBED
    p = "backbone."
    s.append((p + "layer4_outconv.weight", (e[3], e[3], 1, 1), "conv_fan_out:1"))
    s.append((p + "layer3_outconv.weight", (e[3], e[2], 1, 1), "conv_fan_out:1"))
    def outconv2(nm, a, b):
        s.append((f"{p}{nm}.0.weight", (a, a, 3, 3), "conv_fan_out:1"))
        _bn(s, f"{p}{nm}.1.", a)
        s.append((f"{p}{nm}.3.weight", (b, a, 3, 3), "conv_fan_out:1"))
    outconv2("layer3_outconv2", e[3], e[2])
    s.append((p + "layer2_outconv.weight", (e[2], e[1], 1, 1), "conv_fan_out:1"))
    outconv2("layer2_outconv2", e[2], e[1])
    s.append((p + "layer1_outconv.weight", (e[1], e[0], 1, 1), "conv_fan_out:1"))
    outconv2("layer1_outconv2", e[1], e[0])
    # fine-level modules: present in every checkpoint, unused when fine.enable=False (matchformer.py:17-18)
    s += [("fine_preprocess.down_proj.weight", (128, 256), "kaiming_out"), ("fine_preprocess.down_proj.bias", (128,), "zeros"),
          ("fine_preprocess.merge_feat.weight", (128, 256), "kaiming_out"), ("fine_preprocess.merge_feat.bias", (128,), "zeros")]
    return s


def random_state_dict(spec: Spec, seed: int = 0) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name, shape, kind in spec:
        if kind == "kaiming_out":          # nn.init.kaiming_normal_(mode='fan_out', relu)
            fan_out = shape[0] * (shape[2] * shape[3] if len(shape) == 4 else 1)
            t = torch.randn(shape, generator=g) * math.sqrt(2.0 / fan_out)
        elif kind == "xavier":             # nn.init.xavier_uniform_ (fan = channels * receptive field)
            rf = 1
            for k in shape[2:]:
                rf *= k
            bound = math.sqrt(6.0 / ((shape[0] + shape[1]) * rf))
            t = (torch.rand(shape, generator=g) * 2 - 1) * bound
        elif kind.startswith("const:"):
            t = torch.full(shape, float(kind.split(":")[1]))
        elif kind == "sample_offset":      # HierachicalAttention.__init__ (aspan_module/attention.py:38-39)
            n = int(round(math.sqrt(shape[0])))
            t = torch.tensor([[a - n / 2 + 0.5, b - n / 2 + 0.5] for a in range(n) for b in range(n)])
        elif kind == "tn02":               # timm trunc_normal_(std=.02) (match_LA_large.py:208-211)
            t = (torch.randn(shape, generator=g) * 0.02).clamp(-2.0, 2.0)
        elif kind.startswith("conv_fan_out"):   # normal_(0, sqrt(2 / (kh*kw*out / groups))) (:215-218)
            groups = int(kind.split(":")[1])
            t = torch.randn(shape, generator=g) * math.sqrt(2.0 / (shape[0] * shape[2] * shape[3] / groups))
        elif kind == "torch_conv_w":       # kaiming_uniform_(a=sqrt(5)) -> U(-1/sqrt(fan_in), ..)
            bound = 1.0 / math.sqrt(shape[1] * shape[2] * shape[3])
            t = (torch.rand(shape, generator=g) * 2 - 1) * bound
        elif kind.startswith("torch_conv_b"):
            bound = 1.0 / math.sqrt(int(kind.split(":")[1]))
            t = (torch.rand(shape, generator=g) * 2 - 1) * bound
        elif kind in ("ones",):
            t = torch.ones(shape)
        elif kind in ("zeros",):
            t = torch.zeros(shape)
        elif kind == "bn_w":
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif kind == "bn_b":
            t = 0.1 * torch.randn(shape, generator=g)
        elif kind == "bn_mean":
            t = 0.1 * torch.randn(shape, generator=g)
        elif kind == "bn_var":
            t = torch.rand(shape, generator=g) + 0.5
        elif kind == "counter":
            t = torch.zeros((), dtype=torch.long)
        else:
            raise ValueError(kind)
        sd[name] = t
    return sd


def planted_loftr_state_dict(spec: Spec, seed: int = 0, alpha: float = 3.0) -> Dict[str, torch.Tensor]:
    """``random_state_dict`` with ONE tensor re-scaled so that synthetic frames produce real matches.

    Seeded random weights make the coarse features a large position-independent vector plus a small content
    term, so the dual-softmax is flat (0 matches at thr 0.2).  Here ``backbone.layer3_outconv``
    becomes ``alpha * W (I - mu mu^T / |mu|^2)``: mu is the position mean of that layer's input for the seeded
    weights (a committed 256-vector, data/planted_mu_seed<seed>.npy, measured once by oracle/make_planted.py),
    so the common vector is projected out and the content term amplified.  On ``synth.coarse_pair_batch``
    frames (image1 = image0 rolled by a whole number of coarse cells) ~3600 of 4800 cells then match at
    thr 0.2 with confidences spread over (0.2, 1] -- the tables the benchmarks and end-to-end parity tests use.
    Everything else (BN statistics, the transformer) stays the seeded random tensor."""
    import os
    import numpy as np
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", f"planted_mu_seed{seed}.npy")
    if not os.path.exists(path):
        raise FileNotFoundError(f"{path}: run `python -m oracle.make_planted` for this seed")
    sd = random_state_dict(spec, seed)
    mu = torch.from_numpy(np.load(path)).double()
    W = sd["backbone.layer3_outconv.weight"][:, :, 0, 0].double()
    Wn = alpha * (W - torch.outer(W @ mu, mu) / (mu @ mu))
    sd["backbone.layer3_outconv.weight"] = Wn.float()[:, :, None, None].contiguous()
    return sd


def planted_matchformer_state_dict(spec: Spec, seed: int = 0, alpha: float = 3.0) -> Dict[str, torch.Tensor]:
    """The MatchFormer counterpart of ``planted_loftr_state_dict``: the last FPN convolution that produces the coarse
    features (``backbone.layer3_outconv2.3``, 3x3, no bias) gets every tap projected off the position mean mu of its
    input, ``alpha * W_t (I - mu mu^T / |mu|^2)`` (mu: 512 committed doubles, oracle/make_planted.py), so that synthetic
    frames give real matches at thr 0.2."""
    import os
    import numpy as np
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", f"planted_mu_matchformer_seed{seed}.npy")
    if not os.path.exists(path):
        raise FileNotFoundError(f"{path}: run `python -m oracle.make_planted matchformer` for this seed")
    sd = random_state_dict(spec, seed)
    mu = torch.from_numpy(np.load(path)).double()
    W = sd["backbone.layer3_outconv2.3.weight"].double()
    proj = torch.einsum("oikl,i->okl", W, mu)
    Wn = alpha * (W - proj[:, None] * mu[None, :, None, None] / (mu @ mu))
    sd["backbone.layer3_outconv2.3.weight"] = Wn.float().contiguous()
    return sd



def planted_aspanformer_state_dict(spec: Spec, seed: int = 0, alpha: float = 3.0) -> Dict[str, torch.Tensor]:
    """Seeded ASpanFormer weights whose backbone is the planted LoFTR backbone (the two specs start with the same 107
    backbone entries, so the seeded draws -- and the committed position mean -- are the same): on ``synth`` frames the
    xavier-initialised span transformer then keeps enough content for real matches at thr 0.2."""
    sd = random_state_dict(spec, seed)
    n_bb = sum(1 for name, _, _ in spec if name.startswith("backbone."))
    planted = planted_loftr_state_dict([e for e in spec[:n_bb]], seed, alpha)
    for k, v in planted.items():
        sd[k] = v
    return sd

def planted_multiview_state_dict(spec: Spec, seed: int = 1, alpha: float = 32.0) -> Dict[str, torch.Tensor]:
    """``random_state_dict`` of the refinement head with PEAKED fine heat-maps.

    With seeded weights the two adaptation layers of S2DNet emit a large per-channel constant plus a small content term (mean /
    std ~ 6), every window feature correlates equally with every other, the 15 x 15 heat-maps are flat and the 49 candidate
    scores of a track differ by ~1e-5 -- the argmin of fine_matching.py:129-179 is then decided by rounding, which exercises the
    tie rule, not the expectation arithmetic.  Here the BatchNorm affine that ends each adaptation layer becomes
    ``alpha * (BN(x) - mu)``: mu = that layer's per-channel output mean for the seeded weights (committed, [2, 128] doubles,
    data/planted_mu_refine_seed<seed>.npy, measured once by ``python -m oracle.make_planted refine``), so the constant is removed
    and the content amplified.  On ``synth.refine_bag`` bags the candidate scores then spread over (0.36, 1.29) with the best and
    the second-best candidate of every track >= 1.7e-3 apart.  Everything else stays the seeded tensor."""
    import os
    import numpy as np
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", f"planted_mu_refine_seed{seed}.npy")
    if not os.path.exists(path):
        raise FileNotFoundError(f"{path}: run `python -m oracle.make_planted refine` for this seed")
    sd = random_state_dict(spec, seed)
    mu = torch.from_numpy(np.load(path)).double()
    for i in (0, 1):
        q = f"backbone.adaptation_layers.adap_layer_{i}.3."
        w, b = sd[q + "weight"].double(), sd[q + "bias"].double()
        sd[q + "weight"] = (alpha * w).float()
        sd[q + "bias"] = (alpha * (b - mu[i])).float()
    return sd


class ParamModule(nn.Module):
    """nn.Module whose parameters/buffers carry the reference's dotted names.

    ``register_spec`` builds the nested container modules so that ``state_dict()`` /
    ``load_state_dict(strict=True)`` / ``.cuda()`` behave exactly like the reference model's.
    ``p(name)`` fetches a tensor by its dotted name."""

    def register_spec(self, spec: Spec):
        self._names = []
        for name, shape, kind in spec:
            parts = name.split(".")
            mod = self
            for part in parts[:-1]:
                if not hasattr(mod, part):
                    mod.add_module(part, nn.Module())
                mod = getattr(mod, part)
            if kind in ("bn_mean", "bn_var", "counter"):
                init = torch.zeros(shape, dtype=torch.long if kind == "counter" else torch.float32)
                if kind == "bn_var":
                    init = torch.ones(shape)
                mod.register_buffer(parts[-1], init)
            else:
                init = torch.ones(shape) if kind in ("ones", "bn_w") else torch.zeros(shape)
                if kind.startswith("const:") or kind == "sample_offset":      # constants the reference's constructors set
                    init = random_state_dict([(name, shape, kind)], 0)[name]
                mod.register_parameter(parts[-1], nn.Parameter(init, requires_grad=False))
            self._names.append(name)

    def load_state_dict(self, state_dict, *args, **kwargs):
        """New weights: the next call of every entry point runs the split-plane range sweep again (ops.range_sweep)."""
        self.__dict__["_range_done"] = set()
        return super().load_state_dict(state_dict, *args, **kwargs)

    def p(self, name: str) -> torch.Tensor:
        obj = self
        for part in name.split("."):
            obj = getattr(obj, part)
        return obj
