"""TEST INFRASTRUCTURE ONLY -- calibration constant of the "planted" synthetic LoFTR weights.

``python -m oracle.make_planted`` -> detectorfreesfm_amd/data/planted_mu_seed0.npy

Seeded random LoFTR weights give a flat confidence matrix (0 matches at thr 0.2, SURVEY.md 8c "test-input
caveat"): the coarse features are a large position-independent vector plus a small content term.
``params.planted_loftr_state_dict`` removes that common vector analytically from the ONE layer that has no
successor bias -- ``backbone.layer3_outconv.weight`` W -> alpha * W (I - mu mu^T / |mu|^2) -- where mu is the
mean of the layer's input over positions.  mu depends only on the seed-0 weights; it is measured once here,
through the oracle's backbone (restate.resnet_fpn_8_2, i.e. the reference's ResNetFPN_8_2 arithmetic) on the
first config-2 frame, and committed (256 doubles), so the weights are bit-identical on every machine."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import restate  # noqa: E402
from detectorfreesfm_amd import synth  # noqa: E402
from detectorfreesfm_amd.config import loftr_coarse_only_config  # noqa: E402
from detectorfreesfm_amd.params import loftr_param_spec, random_state_dict  # noqa: E402


def main(seed=0):
    cfg = loftr_coarse_only_config(0.2)
    sd = random_state_dict(loftr_param_spec(cfg), seed)
    img = synth.coarse_pair_batch(1, 480, 640, seed=1000)["image0"]
    with torch.no_grad():
        c, _ = restate.resnet_fpn_8_2(sd, "backbone.", img, False)
    W = sd["backbone.layer3_outconv.weight"][:, :, 0, 0].double()
    mu = torch.linalg.solve(W, c.double().mean((0, 2, 3)))
    out = os.path.join(ROOT, "detectorfreesfm_amd", "data", f"planted_mu_seed{seed}.npy")
    np.save(out, mu.numpy())
    print(out, mu.shape, float(mu.mean()))


def main_matchformer(seed=0):
    """mu of ``params.planted_matchformer_state_dict``: position mean of the input of backbone.layer3_outconv2.3 (the
    LeakyReLU output of the FPN's level-3 block) for the seeded MatchFormer weights, on the first 240x320 config-2 pair
    through the oracle's backbone (restate_matchformer)."""
    import torch.nn.functional as F
    from oracle import restate_matchformer as rm
    from detectorfreesfm_amd.params import matchformer_param_spec
    sd = rm.as_params(random_state_dict(matchformer_param_spec(), seed))
    pair = synth.coarse_pair_batch(1, 240, 320, seed=1000)
    x = torch.cat([pair["image0"], pair["image1"]], 0)
    p = "backbone."
    with torch.no_grad():
        outs = []
        for s in range(4):
            x = rm.attention_block(sd, f"{p}AttentionBlock{s + 1}.", x, s)
            outs.append(x)
        c4 = F.conv2d(outs[3], sd[p + "layer4_outconv.weight"])
        t = F.conv2d(outs[2], sd[p + "layer3_outconv.weight"]) + F.interpolate(c4, size=outs[2].shape[2:], mode="bilinear", align_corners=True)
        q = p + "layer3_outconv2."
        t = F.conv2d(t, sd[q + "0.weight"], None, 1, 1)
        t = F.batch_norm(t, sd[q + "1.running_mean"], sd[q + "1.running_var"], sd[q + "1.weight"], sd[q + "1.bias"], False, 0.0, 1e-5)
        mu = F.leaky_relu(t, 0.01).double().mean((0, 2, 3))
    out = os.path.join(ROOT, "detectorfreesfm_amd", "data", f"planted_mu_matchformer_seed{seed}.npy")
    np.save(out, mu.numpy())
    print(out, mu.shape, float(mu.mean()))


def main_refine(seed=1):
    """mu of ``params.planted_multiview_state_dict``: per-channel means of the two adaptation-layer outputs of S2DNet (after their
    BatchNorm, s2dnet.py:24-52) for the seeded refinement weights, over the patches of a 24-track x 5-view config-3 bag through the
    oracle's arithmetic (restate.s2dnet_forward's layers).  [2, 128] doubles."""
    import torch.nn.functional as F
    from detectorfreesfm_amd.config import multiview_refinement_config
    from detectorfreesfm_amd.params import multiview_param_spec
    cfg = multiview_refinement_config()
    sd = random_state_dict(multiview_param_spec(cfg), seed)
    data = synth.refine_bag(24, 5, 480, 640, seed=2000)
    p = "backbone."
    pts = torch.cat([data["query_points"][:, None], data["reference_points_coarse"]], 1)[0]            # [V, T, 2]
    with torch.no_grad():
        patches = torch.cat([restate.extract_local_patches(data["images"][v], pts[v], 35) for v in range(5)], 0)
        mean = patches.new_tensor(restate.IMAGENET_MEAN)[:, None, None]
        std = patches.new_tensor(restate.IMAGENET_STD)[:, None, None]
        x = (patches - mean) / std
        conv = lambda i, t: F.relu(F.conv2d(t, sd[f"{p}encoder.{i}.weight"], sd[f"{p}encoder.{i}.bias"], 1, 1))
        f0 = conv(2, conv(0, x))
        x = conv(7, conv(5, F.max_pool2d(f0, 3, 2, 1)))
        f1 = conv(14, conv(12, conv(10, F.max_pool2d(x, 3, 2, 1))))
        mus = []
        for i, t in ((0, f0), (1, f1)):
            q = f"{p}adaptation_layers.adap_layer_{i}."
            t = F.relu(F.conv2d(t, sd[q + "0.weight"], sd[q + "0.bias"]))
            t = restate._bn(sd, q + "3.", F.conv2d(t, sd[q + "2.weight"], sd[q + "2.bias"], 1, 2))
            mus.append(t.double().mean((0, 2, 3)))
    out = os.path.join(ROOT, "detectorfreesfm_amd", "data", f"planted_mu_refine_seed{seed}.npy")
    np.save(out, torch.stack(mus).numpy())
    print(out, torch.stack(mus).shape, float(torch.stack(mus).abs().mean()))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "matchformer":
        main_matchformer()
    elif len(sys.argv) > 1 and sys.argv[1] == "refine":
        main_refine()
    else:
        main()
