"""End-to-end parity of the two plugins on a real MI355X: against fixtures produced by the real reference, and
against the oracle on fresh seeded inputs.

Assertions are the north_star's: match indices identical, confidences / refined coordinates within 1e-4.  The only
exemptions are per entry and analysed against the oracle's own dense confidence matrix / candidate scores
(tests/parity.py): a confidence within 1e-5 of the threshold, a mutual-max tie within 1e-6, an argmin between
candidate scores closer than 1e-5.  Every exempted entry is listed; none is waved through as a percentage."""
import numpy as np
import pytest
import torch

import parity
from detectorfreesfm_amd import HipLoFTR, HipMultiviewMatcher, ops, plugin, synth
from detectorfreesfm_amd.config import loftr_coarse_only_config, multiview_refinement_config
from detectorfreesfm_amd.params import (loftr_param_spec, multiview_param_spec, planted_loftr_state_dict,
                                        random_state_dict)
from oracle import restate

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
MATCH_KEYS = ("b_ids", "i_ids", "j_ids", "mconf", "mkpts0_f", "mkpts1_f")


def _case(npz):
    return {k: npz[k].item() for k in npz.files if npz[k].ndim == 0}


def _loftr(thr, planted=True, seed=0):
    cfg = loftr_coarse_only_config(thr)
    spec = loftr_param_spec(cfg)
    sd = planted_loftr_state_dict(spec, seed) if planted else random_state_dict(spec, seed)
    m = HipLoFTR(cfg)
    m.load_state_dict(sd, strict=True)
    return cfg, sd, m.eval().to(DEV)


def _refiner(seed=1):
    cfg = multiview_refinement_config()
    sd = random_state_dict(multiview_param_spec(cfg), seed)
    m = HipMultiviewMatcher(cfg, test=True)
    m.load_state_dict(sd, strict=True)
    return cfg, sd, m.eval().to(DEV)


def _oracle_coarse(sd, cfg, data):
    """Oracle tables + its dense confidence matrix (for the per-entry exemption analysis)."""
    with torch.no_grad():
        o = restate.loftr_coarse_forward(sd, cfg, data, with_fine_backbone=False)
        m0 = data["mask0"].flatten(-2) if "mask0" in data else None
        m1 = data["mask1"].flatten(-2) if "mask1" in data else None
        conf = restate.dual_softmax_conf(o["feat_c0"], o["feat_c1"], cfg["match_coarse"]["dsmax_temperature"], m0, m1)
    return o, conf


def _strict_coarse(d, ref, conf, thr, label, exact=None):
    ex = parity.check_coarse(d, ref, conf, thr, exact=exact)
    parity.check_coarse_rows(d, ref, [e for e in ex if e[0] != "oracle-noise"])      # "oracle-noise" rows are in both tables
    print(f"[{label}] {len(ref['i_ids'])} reference matches, {len(d['i_ids'])} on the GPU, exempted entries: {ex[:12]}"
          f"{' ...' if len(ex) > 12 else ''} ({len(ex)} in total)")
    return ex


def _same_tables(a, b, thr):
    """HIP vs HIP (different batch composition): same rows; a row may be absent from one side only if its
    confidence is within parity.TOL_THR of thr."""
    A = {(int(x), int(y)): (int(j), float(c)) for x, y, j, c in zip(a["b_ids"], a["i_ids"], a["j_ids"], a["mconf"])}
    B = {(int(x), int(y)): (int(j), float(c)) for x, y, j, c in zip(b["b_ids"], b["i_ids"], b["j_ids"], b["mconf"])}
    bad = [(k, A.get(k), B.get(k)) for k in set(A) ^ set(B) if abs((A.get(k) or B.get(k))[1] - thr) > parity.TOL_THR]
    bad += [(k, A[k], B[k]) for k in set(A) & set(B) if A[k][0] != B[k][0] or abs(A[k][1] - B[k][1]) > parity.TOL_CONF]
    assert not bad, bad[:5]
    return len(set(A) & set(B))


# ---------------------------------------------------------------- coarse matcher
def test_loftr_e2e_golden(built_lib, golden):
    """Seeded random weights, thr 1e-3 (flat confidences: the hardest case for index parity)."""
    gz = golden("loftr_e2e")
    c = _case(gz)
    cfg, sd, m = _loftr(c["thr"], planted=False, seed=c["weight_seed"])
    data = synth.coarse_pair_batch(c["n_pairs"], c["H"], c["W"], c["data_seed"])
    data["scale0"], data["scale1"] = torch.from_numpy(gz["scale0"]), torch.from_numpy(gz["scale1"])
    d = synth.to_device(data, DEV)
    m(d)
    _, conf = _oracle_coarse(sd, cfg, data)
    ex = _strict_coarse(d, {k: gz[k] for k in MATCH_KEYS}, conf, c["thr"], "loftr_e2e")
    assert len(ex) <= 2 and (d["m_bids"] == 0).all()


def test_loftr_e2e_planted_golden(built_lib, golden):
    """Planted weights, production threshold 0.2, 2 pairs with different scales: fixture from the real reference."""
    gz = golden("loftr_e2e_planted")
    c = _case(gz)
    cfg, sd, m = _loftr(c["thr"], seed=c["weight_seed"])
    data = synth.coarse_pair_batch(c["n_pairs"], c["H"], c["W"], c["data_seed"])
    data["scale0"], data["scale1"] = torch.from_numpy(gz["scale0"]), torch.from_numpy(gz["scale1"])
    d = synth.to_device(data, DEV)
    m(d)
    _, conf = _oracle_coarse(sd, cfg, data)
    ex = _strict_coarse(d, {k: gz[k] for k in MATCH_KEYS}, conf, c["thr"], "loftr_e2e_planted")
    assert len(ex) == 0 and len(d["i_ids"]) == len(gz["i_ids"]) > 100
    assert np.array_equal(d["i_ids"].cpu().numpy(), gz["i_ids"]) and np.array_equal(d["j_ids"].cpu().numpy(), gz["j_ids"])


def test_loftr_e2e_two_sizes_golden(built_lib, golden):
    """Frames of different size: two backbone calls, L != S (loftr.py:45-49)."""
    gz = golden("loftr_e2e_two_sizes")
    c = _case(gz)
    cfg, sd, m = _loftr(c["thr"], seed=c["weight_seed"])
    data = synth.coarse_pair_two_sizes(c["H0"], c["W0"], c["H1"], c["W1"], c["data_seed"])
    data["scale0"], data["scale1"] = torch.from_numpy(gz["scale0"]), torch.from_numpy(gz["scale1"])
    d = synth.to_device(data, DEV)
    m(d)
    assert tuple(d["hw0_c"]) == (12, 16) and tuple(d["hw1_c"]) == (10, 14)
    _, conf = _oracle_coarse(sd, cfg, data)
    ex = _strict_coarse(d, {k: gz[k] for k in MATCH_KEYS}, conf, c["thr"], "loftr_e2e_two_sizes")
    assert len(ex) == 0 and len(gz["i_ids"]) > 30


def test_loftr_e2e_masked_golden(built_lib, golden):
    """Padded frames with mask0 / mask1 (loftr.py:61-65): fixture from the real LoFTR module -- masks in every linear
    attention, the dual-softmax and mask_border_with_padding; both pairs of the batch share the 2N-sequence self layers."""
    gz = golden("loftr_e2e_masked")
    c = _case(gz)
    cfg, sd, m = _loftr(c["thr"], seed=c["weight_seed"])
    data = synth.coarse_pair_padded(c["n_pairs"], c["H"], c["W"], c["data_seed"])
    data["scale0"], data["scale1"] = torch.from_numpy(gz["scale0"]), torch.from_numpy(gz["scale1"])
    d = synth.to_device(data, DEV)
    m(d)
    _, conf = _oracle_coarse(sd, cfg, data)
    ex = _strict_coarse(d, {k: gz[k] for k in MATCH_KEYS}, conf, c["thr"], "loftr_e2e_masked")
    assert len(ex) == 0 and len(d["i_ids"]) == len(gz["i_ids"]) > 60
    assert np.array_equal(d["i_ids"].cpu().numpy(), gz["i_ids"]) and np.array_equal(d["j_ids"].cpu().numpy(), gz["j_ids"])
    # the cached-token scene entry point takes the same masks
    t0, hw = m.image_tokens(d["image0"])
    t1, _ = m.image_tokens(d["image1"])
    mt = m.match_tokens(t0, t1, hw, hw, (c["H"], c["W"]), d["scale0"], d["scale1"], mask0=d["mask0"], mask1=d["mask1"])
    assert torch.equal(mt["i_ids"], d["i_ids"]) and torch.equal(mt["j_ids"], d["j_ids"])


def test_loftr_masked_640x480_and_two_sizes_vs_oracle(built_lib):
    """Masks at BASELINE configs[1]'s frame size (a 640x480 canvas holding a 600x440 and a 560x480 frame) and on frames of
    two sizes (per-image self layers, L != S): rows identical to the oracle's."""
    cfg, sd, m = _loftr(0.2)
    data = synth.coarse_pair_padded(1, 480, 640, seed=1100, valid=[((55, 75), (60, 70))])
    d = synth.to_device(data, DEV)
    m(d)
    o, conf = _oracle_coarse(sd, cfg, data)
    assert o["i_ids"].numel() > 2000
    ex = _strict_coarse(d, o, conf, 0.2, "640x480 masked")
    assert len(ex) <= 3
    two = synth.coarse_pair_two_sizes(96, 128, 80, 112, 1000)
    two["mask0"] = torch.ones((1, 12, 16), dtype=torch.bool)
    two["mask1"] = torch.ones((1, 10, 14), dtype=torch.bool)
    two["mask0"][0, 10:] = False
    two["mask1"][0, :, 11:] = False
    d2 = synth.to_device(two, DEV)
    m(d2)
    o2, conf2 = _oracle_coarse(sd, cfg, two)
    assert o2["i_ids"].numel() > 20
    assert len(_strict_coarse(d2, o2, conf2, 0.2, "two sizes masked")) == 0


@pytest.mark.parametrize("H,W", [(800, 1200)])     # 1600x1064: test_loftr_benched_batch2_1600x1064_vs_oracle (two pairs, same rules; r05)
def test_loftr_production_frame_sizes_vs_oracle(built_lib, H, W):
    """The frame sizes the reference actually feeds the matcher: every shipped config resizes to 1200 or 1600 px
    (src/coarse_match/coarse_match.py:15, hydra_configs/eth3d_sfm/dfsfm.yaml:76, hydra_configs/demo/dfsfm.yaml:48) ->
    coarse grids 150x100 / 200x133, L = 15 000 / 26 600 (the positional-encoding buffer allows 256x256,
    position_encoding.py:11-22).  One planted pair end to end against the oracle under the north_star rules: K1 with
    S = 26 600 per image, K3 with 208 column tiles and its full candidate workspace, cm_compact over 26 600 rows."""
    cfg, sd, m = _loftr(0.2)
    data = synth.coarse_pair_batch(1, H, W, seed=1300)
    d = synth.to_device(data, DEV)
    m(d)
    assert tuple(d["hw0_c"]) == (H // 8, W // 8)
    o, conf = _oracle_coarse(sd, cfg, data)
    assert o["i_ids"].numel() > 0.5 * (H // 8) * (W // 8)
    # rule "oracle-noise" (tests/parity.py): at these grid sizes ATen's fp32 softmax is itself up to 1.15e-4 from the exact
    # confidence of its own features; entries beyond 1e-4 of the fp32 oracle must be within 1e-4 of that exact value
    ex = _strict_coarse(d, o, conf, 0.2, f"{W}x{H} planted", exact=(o["feat_c0"], o["feat_c1"], cfg["match_coarse"]["dsmax_temperature"]))
    sums = [e for e in ex if e[0] == "oracle-noise"]
    assert len(ex) - len(sums) <= 3 and len(sums) <= 0.005 * o["i_ids"].numel()


def test_loftr_640x480_planted_vs_oracle(built_lib):
    """BASELINE configs[1] frame size, production threshold, ~3600 matches per pair: every row identical to the
    oracle's (indices, pixel coordinates), confidences within 1e-4."""
    cfg, sd, m = _loftr(0.2)
    data = synth.coarse_pair_batch(1, 480, 640, seed=1100)
    d = synth.to_device(data, DEV)
    m(d)
    o, conf = _oracle_coarse(sd, cfg, data)
    assert o["i_ids"].numel() > 3000
    ex = _strict_coarse(d, o, conf, 0.2, "640x480 planted")
    assert len(ex) <= 3


def _batched_vs_oracle(nb, H, W, seed, label, min_rows_per_pair, noise_rule=False):
    """The batch ``bench.py`` times, held to the oracle: ONE ``HipLoFTR.forward`` over ``nb`` same-shape pairs against ONE
    batched oracle forward -- the reference sends same-shape frames through the backbone as one 2N batch
    (third_party/LoFTR/src/loftr/loftr.py:45-47) and runs the self layers on N-sequence batches
    (loftr_module/transformer.py:90-97), so batch size is part of the configuration: it changes the launch shapes here
    (`use_ln160(M)`, the chunking of enc256_kv, K3's tile count and workspace carving).  Checked pair by pair under the plain
    north_star rules (no oracle-noise rule): rows identical, confidences within 1e-4."""
    cfg, sd, m = _loftr(0.2)
    data = synth.coarse_pair_batch(nb, H, W, seed=seed)
    d = synth.to_device(data, DEV)
    m(d)
    o, conf = _oracle_coarse(sd, cfg, data)
    total_ex = []
    hb, ob = d["b_ids"].cpu(), o["b_ids"]
    for p in range(nb):
        hs, os_ = (hb == p), (ob == p)
        hp = {k: d[k].cpu()[hs] for k in MATCH_KEYS}
        op = {k: o[k][os_] for k in MATCH_KEYS if k in o}
        hp["b_ids"], op["b_ids"] = torch.zeros_like(hp["b_ids"]), torch.zeros_like(op["b_ids"])
        assert op["i_ids"].numel() > min_rows_per_pair, (p, op["i_ids"].numel())
        exact = (o["feat_c0"][p:p + 1], o["feat_c1"][p:p + 1], cfg["match_coarse"]["dsmax_temperature"]) if noise_rule else None
        total_ex += _strict_coarse(hp, op, conf[p:p + 1], 0.2, f"{label} pair {p}", exact=exact)
    noise = [e for e in total_ex if e[0] == "oracle-noise"]
    assert len(total_ex) - len(noise) <= 3 * nb
    assert len(noise) <= (0.005 * o["i_ids"].numel() if noise_rule else 0)
    # whole-table layout as the plugin reads it: ascending (b, i), every pair present
    key = d["b_ids"].cpu() * (H // 8) * (W // 8) + d["i_ids"].cpu()
    assert (key[1:] > key[:-1]).all() and set(hb.tolist()) == set(range(nb))


def test_loftr_benched_batch8_640x480_vs_oracle(built_lib):
    """BASELINE configs[1] exactly as ``bench.py`` steps it (rank 0, first rotating batch: ``coarse_pair_batch(8, 480, 640,
    seed=1000)``, planted weights, thr 0.2): all 8 pairs against the batched oracle."""
    _batched_vs_oracle(8, 480, 640, 1000, "bench batch 8x640x480", 3000)


def test_loftr_benched_batch4_832_vs_oracle(built_lib):
    """``bench.py --workload hires832`` step (configs[4] frame size, 4 pairs, seed 500) against the batched oracle."""
    _batched_vs_oracle(4, 832, 832, 500, "bench batch 4x832x832", 6000)


def test_loftr_benched_batch2_1600x1064_vs_oracle(built_lib):
    """``bench.py --workload eth3d1600`` step (the frame size of hydra_configs/eth3d_sfm/dfsfm.yaml:76, 2 pairs, seed 500: L = S =
    26 600) against the batched oracle.  At this grid size ATen's fp32 softmax is itself up to 1.15e-4 from the exact confidence of
    its own features (tests/parity.py rule "oracle-noise", profiles/r04_loftr_hires_noise_study.txt): entries beyond 1e-4 of the
    fp32 oracle must be within 1e-4 of the float64 value -- the one test of the three bench batches that needs the rule."""
    _batched_vs_oracle(2, 1064, 1600, 500, "bench batch 2x1600x1064", 8000, noise_rule=True)


def test_flattened_k_conv_schedule_equals_same_schedule(built_lib):
    """``model.same_conv = False`` packs the stride-1 3x3 / 5x5 convolutions for the flattened-K kernel instead of the
    tap-reuse schedule (another summation order of the same products): both matchers must still meet the oracle under the
    north_star rules, and agree with the default schedule to rounding."""
    cfg, sd, m = _loftr(0.2)
    alt = HipLoFTR(cfg)
    alt.same_conv = False
    alt.load_state_dict(sd, strict=True)
    alt = alt.eval().to(DEV)
    data = synth.coarse_pair_batch(2, 96, 128, seed=1000)
    d, e = synth.to_device(data, DEV), synth.to_device(data, DEV)
    m(d)
    alt(e)
    assert not any(pw.tap_padded for pw in _packed_convs(alt)) and any(pw.tap_padded for pw in _packed_convs(m))
    o, conf = _oracle_coarse(sd, cfg, data)
    assert o["i_ids"].numel() > 100
    assert len(_strict_coarse(e, o, conf, 0.2, "flattened-K backbone")) <= 2
    assert _same_tables({k: d[k].cpu() for k in MATCH_KEYS}, {k: e[k].cpu() for k in MATCH_KEYS}, 0.2) > 100

    rcfg, rsd, rm = _refiner(1)
    ralt = HipMultiviewMatcher(rcfg, test=True)
    ralt.same_conv = False
    ralt.load_state_dict(rsd, strict=True)
    ralt = ralt.eval().to(DEV)
    rdata = synth.refine_bag(T=40, V=4, H=120, W=160, seed=2300, variable_lengths=True)
    rd = synth.to_device(rdata, DEV)
    ralt(rd)
    with torch.no_grad():
        ro = restate.multiview_matcher_forward(rsd, rcfg, rdata)
    assert len(_strict_refine(rd, ro, rdata, 7, "flattened-K S2DNet")) <= 2


def _packed_convs(model):
    P = model._packed
    found = []

    def walk(x):
        if hasattr(x, "tap_padded"):
            found.append(x)
        elif isinstance(x, dict):
            for v in x.values():
                walk(v)
        elif isinstance(x, (list, tuple)):
            for v in x:
                walk(v)
    walk(P)
    return found


def test_refine_backbone_patch_chunks_are_invisible(built_lib):
    """``max_backbone_patches`` bounds S2DNet's activation buffers by running the patches in chunks; patches are independent
    and no kernel's summation order depends on how many share a launch, so a 64-patch chunking must reproduce the one-pass
    result exactly."""
    cfg, sd, m = _refiner(1)
    small = HipMultiviewMatcher(cfg, test=True, max_backbone_patches=64)
    small.load_state_dict(sd, strict=True)
    small = small.eval().to(DEV)
    data = synth.refine_bag(T=96, V=4, H=120, W=160, seed=2400, variable_lengths=True)
    a, b = synth.to_device(data, DEV), synth.to_device(data, DEV)
    m(a)
    small(b)
    assert small.max_backbone_patches == 64 and m.max_backbone_patches > 96 * 4
    assert torch.equal(a["query_points_refined"], b["query_points_refined"])
    assert torch.equal(a["reference_points_refined"][-1], b["reference_points_refined"][-1])
    assert torch.equal(a["std"][-1], b["std"][-1])


def test_refine_fused_front_equals_three_launches(built_lib):
    """``model.fused_front = False`` runs S2DNet's conv1_1 / conv1_2 / max-pool as the three separate launches the fused front
    end (csrc/s2d_front.hip, the default) replaces: both must meet the oracle, and agree with each other far inside the tolerance
    (conv1_2 sums the same products in another K order)."""
    cfg, sd, m = _refiner(1)
    alt = HipMultiviewMatcher(cfg, test=True)
    alt.fused_front = False
    alt.load_state_dict(sd, strict=True)
    alt = alt.eval().to(DEV)
    data = synth.refine_bag(T=64, V=5, H=120, W=160, seed=2310, variable_lengths=True)
    a, b = synth.to_device(data, DEV), synth.to_device(data, DEV)
    assert m.fused_front
    m(a)
    alt(b)
    with torch.no_grad():
        o = restate.multiview_matcher_forward(sd, cfg, data)
    assert len(_strict_refine(a, o, data, 7, "fused S2DNet front end")) <= 2
    assert len(_strict_refine(b, o, data, 7, "three-launch S2DNet front end")) <= 2
    same = (a["query_points_refined"] == b["query_points_refined"]).all(-1)[0]
    assert int(same.sum()) >= same.numel() - 2               # a near-tied candidate may flip between two summation orders
    d = (a["reference_points_refined"][-1] - b["reference_points_refined"][-1]).abs().amax(-1)[0][:, same]
    assert float(d.max()) < 1e-3


def test_refine_direct_split_features_equal_split_rows_path(built_lib):
    """r06: on a bag whose every (track, view) slot is valid the backbone writes its features as split planes in the transformer's
    order and the first encoder layer reads them directly (``model.direct_features``, the default); ``False`` = fp32 features +
    split_rows as before.  The same values take the same split, so the refined points must be IDENTICAL; a ragged bag takes the old
    path whatever the flag says, and both meet the oracle."""
    cfg, sd, m = _refiner(1)
    alt = HipMultiviewMatcher(cfg, test=True)
    alt.direct_features = False
    alt.load_state_dict(sd, strict=True)
    alt = alt.eval().to(DEV)
    data = synth.refine_bag(T=96, V=5, H=120, W=160, seed=2311)                       # dense: all five views of every track
    warm = synth.to_device(synth.refine_bag(T=8, V=5, H=120, W=160, seed=1), DEV)   # closes the first-call range sweep (it takes the old path)
    m(dict(warm)); alt(dict(warm))
    a, b = synth.to_device(data, DEV), synth.to_device(data, DEV)
    calls = []
    keep = ops.split_rows
    ops.split_rows = lambda *x, **k: (calls.append(1), keep(*x, **k))[1]
    try:
        m(a)
        n_direct = len(calls)
        alt(b)
    finally:
        ops.split_rows = keep
    assert n_direct == 0 and len(calls) == 2                   # the direct path has no split_rows launch at all
    assert torch.equal(a["query_points_refined"], b["query_points_refined"])
    assert torch.equal(a["reference_points_refined"][-1], b["reference_points_refined"][-1])
    assert torch.equal(a["std"][-1], b["std"][-1])
    with torch.no_grad():
        o = restate.multiview_matcher_forward(sd, cfg, data)
    assert len(_strict_refine(a, o, data, 7, "direct split-plane features")) <= 2
    ragged = synth.refine_bag(T=64, V=5, H=120, W=160, seed=2312, variable_lengths=True)
    c = synth.to_device(ragged, DEV)
    m(c)
    with torch.no_grad():
        o2 = restate.multiview_matcher_forward(sd, cfg, ragged)
    assert len(_strict_refine(c, o2, ragged, 7, "ragged bag (split_rows path)")) <= 2


def test_loftr_features_vs_oracle_640x480(built_lib):
    """BASELINE config 2 frame size: transformer output features vs the oracle (1e-4 relative)."""
    for planted in (False, True):
        cfg, sd, m = _loftr(0.2, planted=planted)
        data = synth.coarse_pair_batch(1, 480, 640, seed=1000)
        with torch.no_grad():
            f0, f1, hw0, hw1 = m.coarse_features(data["image0"].to(DEV), data["image1"].to(DEV))
            o = restate.loftr_coarse_forward(sd, cfg, data, with_fine_backbone=False)
        assert hw0 == (60, 80)
        for a, b in ((f0, o["feat_c0"]), (f1, o["feat_c1"])):
            err = (a.cpu() - b).abs().max() / b.abs().max()
            assert err < 1e-4, err


def test_loftr_832_config5_batch_invariance(built_lib):
    """BASELINE config 5 frame size (832x832 -> 104x104 grid, L = S = 10816): a batch of 2 pairs gives the same
    features (to summation-order noise) and the same match rows as the two pairs run alone."""
    cfg, sd, m = _loftr(0.2)
    data = synth.coarse_pair_batch(2, 832, 832, seed=5)
    with torch.no_grad():
        f0, f1, hw0, hw1 = m.coarse_features(data["image0"].to(DEV), data["image1"].to(DEV))
        assert hw0 == (104, 104) and f0.shape == (2, 10816, 256)
        assert torch.isfinite(f0).all() and torch.isfinite(f1).all()
        for p in range(2):
            g0, g1, _, _ = m.coarse_features(data["image0"][p:p + 1].to(DEV), data["image1"][p:p + 1].to(DEV))
            for a, b in ((f0[p], g0[0]), (f1[p], g1[0])):
                assert ((a - b).abs().max() / b.abs().max()).item() < 1e-5
    d = synth.to_device(data, DEV)
    m(d)
    assert d["i_ids"].numel() > 10000 and int(d["j_ids"].max()) < 10816
    key = d["b_ids"] * 10816 + d["i_ids"]
    assert (key[1:] > key[:-1]).all()                      # ascending (b, i) like torch.where
    for p in range(2):
        s = synth.to_device({k: v[p:p + 1] for k, v in data.items()}, DEV)
        m(s)
        sel = (d["b_ids"] == p).cpu()
        a = {k: d[k].cpu()[sel] for k in ("i_ids", "j_ids", "mconf")}
        a["b_ids"] = torch.zeros_like(a["i_ids"])
        n = _same_tables(a, {k: s[k].cpu() for k in ("b_ids", "i_ids", "j_ids", "mconf")}, 0.2)
        assert n > 5000


def test_loftr_832_config5_vs_oracle(built_lib):
    """BASELINE configs[4] frame size against the ORACLE (not only against itself): one planted 832x832 pair end to end
    under the per-entry rules -- L = S = 10816, the 117 M-entry confidence matrix of
    third_party/LoFTR/src/loftr/utils/coarse_matching.py:103-116 and the mutual-NN selection :171-193 at the size where a
    row has the most competitors."""
    cfg, sd, m = _loftr(0.2)
    data = synth.coarse_pair_batch(1, 832, 832, seed=51)
    d = synth.to_device(data, DEV)
    m(d)
    o, conf = _oracle_coarse(sd, cfg, data)
    assert tuple(d["hw0_c"]) == (104, 104) and o["i_ids"].numel() > 6000
    ex = _strict_coarse(d, o, conf, 0.2, "832x832 planted")
    assert len(ex) <= 3


def test_loftr_batch_equals_singles(built_lib):
    """Batch of 4 pairs == 4 single-pair calls (pairs are independent units of work)."""
    cfg, sd, m = _loftr(0.2)
    data = synth.coarse_pair_batch(4, 96, 128, seed=77)
    d = synth.to_device(data, DEV)
    m(d)
    tot = 0
    for p in range(4):
        s = synth.to_device({k: v[p:p + 1] for k, v in data.items()}, DEV)
        m(s)
        sel = (d["b_ids"] == p).cpu()
        a = {k: d[k].cpu()[sel] for k in ("i_ids", "j_ids", "mconf")}
        a["b_ids"] = torch.zeros_like(a["i_ids"])
        tot += _same_tables(a, {k: s[k].cpu() for k in ("b_ids", "i_ids", "j_ids", "mconf")}, 0.2)
    assert tot > 200


def test_coarse_plugin_surface(built_lib, tmp_path):
    """The reference's own call sequence (coarse_match_worker.py:21-99,139-141) on the HIP plugin: build_model
    from a checkpoint file with ``matcher.``-prefixed keys -> .cuda() -> extract_matches -> (M,5) table."""
    cfg = loftr_coarse_only_config(0.2)
    sd = planted_loftr_state_dict(loftr_param_spec(cfg), 0)
    ckpt = tmp_path / "loftr.ckpt"
    torch.save({"state_dict": {"matcher." + k: v for k, v in sd.items()}}, ckpt)
    detector, matcher = plugin.build_model({"matcher": "loftr_hip", "type": "coarse_only", "match_thr": 0.2, "seed": 666,
                                            "loftr_hip": {"weight_path": str(ckpt)}})
    detector.cuda()
    matcher.cuda()
    data = synth.coarse_pair_batch(1, 96, 128, seed=1003)
    data["scale0"], data["scale1"] = torch.tensor([[1.5, 2.0]]), torch.tensor([[0.75, 1.25]])
    d = {k: v.cuda() for k, v in data.items()}
    d.update(pair_key="a b", f_name0="a", f_name1="b")
    mk0, mk1, mc = plugin.extract_matches(d, detector=detector, matcher=matcher)
    table = plugin.match_table(d)
    assert table.shape == (len(mc), 5) and table.dtype == np.float32 and len(mc) > 50
    assert np.array_equal(table[:, :2], mk0) and np.array_equal(table[:, 2:4], mk1) and np.array_equal(table[:, 4], mc)
    o, conf = _oracle_coarse(sd, cfg, data)
    ex = _strict_coarse(d, o, conf, 0.2, "coarse plugin")
    assert len(ex) == 0
    assert np.array_equal(mk0, o["mkpts0_f"].numpy()) and np.array_equal(mk1, o["mkpts1_f"].numpy())
    assert np.abs(mc - o["mconf"].numpy()).max() <= parity.TOL_CONF
    with pytest.raises(NotImplementedError):
        plugin.build_model({"matcher": "loftr_official", "match_thr": 0.2})


# ---------------------------------------------------------------- refinement head
def _strict_refine(d, o, data, left, label, tol_px=parity.TOL_PX):
    """d: HIP data dict after forward; o: oracle outputs (or a fixture dict with the same keys + cand_score)."""
    valid = data["track_valid_mask"][0]
    qs = data["scales"][0, data["query_img_idxs"][0]][:, [1, 0]] if "scales" in data else torch.ones_like(data["query_points"][0])
    flips = parity.check_refine(d["query_points_refined"][0], d["reference_points_refined"][-1][0], d["std"][-1][0],
                                o["query_points_refined"][0], o["reference_points_refined"][0], o["std"][0], valid,
                                data["query_points"][0], qs, o["cand_score"], left, tol_px)
    print(f"[{label}] {valid.shape[1]} tracks, argmin flips between near-tied candidates (track, hip, ref, gap): {flips}")
    return flips


def test_multiview_e2e_golden(built_lib, golden):
    gz = golden("multiview_e2e")
    c = _case(gz)
    cfg, sd, m = _refiner(c["weight_seed"])
    data = synth.refine_bag(c["T"], c["V"], c["H"], c["W"], c["data_seed"], variable_lengths=True)
    data["scales"] = torch.from_numpy(gz["scales"])
    d = synth.to_device(data, DEV)
    m(d)
    with torch.no_grad():
        o = restate.multiview_matcher_forward(sd, cfg, data)      # candidate scores (bit-pinned to the fixture on CPU)
    ref = {"query_points_refined": gz["query_points_refined"], "reference_points_refined": gz["reference_points_refined"],
           "std": gz["std"], "cand_score": o["cand_score"]}
    flips = _strict_refine(d, ref, data, 7, "multiview_e2e")
    assert len(flips) <= 1


def test_multiview_vs_oracle_config3_shape(built_lib):
    """BASELINE config 3 shapes at a size the oracle finishes in seconds: 5 views, 480x640 frames, ragged lengths."""
    cfg, sd, m = _refiner(1)
    data = synth.refine_bag(T=60, V=5, H=480, W=640, seed=2000, variable_lengths=True)
    d = synth.to_device(data, DEV)
    m(d)
    with torch.no_grad():
        o = restate.multiview_matcher_forward(sd, cfg, data)
    flips = _strict_refine(d, o, data, 7, "config3 shape")
    assert len(flips) <= 1


def _subset_bag(data, idx):
    """The tracks ``idx`` (ascending -> still sorted by descending length) of a bag, as a bag of their own."""
    part = dict(data)
    for k in ("query_points", "reference_points_coarse", "view_point_vector"):
        part[k] = data[k][..., idx, :]
    for k in ("query_img_idxs", "reference_img_idxs", "track_valid_mask", "scales_relative", "query_movable_mask"):
        part[k] = data[k][..., idx]
    return part


def test_multiview_config3_T2000_ragged(built_lib):
    """BASELINE configs[2] at full size (2000 tracks x 5 views, ragged lengths 2..5, some immovable): tracks are
    independent units, so a seeded sample of 96 tracks is checked against the oracle run on those tracks alone."""
    cfg, sd, m = _refiner(1)
    data = synth.refine_bag(T=2000, V=5, H=480, W=640, seed=2001, variable_lengths=True)
    g = torch.Generator().manual_seed(3)
    data["query_movable_mask"] = (torch.rand((1, 2000), generator=g) > 0.1)
    d = synth.to_device(data, DEV)
    m(d)
    assert torch.isfinite(d["query_points_refined"]).all()
    idx = torch.sort(torch.randperm(2000, generator=g)[:96])[0]
    sub = _subset_bag(data, idx)
    with torch.no_grad():
        o = restate.multiview_matcher_forward(sd, cfg, sub)
    dsub = {"query_points_refined": d["query_points_refined"][:, idx.to(DEV)],
            "reference_points_refined": [d["reference_points_refined"][-1][:, :, idx.to(DEV)]],
            "std": [d["std"][-1][:, :, idx.to(DEV)]]}
    flips = _strict_refine(dsub, o, sub, 7, "config3 T=2000 ragged (96-track sample)")
    assert len(flips) <= 2
    imm = ~data["query_movable_mask"][0]
    assert torch.equal(d["query_points_refined"][0].cpu()[imm], data["query_points"][0][imm])     # unmovable => centre


def _whole_bag_vs_oracle(cfg, sd, m, data, label, chunk=250):
    """Every track of ``data`` against the oracle: tracks are independent units (attention batch dimension = track,
    src/MultiviewMatcher/matcher_module/transformer.py:132-178), so the oracle runs the bag in chunks of 250 tracks (its S2DNet pass
    holds ~1.5 GB per 1000 patches) while the HIP model ran it as ONE launch sequence."""
    d = synth.to_device(data, DEV)
    m(d)
    T = data["query_points"].shape[1]
    q, r, sdv = d["query_points_refined"], d["reference_points_refined"][-1], d["std"][-1]
    assert q.shape == (1, T, 2) and torch.isfinite(q).all() and torch.isfinite(r).all()
    flips = []
    for lo in range(0, T, chunk):
        idx = torch.arange(lo, min(T, lo + chunk))
        sub = _subset_bag(data, idx)
        with torch.no_grad():
            o = restate.multiview_matcher_forward(sd, cfg, sub)
        dsub = {"query_points_refined": q[:, idx.to(DEV)], "reference_points_refined": [r[:, :, idx.to(DEV)]],
                "std": [sdv[:, :, idx.to(DEV)]]}
        flips += [(lo + f[0],) + tuple(f[1:]) for f in _strict_refine(dsub, o, sub, 7, f"{label}, tracks {lo}..{lo + len(idx) - 1}")]
    return d, flips


def test_multiview_benched_bag_vs_oracle(built_lib):
    """The refinement bag ``bench.py`` steps (configs[2]: ``synth.refine_bag(2000, 5, 480, 640, seed=2000)``, every view valid, seeded
    weights 1 -- `secondary` of the bench line) held to the oracle, ALL 2000 tracks (r06; r05 checked a 128-track sample).  With the
    seeded weights the fine heat-maps are flat and a track's 49 candidate scores differ by ~1e-5: a handful of tracks may pick the
    other of two candidates whose ORACLE scores are within 1e-5 (listed; tests/parity.py); every other track agrees within 1e-4 px
    on all three outputs, and the whole bag stays inside the +-7.5 px search window."""
    cfg, sd, m = _refiner(1)
    data = synth.refine_bag(2000, 5, 480, 640, seed=2000)
    d, flips = _whole_bag_vs_oracle(cfg, sd, m, data, "bench bag 2000 x 5 (whole)")
    assert (d["reference_points_refined"][-1].cpu() - data["reference_points_coarse"]).abs().max().item() <= 7.5
    print(f"[bench bag, whole] {len(flips)} near-tied argmin flips of 2000 tracks: {flips}")
    assert len(flips) <= 20


def test_multiview_planted_weights_no_flips(built_lib):
    """``params.planted_multiview_state_dict``: peaked heat-maps, candidate scores spread over (0.05, 1.6) with the best two of a
    track >= 1e-3 apart, so the argmin over candidates is decided by margins far above rounding and the expectation / std
    arithmetic (fine_matching.py:195-219, 258-285) carries the comparison: NO track may pick another candidate, on the whole
    2000-track bag with ragged track lengths and on the benched all-valid bag's first 500 tracks."""
    from detectorfreesfm_amd.params import planted_multiview_state_dict
    cfg = multiview_refinement_config()
    sd = planted_multiview_state_dict(multiview_param_spec(cfg), 1)
    m = HipMultiviewMatcher(cfg, test=True)
    m.load_state_dict(sd, strict=True)
    m = m.eval().to(DEV)
    data = synth.refine_bag(2000, 5, 480, 640, seed=2001, variable_lengths=True)
    d, flips = _whole_bag_vs_oracle(cfg, sd, m, data, "planted weights, 2000 ragged tracks")
    assert flips == []
    with torch.no_grad():
        o = restate.multiview_matcher_forward(sd, cfg, _subset_bag(data, torch.arange(0, 250)))
    s2 = torch.sort(o["cand_score"], 1)[0]
    assert float((s2[:, 1] - s2[:, 0]).min()) > 1e-4 and float(s2[:, -1].max() - s2[:, 0].min()) > 0.5     # peaked, decidable
    data = _subset_bag(synth.refine_bag(2000, 5, 480, 640, seed=2000), torch.arange(0, 500))
    d, flips = _whole_bag_vs_oracle(cfg, sd, m, data, "planted weights, bench bag tracks 0..499")
    assert flips == []


def test_multiview_chunk16000_equals_two_chunks(built_lib):
    """BASELINE config 5 refinement chunk (chunk_size = 16000 tracks, 5 views): tracks are independent units and no
    kernel's summation order depends on how many tracks share a launch, so one 16000-track bag must reproduce its
    two 8000-track halves exactly."""
    cfg, sd, m = _refiner(1)
    T = 16000
    data = synth.refine_bag(T, 5, 480, 640, seed=31)
    d = synth.to_device(data, DEV)
    m(d)
    q, r, s = d["query_points_refined"], d["reference_points_refined"][-1], d["std"][-1]
    assert q.shape == (1, T, 2) and r.shape == (1, 4, T, 2)
    assert torch.isfinite(q).all() and torch.isfinite(r).all()
    # refined points stay inside the search window around the coarse points (W = 15 -> +-7 px)
    assert (r.cpu() - data["reference_points_coarse"]).abs().max().item() <= 7.5
    for lo, hi in ((0, 8000), (8000, T)):
        h = synth.to_device(_subset_bag(data, torch.arange(lo, hi)), DEV)
        m(h)
        assert torch.equal(h["query_points_refined"], q[:, lo:hi])
        assert (h["reference_points_refined"][-1] - r[:, :, lo:hi]).abs().max().item() <= 1e-6
        assert (h["std"][-1] - s[:, :, lo:hi]).abs().max().item() <= 1e-6
    # ... and the 16000-track launch itself against the ORACLE (r05): a seeded sample of 64 of its tracks, spread over the whole chunk
    # (first / last tiles of the persistent encoder grid, the S2DNet pass boundary at 16384 patches), run through the oracle alone
    g = torch.Generator().manual_seed(5)
    idx = torch.sort(torch.cat([torch.tensor([0, 1, 3275, 3276, 3277, 15998, 15999]), torch.randperm(T, generator=g)[:57]]).unique())[0]
    sub = _subset_bag(data, idx)
    with torch.no_grad():
        o = restate.multiview_matcher_forward(sd, cfg, sub)
    dsub = {"query_points_refined": q[:, idx.to(DEV)], "reference_points_refined": [r[:, :, idx.to(DEV)]], "std": [s[:, :, idx.to(DEV)]]}
    assert len(_strict_refine(dsub, o, sub, 7, f"chunk 16000 ({len(idx)}-track sample)")) <= 2


def test_refine_plugin_surface(built_lib, tmp_path):
    """The reference's call sequence (multiview_match_worker.py:16-82) on the HIP plugin: build_model from a Lightning
    checkpoint (``matcher.`` prefix, ``loftr_fine`` -> ``fine_transformer``, foreign keys dropped) -> .cuda() ->
    extract_results -> the two [points, img_ids, pt2d_idxs] lists."""
    cfg = multiview_refinement_config()
    sd = random_state_dict(multiview_param_spec(cfg), 1)
    ck = {("matcher." + k.replace("fine_transformer", "loftr_fine")): v for k, v in sd.items()}
    ck["matcher.loftr_coarse.layers.0.q_proj.weight"] = torch.zeros(4, 4)
    ck["loss.some_buffer"] = torch.zeros(3)
    path = tmp_path / "mv.ckpt"
    torch.save({"state_dict": ck}, path)
    matcher = plugin.build_refine_model({"weight_path": [str(path)], "seed": 666}, None, 0)
    matcher.cuda()
    T, V = 48, 4
    data = synth.refine_bag(T, V, 120, 160, seed=2100, variable_lengths=True)
    g = torch.Generator().manual_seed(8)
    data["query_movable_mask"] = torch.rand((1, T), generator=g) > 0.2
    data["query_img_ids"] = torch.randint(1, 50, (1, T), generator=g)
    data["query_pt2d_idxs"] = torch.randint(0, 5000, (1, T), generator=g)
    data["reference_img_ids"] = torch.randint(1, 50, (1, V - 1, T), generator=g)
    data["reference_pt2d_idxs"] = torch.randint(0, 5000, (1, V - 1, T), generator=g)
    d = synth.to_device(data, DEV)
    (q_pts, q_ids, q_idx), (r_pts, r_ids, r_idx), _ = plugin.extract_results(d, matcher=matcher)
    with torch.no_grad():
        o = restate.multiview_matcher_forward(sd, cfg, data)
    flips = _strict_refine(d, o, data, 7, "refine plugin")
    ok = np.ones(T, bool)
    ok[[f[0] for f in flips]] = False
    mask = data["track_valid_mask"].numpy()
    mov = data["query_movable_mask"].numpy()
    assert q_pts.shape == (mask.sum(), 2) and r_pts.shape == (mov.sum(), 2)
    assert np.array_equal(q_ids, data["reference_img_ids"].numpy()[mask]) and np.array_equal(q_idx, data["reference_pt2d_idxs"].numpy()[mask])
    assert np.array_equal(r_ids, data["query_img_ids"].numpy()[mov]) and np.array_equal(r_idx, data["query_pt2d_idxs"].numpy()[mov])
    keep = np.broadcast_to(ok[None, None], mask.shape)[mask]
    assert np.abs(q_pts - o["reference_points_refined"].numpy()[mask])[keep].max() <= parity.TOL_PX
    keep = ok[None][mov]
    assert np.abs(r_pts - o["query_points_refined"].numpy()[mov])[keep].max() <= parity.TOL_PX
    # window shrink of later refinement iterations (multiview_match_worker.py:20-34)
    m2 = plugin.build_refine_model({"weight_path": [None]}, rewindow_size_factor=2)
    assert m2.config["multiview_transform"]["window_size"] == 11
    assert m2.config["multiview_matching_test"]["left_point_movement_window_size"] == 3


def test_multiview_16_views_ragged_vs_oracle(built_lib):
    """The production track length: ``max_track_length: 16`` (src/post_optimization/post_optimization.py:25) => one
    reference view + up to 15 query views per track.  A ragged bag with lengths 2..16 (so every view-count group of
    MultiviewMatcher.py:117-133 from 15 query views down to 1 occurs) through the whole head, against the oracle."""
    cfg, sd, m = _refiner(1)
    data = synth.refine_bag(T=40, V=16, H=240, W=320, seed=2200, variable_lengths=True)
    lens = data["track_valid_mask"][0].sum(0) + 1
    assert int(lens.max()) == 16 and int(lens.min()) <= 4
    g = torch.Generator().manual_seed(9)
    data["scales"] = 0.75 + 0.5 * torch.rand((1, 16, 2), generator=g)
    d = synth.to_device(data, DEV)
    m(d)
    with torch.no_grad():
        o = restate.multiview_matcher_forward(sd, cfg, data)
    flips = _strict_refine(d, o, data, 7, "16 views, ragged")
    assert len(flips) <= 2
    assert d["reference_points_refined"][-1].shape == (1, 15, 40, 2)


def test_multiview_second_iteration_window_vs_oracle(built_lib):
    """The matcher of the second refinement iteration: ``build_model(args, rewindow_size_factor=2)`` shrinks the search window
    to W = 11 and the left-point window to 3 (multiview_match_worker.py:20-34; crop 35 kept, so S2DNet's centre slice, the
    bicubic window, K1's 121-token groups and K11's 9 candidates all change shape)."""
    sd = random_state_dict(multiview_param_spec(multiview_refinement_config()), 1)
    m = plugin.build_refine_model({"weight_path": [None], "seed": 1}, rewindow_size_factor=2)
    m.load_state_dict(sd, strict=True)
    m = m.eval().to(DEV)
    cfg = m.config
    assert cfg["multiview_transform"]["window_size"] == 11 and cfg["multiview_matching_test"]["left_point_movement_window_size"] == 3
    data = synth.refine_bag(T=48, V=5, H=240, W=320, seed=2300, variable_lengths=True)
    g = torch.Generator().manual_seed(4)
    data["query_movable_mask"] = torch.rand((1, 48), generator=g) > 0.15
    d = synth.to_device(data, DEV)
    m(d)
    with torch.no_grad():
        o = restate.multiview_matcher_forward(sd, cfg, data)
    assert o["cand_score"].shape[-1] == 9
    flips = _strict_refine(d, o, data, 3, "W=11 / left=3")
    assert len(flips) <= 2
    # refined references stay inside the shrunk window
    dr = (d["reference_points_refined"][-1].cpu() - data["reference_points_coarse"]).abs()
    assert dr[data["track_valid_mask"]].max().item() <= 5.5


def test_multiview_padded_image_tensor_vs_oracle(built_lib):
    """``data['images']`` as ONE padded tensor [1, N, 3, h, w] (MultiviewMatcher.py:60-62, 201-203: the reference slices
    ``images[:, img_idx]``) instead of a list: frames of different sizes padded bottom / right with zeros, tracks placed so
    that crops run across the frame border into the padding (RoIAlign then normalises by the PADDED size)."""
    cfg, sd, m = _refiner(1)
    T, V, hp, wp = 36, 4, 200, 264
    data = synth.refine_bag(T=T, V=V, H=hp, W=wp, seed=2400, variable_lengths=True)
    sizes = [(200, 264), (168, 264), (200, 216), (152, 200)]
    imgs = torch.zeros((1, V, 3, hp, wp))
    for v, (h, w) in enumerate(sizes):
        imgs[0, v, :, :h, :w] = data["images"][v][0, :, :h, :w]
    g = torch.Generator().manual_seed(12)
    for v, (h, w) in enumerate(sizes):                     # a third of the points of every view hug its own frame border
        pts = data["query_points"][0] if v == 0 else data["reference_points_coarse"][0, v - 1]
        sel = torch.rand(T, generator=g) < 0.34
        pts[sel, 0] = w - 1 - 6 * torch.rand(int(sel.sum()), generator=g)
        pts[sel, 1] = torch.minimum(pts[sel, 1], torch.tensor(float(h - 1)))
    data_t = dict(data)
    data_t["images"] = imgs
    d = synth.to_device({k: v for k, v in data_t.items()}, DEV)
    assert isinstance(d["images"], torch.Tensor) and d["images"].shape == (1, V, 3, hp, wp)
    m(d)
    data_l = dict(data)
    data_l["images"] = [imgs[:, v].contiguous() for v in range(V)]
    with torch.no_grad():
        o = restate.multiview_matcher_forward(sd, cfg, data_l)
    flips = _strict_refine(d, o, data_l, 7, "padded image tensor")
    assert len(flips) <= 2


def test_scene_matching_cached_tokens_equals_pairwise(built_lib):
    """plugin.match_scene_cached: backbone once per image, every pair matched from the cached tokens -- the same
    tables as feeding each pair through HipLoFTR.forward."""
    cfg, sd, m = _loftr(0.2)
    base = synth.coarse_pair_batch(3, 96, 128, seed=1000)
    images = torch.cat([base["image0"], base["image1"]], 0)                         # 6 images
    pairs = [(i, j) for i in range(6) for j in range(i + 1, 6)]
    tables = plugin.match_scene_cached(m, images, pairs, batch=4)
    total = 0
    for (i, j) in pairs:
        d = synth.to_device({"image0": images[i:i + 1], "image1": images[j:j + 1]}, DEV)
        m(d)
        ref = torch.cat([d["mkpts0_f"], d["mkpts1_f"], d["mconf"][:, None]], -1).cpu().numpy()
        got = tables[(i, j)]
        A = {tuple(r[:4]): r[4] for r in got}
        B = {tuple(r[:4]): r[4] for r in ref}
        bad = [k for k in set(A) ^ set(B) if abs((A.get(k) if k in A else B.get(k)) - 0.2) > parity.TOL_THR]
        bad += [k for k in set(A) & set(B) if abs(A[k] - B[k]) > parity.TOL_CONF]
        assert not bad, bad[:5]
        total += len(B)
    assert total > 100


def test_scene_path_640x480_vs_oracle(built_lib):
    """BASELINE configs[3] at its frame size, against the ORACLE (r05): the first five frames of ``bench.py --workload scene300``'s camera
    sweep (same generator), all ten pairs through ``plugin.match_scene_cached`` -- backbone once per image, pairs of DIFFERENT images
    sharing a transformer batch, the pipelined match-count read -- and every per-pair table against the oracle's forward on that pair
    under the north_star rules."""
    cfg, sd, m = _loftr(0.2)
    g = torch.Generator().manual_seed(4242)
    base = torch.rand((1, 1, 480, 640), generator=g)
    images = torch.cat([torch.roll(base, shifts=(8 * (k % 7), 8 * k), dims=(2, 3)) + 0.02 * torch.randn((1, 1, 480, 640), generator=g)
                        for k in range(5)], 0)
    pairs = [(i, j) for i in range(5) for j in range(i + 1, 5)]
    tables = plugin.match_scene_cached(m, images, pairs, batch=4, to_host=False)      # 4 + 4 + 2 pairs: a partial last batch
    total = 0
    for (i, j) in pairs:
        data = {"image0": images[i:i + 1], "image1": images[j:j + 1]}
        o, conf = _oracle_coarse(sd, cfg, data)
        t = tables[(i, j)].cpu()
        # rows are (x0, y0, x1, y1, conf) on the 8-px coarse grid at scale 1: recover the cell indices
        w_c = 640 // 8
        hip = {"b_ids": torch.zeros(len(t), dtype=torch.long), "i_ids": (t[:, 1] / 8).long() * w_c + (t[:, 0] / 8).long(),
               "j_ids": (t[:, 3] / 8).long() * w_c + (t[:, 2] / 8).long(), "mconf": t[:, 4],
               "mkpts0_f": t[:, 0:2], "mkpts1_f": t[:, 2:4]}
        ex = _strict_coarse(hip, o, conf, 0.2, f"scene pair {(i, j)}")
        assert len(ex) <= 3
        total += int(o["i_ids"].numel())
    assert total > 10 * 2000


def test_refine_scene_worker_vs_oracle(built_lib):
    """The reference's worker loop (multiview_match_worker.py:111-141) on a synthetic COLMAP-shaped scene: BagPlanner
    bags -> DeviceUpdatedQueryPts -> HipMultiviewMatcher -> [M,4] rows, against the oracle fed with the same bags."""
    from detectorfreesfm_amd.bags import BagPlanner
    from detectorfreesfm_amd.synth import SyntheticSfMScene
    cfg, sd, m = _refiner(1)
    scene = SyntheticSfMScene(n_images=20, n_points=150, seed=5, hw=(120, 160), max_views=7)
    dcfg = {"max_track_length": 6, "chunk": 60}
    results = plugin.match_tracks_worker(scene, m, None, dcfg, device=DEV)
    planner = BagPlanner(scene, dcfg)
    assert len(results) == len(planner) > 2
    n_rows = 0
    for k in range(len(planner)):
        bag = {key: (v if isinstance(v, list) else v[None]) for key, v in planner.bag_tensors(k).items()}
        bag["images"] = [im[None] for im in bag["images"]]
        bag["query_movable_mask"] = torch.ones_like(bag["query_img_idxs"], dtype=torch.bool)   # the reference's lookup never hits
        with torch.no_grad():
            o = restate.multiview_matcher_forward(sd, cfg, bag)
        mask = bag["track_valid_mask"].numpy()
        T = mask.shape[-1]
        rows = results[k]
        assert rows.shape == (mask.sum() + T, 4) and rows.dtype == np.float64
        assert np.array_equal(rows[:mask.sum(), 2], bag["reference_img_ids"].numpy()[mask])
        assert np.array_equal(rows[:mask.sum(), 3], bag["reference_pt2d_idxs"].numpy()[mask])
        assert np.array_equal(rows[mask.sum():, 2], bag["query_img_ids"].numpy()[0])
        # per-track comparison with the argmin-tie rule of tests/parity.py
        hip = {"query_points_refined": torch.from_numpy(rows[mask.sum():, :2])[None], "std": [o["std"]]}
        r = np.zeros(mask.shape + (2,))
        r[mask] = rows[:mask.sum(), :2]
        hip["reference_points_refined"] = [torch.from_numpy(r)]
        flips = parity.check_refine(hip["query_points_refined"][0], hip["reference_points_refined"][-1][0], o["std"][0],
                                    o["query_points_refined"][0], o["reference_points_refined"][0], o["std"][0],
                                    bag["track_valid_mask"][0], bag["query_points"][0],
                                    bag["scales"][0][bag["query_img_idxs"][0]][:, [1, 0]], o["cand_score"], 7)
        assert len(flips) <= 3, flips           # each one already verified as a < 1e-5 score tie by check_refine
        n_rows += rows.shape[0]
    assert n_rows > 400


# ---------------------------------------------------------------- MatchFormer-LA coarse matcher
def _matchformer(thr=0.2):
    from detectorfreesfm_amd import HipMatchformer, matchformer_coarse_only_config
    from detectorfreesfm_amd.params import matchformer_param_spec, planted_matchformer_state_dict
    cfg = matchformer_coarse_only_config(thr)
    sd = planted_matchformer_state_dict(matchformer_param_spec(), 0)
    m = HipMatchformer(cfg)
    m.load_state_dict(sd, strict=True)
    return cfg, sd, m.eval().to(DEV)


@pytest.mark.parametrize("tag", ["plain", "masked"])
def test_matchformer_e2e_golden(built_lib, golden, tag):
    """Fixture written by the real Matchformer module (oracle/make_golden.py matchformer): 2 pairs, per-pair scales,
    without / with padding masks; same per-entry rules as the LoFTR tests."""
    from oracle import restate_matchformer as rmf
    from oracle.make_golden import matchformer_masks
    gz = golden("matchformer_e2e")
    c = _case(gz)
    cfg, sd, m = _matchformer(c["thr"])
    data = synth.coarse_pair_batch(c["n_pairs"], c["H"], c["W"], c["data_seed"])
    data["scale0"], data["scale1"] = torch.from_numpy(gz["scale0"]), torch.from_numpy(gz["scale1"])
    if tag == "masked":
        data["mask0"], data["mask1"] = matchformer_masks(c["n_pairs"], c["H"] // 8, c["W"] // 8)
    d = synth.to_device(data, DEV)
    m(d)
    with torch.no_grad():
        o = rmf.matchformer_forward(sd, cfg, data, with_fine_backbone=False)
    ref = {k: gz[f"{tag}_{k}"] for k in MATCH_KEYS}
    ex = _strict_coarse(d, ref, o["conf_matrix"], c["thr"], f"matchformer_e2e {tag}")
    assert len(ex) <= 1 and len(ref["i_ids"]) > 60 and (d["m_bids"] == d["b_ids"]).all()


def test_matchformer_240x320_vs_oracle(built_lib):
    """A larger frame (1200 coarse cells per image, all four stages with their real token counts: 19200 / 4800 / 1200 /
    300): every match row identical to the oracle's."""
    from oracle import restate_matchformer as rmf
    cfg, sd, m = _matchformer(0.2)
    data = synth.coarse_pair_batch(1, 240, 320, seed=7)
    d = synth.to_device(data, DEV)
    m(d)
    with torch.no_grad():
        o = rmf.matchformer_forward(sd, cfg, data, with_fine_backbone=False)
    assert o["i_ids"].numel() > 500
    ex = _strict_coarse(d, o, o["conf_matrix"], 0.2, "matchformer 240x320")
    assert len(ex) <= 2


def test_matchformer_plugin_surface(built_lib, tmp_path):
    """build_model('matchformer_hip') from a bare state-dict checkpoint (coarse_match_worker.py:59-74) -> extract_matches."""
    from detectorfreesfm_amd.params import matchformer_param_spec, planted_matchformer_state_dict
    from oracle import restate_matchformer as rmf
    sd = planted_matchformer_state_dict(matchformer_param_spec(), 0)
    ckpt = tmp_path / "matchformer.ckpt"
    torch.save({"matcher." + k: v for k, v in sd.items()}, ckpt)
    detector, matcher = plugin.build_model({"matcher": "matchformer_hip", "type": "coarse_only", "match_thr": 0.2, "seed": 666,
                                            "matchformer_hip": {"weight_path": str(ckpt)}})
    matcher.cuda()
    data = synth.coarse_pair_batch(1, 96, 128, seed=1003)
    d = {k: v.cuda() for k, v in data.items()}
    mk0, mk1, mc = plugin.extract_matches(d, detector=detector, matcher=matcher)
    with torch.no_grad():
        o = rmf.matchformer_forward(sd, matcher.config, data, with_fine_backbone=False)
    ex = _strict_coarse(d, o, o["conf_matrix"], 0.2, "matchformer plugin")
    assert len(ex) == 0 and len(mc) > 30
    assert np.array_equal(mk0, o["mkpts0_f"].numpy()) and np.abs(mc - o["mconf"].numpy()).max() <= parity.TOL_CONF


@pytest.mark.parametrize("which", ["loftr_hip", "matchformer_hip", "aspanformer_hip"])
def test_match_worker_from_frames(built_lib, which):
    """plugin.match_worker on the device: decoded uint8 frames -> device LANCZOS resize / pad (per-matcher rule of
    coarse_match.py:82-90) -> matcher -> per-pair tables, against the oracle readers + oracle matcher on the same frames
    (per-entry rules of tests/parity.py; the table rows of every common match are identical)."""
    import test_match_worker_cpu as mw
    frames = mw.scene_frames()
    cfgs, (detector, matcher), oracle = mw.build(which)
    seen = []

    class Tap(torch.nn.Module):                       # records what the matcher wrote for each pair
        def __init__(self, inner):
            super().__init__()
            self.inner = inner

        def forward(self, data):
            self.inner(data)
            seen.append({k: data[k] for k in MATCH_KEYS})
    got = plugin.match_worker([0, 1, 2], list(frames), mw.PAIRS, cfgs, device=DEV, frames=frames, models=(detector, Tap(matcher)))
    rule = plugin._DATA_RULES[which]
    n = 0
    for p, d in zip(mw.PAIRS, seen):
        p0, p1 = p.split(" ")
        (i0, s0, _, _), (i1, s1, _, _) = (mw.rr.read_image(frames[q], resize=(128,), df=rule["df"], pad_to=rule["pad_to"]) for q in (p0, p1))
        data = {"image0": torch.from_numpy(i0)[None], "image1": torch.from_numpy(i1)[None],
                "scale0": torch.from_numpy(s0)[None], "scale1": torch.from_numpy(s1)[None]}
        with torch.no_grad():
            o = oracle(data)
            conf = o["conf_matrix"] if "conf_matrix" in o else restate.dual_softmax_conf(o["feat_c0"], o["feat_c1"], 0.1)
        ex = _strict_coarse(d, o, conf, 0.2, f"match_worker {which} {p}")
        assert len(ex) <= 1
        t = got[p]
        assert t.shape == (d["mconf"].shape[0], 5) and np.array_equal(t[:, :2], d["mkpts0_f"].cpu().numpy())
        n += t.shape[0]
    assert n > 30
