"""Pins the oracle (oracle/restate.py) against fixtures produced by the REAL reference
(oracle/make_golden.py), and -- when /root/reference is present -- against the live reference."""
import numpy as np
import pytest
import torch

from detectorfreesfm_amd import synth
from detectorfreesfm_amd.config import loftr_coarse_only_config, multiview_refinement_config
from detectorfreesfm_amd.params import (loftr_param_spec, multiview_param_spec, planted_loftr_state_dict,
                                        random_state_dict)
from oracle import ref_import, restate
from oracle.make_golden import fine_inputs, la_inputs


def _case(npz):
    return {k: npz[k].item() for k in npz.files if npz[k].ndim == 0}


def test_linear_attention_d32(golden):
    gz = golden("linear_attention")
    q, k, v, qm, km = la_inputs(_case(gz))
    assert np.array_equal(restate.linear_attention(q, k, v, qm, km).numpy(), gz["out"])
    assert np.array_equal(restate.linear_attention(q, k, v).numpy(), gz["out_nomask"])


def test_linear_attention_d16(golden):
    gz = golden("linear_attention_d16")
    c = _case(gz)
    q, k, v, qm, km = la_inputs(c)
    out = restate.linear_attention(q, k, v, None, km.repeat_interleave(c["kv_group"], dim=1))
    assert np.array_equal(out.numpy(), gz["out"])


def test_coarse_matching(golden):
    gz = golden("coarse_matching")
    c = _case(gz)
    f0, f1 = synth.correlated_features(c["N"], c["h0"] * c["w0"], c["h1"] * c["w1"], c["C"], c["seed"], c["noise"])
    o = restate.coarse_matching(f0, f1, (c["h0"], c["w0"]), (c["h1"], c["w1"]), (c["h0"] * 8, c["w0"] * 8), c["thr"],
                                c["border"], 0.1, torch.from_numpy(gz["scale0"]), torch.from_numpy(gz["scale1"]))
    assert len(gz["i_ids"]) > 100
    for k in ("b_ids", "i_ids", "j_ids", "mconf", "mkpts0_c", "mkpts1_c"):
        assert np.array_equal(o[k].numpy(), gz[k]), k


def test_fine_matching(golden):
    gz = golden("fine_matching")
    c = _case(gz)
    ref, qry, mask, movable = fine_inputs(c)
    left_norm, coords, std, best = restate.fine_matching(ref, qry, c["W"], c["left"], mask, movable)
    qr = torch.from_numpy(gz["qpts"])[0] + left_norm * (c["left"] // 2) * torch.from_numpy(gz["sq"])[0]
    rr = torch.from_numpy(gz["rpts"])[0] + coords * (c["W"] // 2) * torch.from_numpy(gz["sr"])[0].transpose(0, 1)
    assert np.array_equal(qr.numpy(), gz["query_refined"][0])
    assert np.array_equal(rr.transpose(0, 1).numpy(), gz["ref_refined"][0])
    assert np.array_equal(std.transpose(0, 1).numpy(), gz["std"][0])
    assert (best[~movable] == (c["left"] ** 2) // 2).all()


def test_loftr_e2e(golden):
    gz = golden("loftr_e2e")
    c = _case(gz)
    cfg = loftr_coarse_only_config(c["thr"])
    sd = random_state_dict(loftr_param_spec(cfg), c["weight_seed"])
    data = synth.coarse_pair_batch(c["n_pairs"], c["H"], c["W"], c["data_seed"])
    data["scale0"], data["scale1"] = torch.from_numpy(gz["scale0"]), torch.from_numpy(gz["scale1"])
    with torch.no_grad():
        o = restate.loftr_coarse_forward(sd, cfg, data)
    assert len(gz["i_ids"]) > 10
    for k in ("b_ids", "i_ids", "j_ids", "mconf", "mkpts0_f", "mkpts1_f"):
        assert np.array_equal(o[k].numpy(), gz[k]), k


def test_loftr_e2e_planted(golden):
    """Planted weights (confident matches at thr 0.2), 2 pairs, per-pair scales: oracle == real reference."""
    gz = golden("loftr_e2e_planted")
    c = _case(gz)
    cfg = loftr_coarse_only_config(c["thr"])
    sd = planted_loftr_state_dict(loftr_param_spec(cfg), c["weight_seed"], c["alpha"])
    data = synth.coarse_pair_batch(c["n_pairs"], c["H"], c["W"], c["data_seed"])
    data["scale0"], data["scale1"] = torch.from_numpy(gz["scale0"]), torch.from_numpy(gz["scale1"])
    with torch.no_grad():
        o = restate.loftr_coarse_forward(sd, cfg, data)
    assert len(gz["i_ids"]) > 100 and gz["mconf"].max() > 0.9
    for k in ("b_ids", "i_ids", "j_ids", "mconf", "mkpts0_f", "mkpts1_f"):
        assert np.array_equal(o[k].numpy(), gz[k]), k


def test_loftr_e2e_two_sizes(golden):
    """Frames of different size (two backbone calls, L != S; loftr.py:45-49): oracle == real reference."""
    gz = golden("loftr_e2e_two_sizes")
    c = _case(gz)
    cfg = loftr_coarse_only_config(c["thr"])
    sd = planted_loftr_state_dict(loftr_param_spec(cfg), c["weight_seed"], c["alpha"])
    data = synth.coarse_pair_two_sizes(c["H0"], c["W0"], c["H1"], c["W1"], c["data_seed"])
    data["scale0"], data["scale1"] = torch.from_numpy(gz["scale0"]), torch.from_numpy(gz["scale1"])
    with torch.no_grad():
        o = restate.loftr_coarse_forward(sd, cfg, data)
    assert len(gz["i_ids"]) > 30
    for k in ("b_ids", "i_ids", "j_ids", "mconf", "mkpts0_f", "mkpts1_f"):
        assert np.array_equal(o[k].numpy(), gz[k]), k


def test_loftr_e2e_masked(golden):
    """Padded frames with mask0 / mask1 (loftr.py:61-65; mask_border_with_padding, coarse_matching.py:25-41): oracle ==
    real reference, and no match lies in the padding or in the last two valid rows / columns."""
    gz = golden("loftr_e2e_masked")
    c = _case(gz)
    cfg = loftr_coarse_only_config(c["thr"])
    sd = planted_loftr_state_dict(loftr_param_spec(cfg), c["weight_seed"], c["alpha"])
    data = synth.coarse_pair_padded(c["n_pairs"], c["H"], c["W"], c["data_seed"])
    data["scale0"], data["scale1"] = torch.from_numpy(gz["scale0"]), torch.from_numpy(gz["scale1"])
    with torch.no_grad():
        o = restate.loftr_coarse_forward(sd, cfg, data)
    assert len(gz["i_ids"]) > 60
    for k in ("b_ids", "i_ids", "j_ids", "mconf", "mkpts0_f", "mkpts1_f"):
        assert np.array_equal(o[k].numpy(), gz[k]), k
    wc = c["W"] // 8
    for n in range(c["n_pairs"]):
        sel = gz["b_ids"] == n
        h0, w0 = int(data["mask0"][n].sum(0).max()), int(data["mask0"][n].sum(1).max())
        assert (gz["i_ids"][sel] // wc < h0 - 2).all() and (gz["i_ids"][sel] % wc < w0 - 2).all()
        assert (gz["i_ids"][sel] // wc >= 2).all() and (gz["i_ids"][sel] % wc >= 2).all()


def test_multiview_e2e(golden):
    gz = golden("multiview_e2e")
    c = _case(gz)
    cfg = multiview_refinement_config()
    sd = random_state_dict(multiview_param_spec(cfg), c["weight_seed"])
    data = synth.refine_bag(c["T"], c["V"], c["H"], c["W"], c["data_seed"], variable_lengths=True)
    data["scales"] = torch.from_numpy(gz["scales"])
    with torch.no_grad():
        o = restate.multiview_matcher_forward(sd, cfg, data)
    mask = data["track_valid_mask"].numpy()
    assert np.array_equal(o["query_points_refined"].numpy(), gz["query_points_refined"])
    assert np.array_equal(o["reference_points_refined"].numpy()[mask], gz["reference_points_refined"][mask])
    assert np.array_equal(o["std"].numpy()[mask], gz["std"][mask])


def test_roi_align_known_answers():
    """KATs for the restated (parity-unpinned) RoIAlign: integer-centred boxes copy pixels exactly,
    samples outside [0,size-1] are 0, half-pixel shifts average neighbours."""
    g = torch.Generator().manual_seed(3)
    img = torch.rand((1, 3, 40, 50), generator=g)
    pts = torch.tensor([[25.0, 20.0], [2.0, 3.0], [48.0, 38.0], [10.5, 12.0]])
    p = restate.extract_local_patches(img, pts, 7)
    # x/(W-1)*(W-1) is not exact in fp32 (upstream's arithmetic), so "copies" hold to ~1e-6
    assert torch.allclose(p[0], img[0, :, 17:24, 22:29], atol=1e-5)
    assert torch.equal(p[1, :, :, 0], torch.zeros(3, 7))                      # x = -1: outside
    assert torch.allclose(p[1, :, :, 1:], img[0, :, 0:7, 0:6], atol=1e-5)
    assert torch.equal(p[2, :, -2:, :], torch.zeros(3, 2, 7)) and torch.equal(p[2, :, :, -2:], torch.zeros(3, 7, 2))
    assert torch.allclose(p[2, :, :5, :5], img[0, :, 35:40, 45:50], atol=1e-5)
    assert torch.allclose(p[3], 0.5 * (img[0, :, 9:16, 7:14] + img[0, :, 9:16, 8:15]), atol=1e-5)


@pytest.mark.skipif(not ref_import.reference_available(), reason="needs /root/reference (build container only)")
def test_oracle_vs_live_reference_and_key_layout():
    """Live cross-check + the product's parameter layout equals the reference's state_dict keys."""
    LoFTR, _ = ref_import.import_loftr()
    cfg = loftr_coarse_only_config(1e-3)
    m = LoFTR(cfg).eval()
    spec = loftr_param_spec(cfg)
    assert [n for n, _, _ in spec] == list(m.state_dict().keys())
    assert all(tuple(m.state_dict()[n].shape) == tuple(s) for n, s, _ in spec)
    MM = ref_import.import_multiview_matcher()
    rcfg = multiview_refinement_config()
    mm = MM(rcfg, test=True).eval()
    rspec = multiview_param_spec(rcfg)
    assert [n for n, _, _ in rspec] == list(mm.state_dict().keys())
    assert all(tuple(mm.state_dict()[n].shape) == tuple(s) for n, s, _ in rspec)
    # different image sizes for the two views (two backbone calls, L != S)
    sd = random_state_dict(spec, 5)
    m.load_state_dict({k: v.clone() for k, v in sd.items()}, strict=True)
    g = torch.Generator().manual_seed(9)
    data = {"image0": torch.rand((1, 1, 64, 96), generator=g), "image1": torch.rand((1, 1, 80, 72), generator=g)}
    with torch.no_grad():
        d = dict(data)
        m(d)
        o = restate.loftr_coarse_forward(sd, cfg, data)
    for k in ("i_ids", "j_ids", "mconf", "mkpts0_f", "mkpts1_f"):
        assert torch.equal(d[k], o[k]), k


# ---------------------------------------------------------------- match-table consumers
def test_merge_oracle_vs_golden(golden):
    """oracle/restate_merge.py against fixtures written by the reference's own Match2Kpts / keypoint_worker /
    update_matches / transform_keypoints (oracle/make_golden.py merge): bit-for-bit."""
    from oracle import restate_merge as rm
    gz = golden("merge_keypoints")
    for tag in ("a", "b", "c"):
        kp, sc, off, ids = rm.merge_keypoints(gz[f"{tag}_rows"], gz[f"{tag}_img0"], gz[f"{tag}_img1"],
                                              int(gz[f"{tag}_n_images"]))
        assert np.array_equal(off, gz[f"{tag}_offsets"])
        assert np.array_equal(kp, gz[f"{tag}_kpts"]) and np.array_equal(sc, gz[f"{tag}_scores"])
        assert np.array_equal(ids, gz[f"{tag}_ids"])


def test_merge_oracle_vs_live_reference():
    from oracle import ref_import, restate_merge as rm
    if not ref_import.reference_available():
        pytest.skip("reference tree not present")
    Match2Kpts, keypoint_worker, update_matches, transform_keypoints = ref_import.import_match_table_consumers()
    matches, names, split = rm.synthetic_scene(7, 15, seed=11)
    keypoints = keypoint_worker(Match2Kpts(matches, names, name_split=split)[0:len(names)], verbose=False)
    upd = update_matches(matches, keypoints, merge=False, verbose=False, pair_name_split=split)
    fk, fs = transform_keypoints(keypoints, verbose=False)
    rows, i0, i1, sl = rm.tables_to_flat(matches, names, split)
    kp, sc, off, ids = rm.merge_keypoints(rows, i0, i1, len(names))
    for i, n in enumerate(names):
        assert np.array_equal(kp[off[i]:off[i + 1]], np.asarray(fk[n], np.float32).reshape(-1, 2))
        assert np.array_equal(sc[off[i]:off[i + 1]], fs[n])
    for k, (lo, hi) in sl.items():
        assert np.array_equal(ids[lo:hi], upd[k].reshape(-1, 2))


# ---------------------------------------------------------------- MatchFormer-LA
def _mf_inputs(gz, tag):
    from oracle.make_golden import matchformer_masks
    c = _case(gz)
    data = synth.coarse_pair_batch(c["n_pairs"], c["H"], c["W"], c["data_seed"])
    data["scale0"], data["scale1"] = torch.from_numpy(gz["scale0"]), torch.from_numpy(gz["scale1"])
    if tag == "masked":
        data["mask0"], data["mask1"] = matchformer_masks(c["n_pairs"], c["H"] // 8, c["W"] // 8)
    return c, data


@pytest.mark.parametrize("tag", ["plain", "masked"])
def test_matchformer_e2e(golden, tag):
    """oracle/restate_matchformer.py == the real Matchformer module (fixture), bit for bit, with and without padding masks."""
    from detectorfreesfm_amd.matchformer import matchformer_coarse_only_config
    from detectorfreesfm_amd.params import matchformer_param_spec, planted_matchformer_state_dict
    from oracle import restate_matchformer as rmf
    gz = golden("matchformer_e2e")
    c, data = _mf_inputs(gz, tag)
    cfg = matchformer_coarse_only_config(c["thr"])
    sd = planted_matchformer_state_dict(matchformer_param_spec(), c["weight_seed"], c["alpha"])
    with torch.no_grad():
        o = rmf.matchformer_forward(sd, cfg, data)
    assert len(gz[f"{tag}_i_ids"]) > 60
    for k in ("b_ids", "i_ids", "j_ids", "mconf", "mkpts0_f", "mkpts1_f"):
        assert np.array_equal(o[k].numpy(), gz[f"{tag}_{k}"]), k


@pytest.mark.skipif(not ref_import.reference_available(), reason="needs /root/reference (build container only)")
def test_matchformer_key_layout_vs_live_reference():
    from detectorfreesfm_amd.matchformer import matchformer_coarse_only_config
    from detectorfreesfm_amd.params import matchformer_param_spec
    MF = ref_import.import_matchformer()
    m = MF(matchformer_coarse_only_config(0.2)).eval()
    spec = matchformer_param_spec()
    assert [n for n, _, _ in spec] == list(m.state_dict().keys()) and len(spec) == 229
    assert all(tuple(m.state_dict()[n].shape) == tuple(s) for n, s, _ in spec)


# ---------------------------------------------------------------- ASpanFormer
ASPAN_KEYS = ("b_ids", "i_ids", "j_ids", "mconf", "mkpts0_f", "mkpts1_f", "conf_matrix", "offset_bids_left", "offset_lids_left",
              "confleft", "offset_kpts0_f_left", "offset_kpts1_f_left", "offset_bids_right", "offset_lids_right", "confright",
              "offset_kpts0_f_right", "offset_kpts1_f_right")


@pytest.mark.parametrize("case", [0, 1, 2])
def test_aspanformer_e2e(golden, case):
    """oracle/restate_aspanformer.py == the real ASpanFormer module (fixture), bit for bit: equal frames, two sizes, and a frame
    that goes through the online resize (whose torchvision call is a stand-in on both sides, see ref_import)."""
    from detectorfreesfm_amd.aspanformer import aspanformer_coarse_only_config
    from detectorfreesfm_amd.params import aspanformer_param_spec, planted_aspanformer_state_dict
    from oracle import restate_aspanformer as ra
    from oracle.make_golden import aspanformer_cases, aspanformer_inputs
    gz = golden("aspanformer_e2e")
    c = _case(gz)
    tag, hw0, hw1 = aspanformer_cases()[case]
    cfg = aspanformer_coarse_only_config(c["thr"])
    sd = planted_aspanformer_state_dict(aspanformer_param_spec(cfg), c["weight_seed"], c["alpha"])
    with torch.no_grad():
        o = ra.aspanformer_forward(sd, cfg, aspanformer_inputs(c, hw0, hw1))
    assert len(gz[f"{tag}_i_ids"]) > 10
    for k in ASPAN_KEYS:
        assert np.array_equal(o[k].numpy(), gz[f"{tag}_{k}"]), k
    pf = o["predict_flow"]
    assert np.array_equal(pf[0].numpy(), gz[f"{tag}_flow0"]) and np.array_equal(pf[1].numpy(), gz[f"{tag}_flow1"])


@pytest.mark.skipif(not ref_import.reference_available(), reason="needs /root/reference (build container only)")
def test_aspanformer_layout_and_config_vs_live_reference():
    from detectorfreesfm_amd.aspanformer import aspanformer_coarse_only_config
    from detectorfreesfm_amd.params import aspanformer_param_spec
    A = ref_import.import_aspanformer()
    cfg = aspanformer_coarse_only_config(0.4)
    ref_cfg = ref_import.aspanformer_coarse_only_config(0.4)                # yacs defaults + aspan_test_coarse_only.py
    for sect in ("coarse", "match_coarse", "fine", "resnetfpn"):
        for k, v in cfg[sect].items():
            assert list(ref_cfg[sect][k]) == list(v) if isinstance(v, (list, tuple)) else ref_cfg[sect][k] == v, (sect, k)
    m = A(config=cfg, online_resize=True).eval()
    spec = aspanformer_param_spec(cfg)
    assert [n for n, _, _ in spec] == list(m.state_dict().keys()) and len(spec) == 217
    assert all(tuple(m.state_dict()[n].shape) == tuple(s) for n, s, _ in spec)
