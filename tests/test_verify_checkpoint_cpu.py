"""tools/verify_checkpoint.py (the maintainer-side real-checkpoint check) exercised end to end without a GPU: seeded
checkpoints written in the Lightning layout the reference's builders read (third_party/LoFTR/src/loftr/loftr.py:83-87;
src/post_optimization/matcher_model/multiview_match_worker.py:40-53), frames on disk, ``--cpu-standins`` behind ``ops``."""
import importlib.util
import os

import numpy as np
import pytest
import torch

from detectorfreesfm_amd import synth
from detectorfreesfm_amd.config import loftr_coarse_only_config, multiview_refinement_config
from detectorfreesfm_amd.params import loftr_param_spec, multiview_param_spec, planted_loftr_state_dict, random_state_dict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tool():
    spec = importlib.util.spec_from_file_location("verify_checkpoint", os.path.join(ROOT, "tools", "verify_checkpoint.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="module")
def scene(tmp_path_factory):
    from PIL import Image
    d = tmp_path_factory.mktemp("scene")
    pair = synth.coarse_pair_batch(1, 96, 128, seed=1000)
    third = synth.coarse_pair_batch(1, 96, 128, seed=1001)["image1"]
    for k, im in enumerate((pair["image0"], pair["image1"], third)):
        a = (im[0, 0].clamp(0, 1) * 255).round().byte().numpy()
        Image.fromarray(np.stack([a, a, a], -1)).save(d / f"frame{k}.png")
    cfg = loftr_coarse_only_config(0.2)
    sd = planted_loftr_state_dict(loftr_param_spec(cfg), 0)
    lck = d / "outdoor_ds.ckpt"
    torch.save({"state_dict": {"matcher." + k: v for k, v in sd.items()}, "epoch": 3}, lck)
    rsd = random_state_dict(multiview_param_spec(multiview_refinement_config()), 1)
    ck = {("matcher." + k.replace("fine_transformer", "loftr_fine")): v for k, v in rsd.items()}
    ck["matcher.loftr_coarse.layers.0.q_proj.weight"] = torch.zeros(4, 4)      # foreign keys the reference's loader drops
    ck["loss.some_buffer"] = torch.zeros(3)
    rck = d / "multiview_matcher.ckpt"
    torch.save({"state_dict": ck}, rck)
    return str(d), str(lck), str(rck)


def test_verify_checkpoint_reports_clean(scene, capsys):
    img_dir, lck, rck = scene
    vc = _tool()
    rc = vc.main(["--loftr-ckpt", lck, "--refine-ckpt", rck, "--images", img_dir, "--resize", "128", "--tracks", "24",
                  "--cpu-standins"])
    text = capsys.readouterr().out
    assert rc == 0, text
    assert "[coarse] parity vs oracle: OK" in text and "[refine] parity vs oracle: OK" in text
    assert "[coarse] range:" in text and "[refine] range:" in text and "headroom" in text
    assert "coarse matches of the pair" in text                     # the pair's own matches became the refinement tracks
    assert "ALL CLEAN" in text


def test_verify_checkpoint_three_views_and_saturating_weights(scene, tmp_path, capsys):
    """A three-view synthetic bag (no coarse step); and a checkpoint whose activations leave the split-plane range is reported,
    not clamped silently."""
    img_dir, lck, rck = scene
    vc = _tool()
    assert vc.main(["--refine-ckpt", rck, "--images", img_dir, "--resize", "128", "--views", "3", "--tracks", "16",
                    "--cpu-standins"]) == 0
    assert "seeded synthetic tracks" in capsys.readouterr().out
    ck = torch.load(lck, map_location="cpu")
    k = "matcher.backbone.layer1.0.conv1.weight"
    ck["state_dict"][k] = ck["state_dict"][k] * 2e5
    hot = tmp_path / "hot.ckpt"
    torch.save(ck, hot)
    from detectorfreesfm_amd import _lib
    with pytest.raises(_lib.DfsfmError, match="split-plane range"):
        vc.main(["--loftr-ckpt", str(hot), "--images", img_dir, "--resize", "128", "--cpu-standins"])
