"""plugin.match_worker / match_worker_sharded: the reference's pair loop (coarse_match_worker.py:101-145) for the three HIP
matchers, from decoded uint8 frames through the device readers to the per-pair tables.  CPU run: the kernels are emulated
(tests/cpu_standins.py); the expected tables come from the oracle readers + oracle matchers on the same frames."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cpu_standins import cpu_ops
from detectorfreesfm_amd import plugin, synth
from oracle import restate, restate_aspanformer as ra, restate_matchformer as rmf, restate_resize as rr


def scene_frames():
    """Three 192x256 uint8 frames: frame k+1 is frame 0's content rolled by whole coarse cells (blocks of 2x2 equal pixels, so
    the 2x LANCZOS reduction to 96x128 keeps the planted matchers' content)."""
    base = synth.coarse_pair_batch(1, 96, 128, seed=1000)
    f = {}
    for name, img in (("scene/a.jpg", base["image0"]), ("scene/b.jpg", base["image1"]),
                      ("scene/c.jpg", torch.roll(base["image0"], shifts=(16, -24), dims=(2, 3)))):
        f[name] = np.kron((img[0, 0].numpy() * 255).round().astype(np.uint8), np.ones((2, 2), dtype=np.uint8))
    return f


def model_and_oracle(which):
    if which == "loftr_hip":
        from detectorfreesfm_amd.config import loftr_coarse_only_config
        from detectorfreesfm_amd.params import loftr_param_spec, planted_loftr_state_dict
        cfg = loftr_coarse_only_config(0.2)
        sd = planted_loftr_state_dict(loftr_param_spec(cfg), 0)
        return cfg, sd, lambda data: restate.loftr_coarse_forward(sd, cfg, data, with_fine_backbone=False)
    if which == "matchformer_hip":
        from detectorfreesfm_amd.matchformer import matchformer_coarse_only_config
        from detectorfreesfm_amd.params import matchformer_param_spec, planted_matchformer_state_dict
        cfg = matchformer_coarse_only_config(0.2)
        sd = planted_matchformer_state_dict(matchformer_param_spec(), 0)
        return cfg, sd, lambda data: rmf.matchformer_forward(sd, cfg, data, with_fine_backbone=False)
    from detectorfreesfm_amd.aspanformer import aspanformer_coarse_only_config
    from detectorfreesfm_amd.params import aspanformer_param_spec, planted_aspanformer_state_dict
    cfg = aspanformer_coarse_only_config(0.2)
    sd = planted_aspanformer_state_dict(aspanformer_param_spec(cfg), 0)
    return cfg, sd, lambda data: ra.aspanformer_forward(sd, cfg, data, with_fine_backbone=False)


def build(which):
    cfg, sd, oracle = model_and_oracle(which)
    detector, matcher = plugin.build_model({"matcher": which, "type": "coarse_only", "match_thr": 0.2, "seed": 0,
                                            which: {"weight_path": None, "cfg": cfg}})
    matcher.load_state_dict(sd, strict=True)
    cfgs = {"data": {"img_resize": 128, "img_type": "grayscale", "img_preload": True},
            "matcher": {"model": {"matcher": which, "type": "coarse_only", "match_thr": 0.2}, "pair_name_split": " ",
                        "round_matches_ratio": 4}}
    return cfgs, (detector, matcher), oracle


PAIRS = ["scene/a.jpg scene/b.jpg", "scene/a.jpg scene/c.jpg", "scene/b.jpg scene/c.jpg"]


def expected_tables(which, oracle, frames):
    rule = plugin._DATA_RULES[which]
    out = {}
    for p in PAIRS:
        p0, p1 = p.split(" ")
        (i0, s0, _, _), (i1, s1, _, _) = (rr.read_image(frames[q], resize=(128,), df=rule["df"], pad_to=rule["pad_to"]) for q in (p0, p1))
        data = {"image0": torch.from_numpy(i0)[None], "image1": torch.from_numpy(i1)[None],
                "scale0": torch.from_numpy(s0)[None], "scale1": torch.from_numpy(s1)[None]}
        with torch.no_grad():
            o = oracle(data)
        out[p] = (o["mkpts0_f"].numpy(), o["mkpts1_f"].numpy(), o["mconf"].numpy())
    return out


@pytest.mark.parametrize("which", ["loftr_hip", "matchformer_hip", "aspanformer_hip"])
def test_match_worker_tables_equal_oracle(which):
    frames = scene_frames()
    cfgs, models, oracle = build(which)
    with cpu_ops():
        got = plugin.match_worker([0, 1, 2], list(frames), PAIRS, cfgs, device="cpu", frames=frames, models=models)
    exp = expected_tables(which, oracle, frames)
    assert list(got) == PAIRS                                              # keys: path0<split>path1, in subset order
    n = 0
    for p in PAIRS:
        t, (m0, m1, mc) = got[p], exp[p]
        assert t.shape == (len(mc), 5) and t.dtype == np.float32
        assert np.array_equal(t[:, :2], m0) and np.array_equal(t[:, 2:4], m1) and np.abs(t[:, 4] - mc).max(initial=0) < 1e-4
        n += len(mc)
    assert n > 30
    with cpu_ops():                                                        # a pair file instead of a list; a subset of it
        import tempfile
        with tempfile.NamedTemporaryFile("w", suffix=".txt", delete=False) as fh:
            fh.write("\n".join(PAIRS) + "\n")
        sub = plugin.match_worker([2], list(frames), fh.name, cfgs, device="cpu", frames=frames, models=models)
        os.unlink(fh.name)
    assert list(sub) == [PAIRS[2]] and np.array_equal(sub[PAIRS[2]], got[PAIRS[2]])


def test_round_matches_ratio_follows_the_reference_identity_test():
    """coarse_match_worker.py:134 guards the grid rounding with ``type is not 'coarse_only'`` (identity).  A literal type string
    is that very object -> no rounding (the case above, ratio 4 set); an equal string built at run time (what YAML / hydra
    hands over) is another object -> the reference rounds, and so does this worker, with the reference's expression."""
    frames = scene_frames()
    cfgs, models, _ = build("loftr_hip")
    with cpu_ops():
        plain = plugin.match_worker([0], list(frames), PAIRS, cfgs, device="cpu", frames=frames, models=models)[PAIRS[0]]
        cfgs["matcher"]["model"]["type"] = "".join(["coarse", "_only"])            # equal, not identical
        assert cfgs["matcher"]["model"]["type"] == "coarse_only" and cfgs["matcher"]["model"]["type"] is not plugin._COARSE_ONLY
        cfgs["matcher"]["round_matches_ratio"] = 12          # coarse cells sit on multiples of 8: ratio 4 would change nothing
        rounded = plugin.match_worker([0], list(frames), PAIRS, cfgs, device="cpu", frames=frames, models=models)[PAIRS[0]]
        cfgs["matcher"]["round_matches_ratio"] = None
        unrounded = plugin.match_worker([0], list(frames), PAIRS, cfgs, device="cpu", frames=frames, models=models)[PAIRS[0]]
    assert np.array_equal(unrounded, plain) and len(plain) > 10
    sc = np.array([[2.0, 2.0]], dtype=np.float32)                                  # 256x192 frames read at 128
    for c in (slice(0, 2), slice(2, 4)):
        assert np.array_equal(rounded[:, c], np.round((plain[:, c] / sc) / 12) * 12 * sc)
    assert np.array_equal(rounded[:, 4], plain[:, 4]) and not np.array_equal(rounded[:, :4], plain[:, :4])


def _sharded(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    frames = scene_frames()
    cfgs, models, _ = build("loftr_hip")
    with cpu_ops():
        on_root = plugin.match_worker_sharded(list(frames), PAIRS, cfgs, device="cpu", frames=frames, models=models, root=0)
        assert (on_root is None) == (rank != 0)                    # opt-in: gather-to-root 0
        got = plugin.match_worker_sharded(list(frames), PAIRS, cfgs, device="cpu", frames=frames, models=models)   # every rank
        if rank == 0:
            assert list(on_root) == PAIRS and all(np.array_equal(on_root[p], got[p]) for p in PAIRS)
            ref = plugin.match_worker([0, 1, 2], list(frames), PAIRS, cfgs, device="cpu", frames=frames, models=models)
            ok = list(got) == PAIRS and all(np.array_equal(got[p], ref[p]) for p in PAIRS) and sum(len(v) for v in got.values()) > 30
        else:
            ok = list(got) == PAIRS
    q.put((rank, bool(ok), int(sum(len(v) for v in got.values()))))
    dist.barrier()
    dist.destroy_process_group()


def test_match_worker_sharded_world2():
    """Two gloo ranks, three pairs: every rank ends with the whole scene's dictionary, equal to the single-process one."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + 733) % 2000
    procs = [ctx.Process(target=_sharded, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok, _ in res) and res[0][2] == res[1][2]


@pytest.mark.parametrize("which", ["loftr_hip", "matchformer_hip", "aspanformer_hip"])
def test_match_worker_empty_tables(which):
    """A threshold nothing passes: every pair still gets its key, with an empty float32 [0,5] table (what the reference's
    np.concatenate of empty predictions gives), and the matchers' data dicts stay well-formed."""
    frames = scene_frames()
    cfgs, (detector, matcher), _ = build(which)
    matcher.config["match_coarse"]["thr"] = 2.0                       # confidences are products of two probabilities: <= 1
    with cpu_ops():
        got = plugin.match_worker([0, 2], list(frames), PAIRS, cfgs, device="cpu", frames=frames, models=(detector, matcher))
    assert list(got) == [PAIRS[0], PAIRS[2]]
    for t in got.values():
        assert t.shape == (0, 5) and t.dtype == np.float32


def test_match_worker_from_jpeg_files(tmp_path):
    """The whole feeding path from FILES (r05): the worker is handed file names (frames=None, the reference's call), every file
    is decoded by jpeg.decode -- here the CPU lane model of the device decoder --, resized and matched; the tables equal the
    oracle matcher on the oracle readers fed the ORACLE decode (oracle/jpeg_baseline.c) of the same files."""
    import io
    from PIL import Image
    from oracle import restate_jpeg as rj
    frames = scene_frames()
    names, decoded = {}, {}
    for k, (name, img) in enumerate(frames.items()):
        b = io.BytesIO()
        Image.fromarray(img).save(b, "JPEG", quality=100, **({"restart_marker_rows": 3} if k == 1 else {}))
        path = tmp_path / os.path.basename(name)
        path.write_bytes(b.getvalue())
        names[name] = str(path)
        decoded[str(path)] = rj.decode(b.getvalue(), False)
        assert np.abs(decoded[str(path)].astype(int) - img).max() <= 3             # quality 100: the planted content survives
    pairs = [" ".join(names[q] for q in p.split(" ")) for p in PAIRS]
    cfgs, models, oracle = build("loftr_hip")
    with cpu_ops():
        got = plugin.match_worker([0, 1, 2], list(names.values()), pairs, cfgs, device="cpu", frames=None, models=models)
    rule = plugin._DATA_RULES["loftr_hip"]
    n = 0
    for p in pairs:
        p0, p1 = p.split(" ")
        (i0, s0, _, _), (i1, s1, _, _) = (rr.read_image(decoded[q], resize=(128,), df=rule["df"], pad_to=rule["pad_to"]) for q in (p0, p1))
        data = {"image0": torch.from_numpy(i0)[None], "image1": torch.from_numpy(i1)[None],
                "scale0": torch.from_numpy(s0)[None], "scale1": torch.from_numpy(s1)[None]}
        with torch.no_grad():
            o = oracle(data)
        t = got[p]
        assert np.array_equal(t[:, :2], o["mkpts0_f"].numpy()) and np.array_equal(t[:, 2:4], o["mkpts1_f"].numpy())
        assert np.abs(t[:, 4] - o["mconf"].numpy()).max(initial=0) < 1e-4
        n += len(t)
    assert n > 30
