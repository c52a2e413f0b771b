"""Host-side feeding of the refinement head (detectorfreesfm_amd/bags.py) against the reference's OWN classes
(MatchingMultiviewData / FeatureTrackStatus / UpdatedQueryPts, compiled unchanged from its source files by
oracle/ref_import.py) on seeded synthetic COLMAP-shaped scenes: bag assignment, every tensor of every bag, and the
refined-keypoint table across a whole sequence of bags."""
import numpy as np
import pytest
import torch

from detectorfreesfm_amd import dist as ddist
from detectorfreesfm_amd.bags import BagPlanner, DeviceUpdatedQueryPts
from detectorfreesfm_amd.synth import SyntheticSfMScene
from oracle import ref_import

needs_ref = pytest.mark.skipif(not ref_import.reference_available(), reason="needs /root/reference (build container only)")

CASES = [  # (scene kwargs, dataset config, worker split)
    (dict(n_images=10, n_points=200, seed=0), {"max_track_length": 16, "chunk": 6000}, None),
    (dict(n_images=24, n_points=400, seed=1, max_views=24), {"max_track_length": 9, "chunk": 50}, None),      # long tracks split, bags chunked
    (dict(n_images=6, n_points=120, seed=2, with_scale=False), {"max_track_length": 4, "chunk": 6000}, None),
    (dict(n_images=22, n_points=300, seed=3, max_views=14), {"max_track_length": 16, "chunk": 6000}, "shard"),              # a rank's track shard
]


def _eq(a, b):
    if isinstance(a, torch.Tensor):
        assert a.shape == b.shape and a.dtype.is_floating_point == b.dtype.is_floating_point, (a.shape, b.shape, a.dtype, b.dtype)
        if a.dtype.is_floating_point and a.dtype == torch.float64:
            return torch.allclose(a, b, rtol=1e-12, atol=1e-12)
        return torch.equal(a.to(b.dtype), b)
    return a == b


@needs_ref
@pytest.mark.parametrize("case", range(len(CASES)))
def test_bags_equal_reference(case):
    kw, cfg, split = CASES[case]
    scene = SyntheticSfMScene(**kw)
    idxs = ddist.shard_tracks(len(scene.point_cloud_assigned_imgID_kptID), 1, 3, seed=7) if split else None
    RefData, _ = ref_import.import_matching_data()
    ref = RefData(scene, cfg, worker_split_idxs=idxs)
    mine = BagPlanner(scene, cfg, worker_split_idxs=idxs)
    assert len(ref) == len(mine) >= (1 if case == 0 else 2)
    for rb, mb in zip(ref.image_bags, mine.image_bags):
        assert [int(i) for i in rb["bag_image_ids"]] == [int(i) for i in mb["bag_image_ids"]]
        assert [int(t) for t in rb["track_ids"]] == [int(t) for t in mb["track_ids"]]
        assert [[int(c[0]), [int(x) for x in c[1]]] for c in rb["track_corresponding_imgs"]] == \
               [[int(c[0]), [int(x) for x in c[1]]] for c in mb["track_corresponding_imgs"]]
    n_tracks = 0
    for k in range(len(ref)):
        rd, md = ref[k], mine[k]
        assert set(rd.keys()) == set(md.keys())
        for key in rd:
            if key == "images":
                assert all(torch.equal(a, b) for a, b in zip(rd[key], md[key]))
            else:
                assert _eq(md[key], rd[key]), (k, key)
        n_tracks += rd["query_points"].shape[0]
        lens = rd["track_valid_mask"].sum(0)
        assert (lens[:-1] >= lens[1:]).all()                  # tracks arrive sorted by descending length
    assert n_tracks >= (len(idxs) if split else kw["n_points"])


@needs_ref
@pytest.mark.parametrize("reference_lookup", [True, False])
def test_updated_query_pts_equal_reference(reference_lookup):
    """The worker loop of multiview_match_worker.py:111-141 with a stand-in 'refinement' (a deterministic shift): the
    reference's dict-of-dicts and the dense table give the same query points / movable masks for every bag.
    reference_lookup=True: the reference exactly as its worker drives it (keypoint indices arrive as 0-dim tensors, whose
    identity hash never matches the stored numpy keys -> nothing is ever found, every point stays movable).
    reference_lookup=False: the same unmodified class driven with numpy indices, so its lookup works as evidently intended."""
    scene = SyntheticSfMScene(n_images=24, n_points=400, seed=1, max_views=24)
    cfg = {"max_track_length": 9, "chunk": 50}
    _, RefBuf = ref_import.import_matching_data()
    planner = BagPlanner(scene, cfg)
    ref_buf = RefBuf(scene.colmap_images)
    dev_buf = DeviceUpdatedQueryPts(scene.colmap_images, reference_lookup=reference_lookup)
    n_fixed = 0
    for k in range(len(planner)):
        bag = planner.bag_tensors(k, with_images=False)
        rd = {key: v[None].clone() for key, v in bag.items()}            # DataLoader adds the batch dimension
        md = {key: v[None].clone() for key, v in bag.items()}
        if not reference_lookup:
            rd["query_pt2d_idxs"] = rd["query_pt2d_idxs"].numpy()        # numpy integers hash like the stored keys
        ref_buf.find_movable_and_update(rd)
        dev_buf.find_movable_and_update(md)
        assert torch.equal(rd["query_movable_mask"], md["query_movable_mask"])
        assert torch.equal(rd["query_points"], md["query_points"])
        n_fixed += int((~md["query_movable_mask"]).sum())
        mov = md["query_movable_mask"][0]
        refined = md["query_points"][0][mov] + 0.25 * (k + 1)            # what extract_results hands back (:59-82)
        ids, kps = md["query_img_ids"][0][mov], md["query_pt2d_idxs"][0][mov]
        ref_buf.update_query_pts(list(refined), ids.numpy(), kps.numpy())   # tensor rows: the reference torch.stack()s them
        dev_buf.update_query_pts(refined, ids, kps)
    assert (n_fixed == 0) if reference_lookup else (n_fixed > 10)        # long tracks really came back in later bags
    assert int(dev_buf.moved.sum()) > 300
    # a repeated key inside one call: the last row wins, as in the dict assignment
    dev_buf.update_query_pts(torch.tensor([[1.0, 1.0], [2.0, 2.0]]), torch.tensor([1, 1]), torch.tensor([0, 0]))
    assert torch.equal(dev_buf.xy[dev_buf._key(torch.tensor([1]), torch.tensor([0]))], torch.tensor([[2.0, 2.0]]))


def test_bags_equal_golden(golden):
    """The same comparison against the committed fixture the reference produced (oracle/make_golden.py bags): runs
    where /root/reference does not exist."""
    import json
    gz = golden("bags")
    planner = BagPlanner(SyntheticSfMScene(**json.loads(str(gz["scene"]))), json.loads(str(gz["cfg"])))
    want = json.loads(str(gz["bags"]))
    got = [{"bag_image_ids": [int(i) for i in b["bag_image_ids"]], "track_ids": [int(t) for t in b["track_ids"]],
            "track_corresponding_imgs": [[int(c[0]), [int(x) for x in c[1]]] for c in b["track_corresponding_imgs"]]}
           for b in planner.image_bags]
    assert got == want and len(want) > 5
    tensors = [planner.bag_tensors(i) for i in range(len(planner))]
    for key in ("query_points", "reference_points_coarse", "track_valid_mask", "query_img_idxs", "reference_img_idxs",
                "query_img_ids", "query_pt2d_idxs", "reference_img_ids", "reference_pt2d_idxs"):
        cat = np.concatenate([t[key].numpy().reshape(-1) for t in tensors])
        assert np.array_equal(cat, gz[key]), key
    for key in ("scales_relative", "view_point_vector"):
        cat = np.concatenate([t[key].numpy().reshape(-1) for t in tensors])
        assert np.allclose(cat, gz[key], rtol=1e-12, atol=1e-12), key


def test_shard_tracks_partitions_and_keeps_tracks_whole():
    """Track-id sharding (multiview_match.py:39-43, chunk_index_balance): every track index lands on exactly one rank,
    sizes differ by at most one, and because a rank builds its bags from its own track subset, all bags of one long
    track (split every max_track_length-1 views and chained through UpdatedQueryPts) stay on that rank."""
    n = 1003
    shards = [ddist.shard_tracks(n, r, 8, seed=5) for r in range(8)]
    flat = sorted(i for s in shards for i in s)
    assert flat == list(range(n)) and max(map(len, shards)) - min(map(len, shards)) <= 1
    assert shards[0] != list(range(0, n, 8))                               # shuffled like the reference
    assert [ddist.shard_tracks(10, r, 3) for r in range(3)] == [[0, 3, 6, 9], [1, 4, 7], [2, 5, 8]]
    scene = SyntheticSfMScene(n_images=24, n_points=300, seed=4, max_views=24)
    cfg = {"max_track_length": 6, "chunk": 6000}
    seen = {}
    for r in range(4):
        p = BagPlanner(scene, cfg, worker_split_idxs=ddist.shard_tracks(300, r, 4, seed=1))
        for b in p.image_bags:
            for t in b["track_ids"]:
                assert seen.setdefault(int(t), r) == r
    assert len(seen) == 300
