"""Host-side feeding of the refinement head: how many tracks per second do bag assignment + bag tensors +
the refined-keypoint table deliver, next to the reference's own classes (compiled unchanged from
src/post_optimization/data_construct/construct_matching_data.py:10-261, 317-476 and multiview_match_worker.py:85-108) and to
the device's refinement rate?  CPU only: python tools/bench_bags.py [n_images] [n_points] [ref_points]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from detectorfreesfm_amd.bags import BagPlanner, DeviceUpdatedQueryPts  # noqa: E402
from detectorfreesfm_amd.synth import SyntheticSfMScene  # noqa: E402

n_images = int(sys.argv[1]) if len(sys.argv) > 1 else 300
n_points = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
ref_points = int(sys.argv[3]) if len(sys.argv) > 3 else 4000
cfg = {"max_track_length": 16, "chunk": 2000}          # post_optimization.py:25 and the bench's 2000-track bags
torch.set_num_threads(8)


def run(cls, updated_cls, scene, with_images):
    t0 = time.perf_counter()
    data = cls(scene, cfg)
    t_assign = time.perf_counter() - t0
    upd = updated_cls(data.colmap_images) if updated_cls is not None else None
    n_tracks, t_bags, t_upd = 0, 0.0, 0.0
    for k in range(len(data)):
        t1 = time.perf_counter()
        bag = data.bag_tensors(k, with_images=with_images) if hasattr(data, "bag_tensors") else data[k]
        t_bags += time.perf_counter() - t1
        T = bag["query_points"].shape[0]
        n_tracks += T
        if upd is not None:
            t1 = time.perf_counter()
            b1 = {kk: (v[None] if isinstance(v, torch.Tensor) else v) for kk, v in bag.items() if kk != "images"}
            upd.find_movable_and_update(b1)
            mov = b1["query_movable_mask"]
            upd.update_query_pts(b1["query_points"][mov], b1["query_img_ids"][mov], b1["query_pt2d_idxs"][mov])
            t_upd += time.perf_counter() - t1
    return len(data), n_tracks, t_assign, t_bags, t_upd


t0 = time.perf_counter()
scene = SyntheticSfMScene(n_images=n_images, n_points=n_points, seed=11, hw=(480, 640), max_views=24)
print(f"scene: {n_images} images, {n_points} 3D points, built in {time.perf_counter() - t0:.1f} s")
nb, nt, ta, tb, tu = run(BagPlanner, DeviceUpdatedQueryPts, scene, with_images=False)
tot = ta + tb + tu
print(f"BagPlanner + DeviceUpdatedQueryPts (host, torch CPU): {nb} bags, {nt} tracks: assign {ta:.2f} s, bag tensors {tb:.2f} s, "
      f"query-point table {tu:.2f} s -> {nt / tot:.0f} tracks/s ({nt / tb:.0f} tracks/s for the tensors alone)")
try:
    from oracle import ref_import
    if ref_import.reference_available():
        small = SyntheticSfMScene(n_images=n_images, n_points=ref_points, seed=11, hw=(480, 640), max_views=24)
        RefData, RefUpd = ref_import.import_matching_data()

        class RefWrap(RefData):
            colmap_images_attr = None
        nb2, nt2, ta2, tb2, _ = run(RefData, None, small, with_images=True)
        nb3, nt3, ta3, tb3, _ = run(BagPlanner, None, small, with_images=False)
        print(f"reference MatchingMultiviewData on {ref_points} points ({nt2} tracks, incl. its image reads): assign {ta2:.2f} s, "
              f"__getitem__ {tb2:.2f} s -> {nt2 / (ta2 + tb2):.0f} tracks/s;  BagPlanner on the same scene: {nt3 / (ta3 + tb3):.0f} tracks/s")
except Exception as e:          # the comparison needs the reference tree
    print("reference classes not available:", type(e).__name__, e)
