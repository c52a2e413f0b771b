"""Host-side feeding of the refinement head (second slice): bag assignment, vectorised bag
tensors, and the device-resident table of already-refined query points.

Reference (paths relative to /root/reference):
* ``MatchingMultiviewData`` / ``FeatureTrackStatus``  src/post_optimization/data_construct/construct_matching_data.py:10-476
  -- ``assign_bags`` :226-261 is a sequential greedy over Python sets (longest track first, then every track whose
  reference node lies in the bag); ``__getitem__`` :317-476 then walks every (track, view) slot in Python to gather
  keypoints, id tables, point scales and view-point vectors.
* ``UpdatedQueryPts``  src/post_optimization/matcher_model/multiview_match_worker.py:85-108 -- a dict of dicts keyed
  (image id, keypoint index) that is probed / filled once per track in Python between two forward passes.

Here: ``BagPlanner.assign`` keeps the greedy on the host (it IS sequential, and its result depends on CPython's set
iteration order, which is reproduced by using the same set expressions on the same integers), but runs on flat arrays
prepared once per scene; ``BagPlanner.bag_tensors`` builds a bag's dict with array gathers (one fancy-indexing pass per
field instead of a Python loop per slot); ``DeviceUpdatedQueryPts`` keeps the refined keypoints in two dense device
tensors so that ``find_movable_and_update`` / ``update_query_pts`` are one gather / one scatter with no host round trip.
Parity: tests/test_bags_cpu.py runs the reference's own classes (compiled unchanged from its source files through
oracle/ref_import.py) on seeded synthetic COLMAP-shaped scenes and compares every field.
"""
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch


def chunks_round_robin(items: Sequence, n_split: int) -> List[list]:
    """``chunks_balance`` (src/utils/ray_utils.py:100-108): element k goes to chunk k % n_split."""
    n_split = max(n_split, 1)
    out = [[] for _ in range(n_split)]
    for k, it in enumerate(items):
        out[k % n_split].append(it)
    return out


class BagPlanner:
    """Drop-in for ``MatchingMultiviewData``: ``len(planner)`` bags, ``planner[i]`` -> the bag's data dict."""

    def __init__(self, colmap_image_dataset, config: dict, worker_split_idxs: Optional[Sequence[int]] = None):
        ds = colmap_image_dataset
        self.max_track_length = config["max_track_length"]
        self.chunk = config["chunk"]
        self.dataset = ds
        self.colmap_images = ds.colmap_images
        self.colmap_3ds = ds.colmap_3ds
        self.intrin_extrin = ds.image_intrin_extrins
        self.frame_dict = ds.keyframe_dict
        assign_all = ds.point_cloud_assigned_imgID_kptID
        if worker_split_idxs is None:
            self.assignment = assign_all
        else:                                   # a rank's shard of the track list (construct_matching_data.py:184-190)
            items = list(assign_all.items())
            self.assignment = {items[i][0]: items[i][1] for i in worker_split_idxs}
        self._prepare()
        self.image_bags = self._chunk(self.assign())

    # -- flat per-track tables -------------------------------------------------------------------------------
    def _prepare(self):
        self.track_ids = list(self.assignment.keys())
        self.track_pos = {t: k for k, t in enumerate(self.track_ids)}
        self.ref_img, self.ref_kpt, self.queries = [], [], []
        for t in self.track_ids:
            img, kpt = self.assignment[t]
            self.ref_img.append(img)
            self.ref_kpt.append(kpt)
            # duplicated image ids collapse; the ORDER is CPython's set order, as in the reference (:37-39)
            self.queries.append(list(set(self.colmap_3ds[t].image_ids) - {img}))
        # per (track, observing image): first keypoint index and the mean position of the track's observations in that image
        # (a track may observe an image more than once; construct_matching_data.py:381-390) -- computed once here with the
        # reference's own expressions instead of once per bag and view
        self._obs = {}
        for t in self.track_ids:
            p3d = self.colmap_3ds[t]
            ids = np.asarray(p3d.image_ids).tolist()
            kps = np.asarray(p3d.point2D_idxs).tolist()
            if len(set(ids)) == len(ids):                    # the common case: one observation per image (mean of one row = the row)
                for img, kp in zip(ids, kps):
                    self._obs[(t, img)] = (kp, np.asarray(self.colmap_images[img].xys[kp], dtype=np.float64))
            else:
                arr = np.asarray(ids)
                for img in set(ids):
                    idx = p3d.point2D_idxs[np.where(arr == img)]
                    self._obs[(t, img)] = (int(idx[0]), np.mean(self.colmap_images[img].xys[idx], axis=0))
        # per image: K, R, t as arrays for the vectorised point-scale / view-point computation
        ids = sorted(self.intrin_extrin.keys())
        self._img_row = {i: k for k, i in enumerate(ids)}
        self._K = np.stack([np.asarray(self.intrin_extrin[i]["intrin"], np.float64) for i in ids])
        self._R = np.stack([np.asarray(self.intrin_extrin[i]["extrin"][0], np.float64) for i in ids])
        self._t = np.stack([np.asarray(self.intrin_extrin[i]["extrin"][1], np.float64) for i in ids])

    # -- greedy bag assignment (assign_bags :226-261, FeatureTrackStatus :10-161) ------------------------------
    def assign(self) -> List[dict]:
        n = len(self.track_ids)
        queries = [list(q) for q in self.queries]          # consumed while bags are formed
        # tracks per image that still have query nodes to place: the reference rescans every track of an image for every
        # bag (:103-110); finished tracks are skipped there, so dropping them from the scan lists (order kept) changes
        # nothing but the cost -- 9 M skipped iterations for a 300-image / 190 K-track scene
        active = {img: list(ts) for img, ts in self.frame_dict.items()}
        stale = {img: 0 for img in active}
        length = np.array([len(q) + 1 for q in queries], dtype=np.int64)
        remaining = int(length.sum() - n)                   # query nodes not yet placed in a bag
        cap = self.max_track_length
        max_bag = 16                                        # FeatureTrackStatus default max_num_img_in_bag (:17)
        bags = []
        while remaining != 0:
            k = int(np.argmax(length))                      # first longest track (:55-58)
            ref = self.ref_img[k]
            if len(queries[k]) > cap - 1:                   # a long track is consumed cap-1 views at a time (:68-75)
                head = queries[k][:cap - 1]
                del queries[k][:cap - 1]
                remaining -= cap - 1
                length[k] -= cap - 1
            else:
                head = queries[k]
                queries[k] = []
                remaining -= int(length[k]) - 1
                length[k] = 1
                if ref in stale:
                    stale[ref] += 1
            bag_imgs = [ref] + head
            bag_set = set(bag_imgs)                         # == set(bag_imgs) at every point below: same insertion sequence
            tracks, corr = [self.track_ids[k]], [[ref, head]]
            # every other track whose reference node is one of the bag's images (:103-157); the bag may grow while
            # it is being scanned, exactly like the reference's loop over the list it appends to
            for img in bag_imgs:
                lst = active.get(img, ())
                if stale.get(img, 0) * 2 > len(lst):        # compact the scan list once half of it is finished tracks
                    lst = active[img] = [t for t in lst if t in self.track_pos and length[self.track_pos[t]] != 1]
                    stale[img] = 0
                for t in lst:
                    if t == self.track_ids[k] or t not in self.track_pos:
                        continue
                    j = self.track_pos[t]
                    if length[j] == 1:
                        continue
                    assert self.ref_img[j] == img
                    q = queries[j]
                    qs = set(q)
                    common = qs & bag_set
                    outside = qs - bag_set
                    quota = max_bag - len(bag_imgs)
                    if quota > 0 and len(outside) != 0:
                        extra = list(outside)[:quota]
                        bag_imgs += extra
                        for e in extra:
                            bag_set.add(e)
                        outside -= set(extra)
                        common |= set(extra)
                    if len(common) != 0:                    # (:138-141 with exclude_value = 0: always true otherwise)
                        assert len(set(common) - set(q)) == 0
                        queries[j] = list(set(q) - set(common))
                        remaining -= len(common)
                        length[j] -= len(common)
                        if length[j] == 1:
                            stale[img] += 1
                        tracks.append(t)
                        corr.append([img, list(common)])
            bags.append({"bag_image_ids": bag_imgs, "track_ids": tracks, "track_corresponding_imgs": corr})
        return bags

    def _chunk(self, bags):
        out = []
        for b in bags:                                      # chunk_bags :204-224
            if len(b["track_ids"]) > self.chunk:
                n_split = len(b["track_ids"]) // self.chunk + 1
                for tr, co in zip(chunks_round_robin(b["track_ids"], n_split),
                                  chunks_round_robin(b["track_corresponding_imgs"], n_split)):
                    out.append({"bag_image_ids": b["bag_image_ids"], "track_ids": tr, "track_corresponding_imgs": co})
            else:
                out.append(b)
        return out

    def __len__(self):
        return len(self.image_bags)

    # -- vectorised bag tensors (__getitem__ :317-476) ---------------------------------------------------------
    def _point_scale(self, img_rows, xyz):
        """f / (depth + 1e-4) of points xyz [n,3] seen from images img_rows [n] (get_point_scale :281-288)."""
        cam = np.einsum("nij,nj->ni", self._R[img_rows], xyz) + self._t[img_rows]
        depth = np.einsum("nj,nj->n", self._K[img_rows][:, 2, :], cam)
        return self._K[img_rows][:, 0, 0] / (depth + 1e-4)

    def _view_vector(self, src_rows, dst_rows, xyz):
        """get_relative_view_point :290-308 for n (source image, destination image, point) triples."""
        def T(rows):
            m = np.zeros((len(rows), 4, 4))
            m[:, :3, :3], m[:, :3, 3], m[:, 3, 3] = self._R[rows], self._t[rows], 1.0
            return m
        src, dst = T(src_rows), T(dst_rows)
        f = np.einsum("nij,nj->ni", src[:, :3, :3], xyz) + src[:, :3, 3]
        t = (src @ np.linalg.inv(dst))[:, :3, 3]
        a = f - t
        nf, nt, na = (np.linalg.norm(v, axis=-1) for v in (f, t, a))
        alpha = np.arccos(np.einsum("ni,ni->n", f, t) / (nf * nt + 1e-6))
        beta = np.arccos(np.einsum("ni,ni->n", a, -t) / (na * nt + 1e-6))
        return (t / (nt + 1e-6)[:, None]) * (np.pi - alpha - beta)[:, None]

    def bag_tensors(self, index: int, with_images: bool = True) -> Dict[str, torch.Tensor]:
        bag = self.image_bags[index]
        bag_imgs = bag["bag_image_ids"]
        img_slot = {img: k for k, img in enumerate(bag_imgs)}
        order = sorted(range(len(bag["track_ids"])), key=lambda k: len(bag["track_corresponding_imgs"][k][1]), reverse=True)
        tracks = [bag["track_ids"][k] for k in order]
        corr = [bag["track_corresponding_imgs"][k] for k in order]
        M, Nq = len(tracks), len(bag_imgs) - 1
        pos = np.array([self.track_pos[t] for t in tracks])
        ref_img = np.array([c[0] for c in corr])
        ref_kpt = np.array([self.ref_kpt[p] for p in pos])
        xyz = np.stack([self.colmap_3ds[t].xyz for t in tracks]).astype(np.float64)
        ref_rows = np.array([self._img_row[i] for i in ref_img])
        ref_xy = np.stack([self.colmap_images[i].xys[k] for i, k in zip(ref_img, ref_kpt)])
        ref_scale = self._point_scale(ref_rows, xyz)

        q_xy = np.ones((M, Nq, 2))                                  # padding values of the reference (:391-398)
        q_mask = np.zeros((M, Nq), dtype=bool)
        q_slot = np.full((M, Nq), -1, dtype=np.int64)
        q_img = np.full((M, Nq), -1, dtype=np.int64)
        q_kpt = np.full((M, Nq), -1, dtype=np.int64)
        q_scale = np.repeat(ref_scale[:, None], Nq, 1)
        q_view = np.zeros((M, Nq, 3))
        # flat list of the valid (track, view) slots, then one gather per field
        tt, vv, ii = [], [], []
        for m, c in enumerate(corr):
            tt += [m] * len(c[1])
            vv += list(range(len(c[1])))
            ii += list(c[1])
        if tt:
            tt, vv, ii = np.array(tt), np.array(vv), np.array(ii)
            kp = np.empty(len(tt), dtype=np.int64)
            xy = np.empty((len(tt), 2))
            obs = self._obs
            for n, (m, img) in enumerate(zip(tt.tolist(), ii.tolist())):
                kp[n], xy[n] = obs[(tracks[m], img)]
            rows = np.array([self._img_row[i] for i in ii])
            q_xy[tt, vv], q_mask[tt, vv] = xy, True
            q_slot[tt, vv] = [img_slot[i] for i in ii]
            q_img[tt, vv], q_kpt[tt, vv] = ii, kp
            q_scale[tt, vv] = self._point_scale(rows, xyz[tt])
            q_view[tt, vv] = self._view_vector(ref_rows[tt], rows, xyz[tt])

        data = {}
        if with_images:
            items = [self.dataset[self.dataset.colmapID2frameID_dict[i]] for i in bag_imgs]
            data["images"] = [it["image"] for it in items]
            if "scale" in items[0]:
                data["scales"] = torch.stack([it["scale"] for it in items], dim=0)
        scales_abs = torch.from_numpy(np.concatenate([ref_scale[:, None], q_scale], axis=-1))
        if "scales" in data:
            scales_abs = scales_abs / data["scales"][..., 0][None]
        scales_rel = scales_abs / scales_abs[..., [0]]
        view = torch.from_numpy(np.concatenate([np.zeros((M, 1, 3)), q_view], axis=-2))
        data.update({
            "query_points": torch.from_numpy(ref_xy).to(torch.float32) - 0.5,
            "reference_points_coarse": torch.from_numpy(q_xy).transpose(0, 1).to(torch.float32) - 0.5,
            "track_valid_mask": torch.from_numpy(q_mask).transpose(0, 1),
            "query_img_idxs": torch.from_numpy(np.array([img_slot[i] for i in ref_img])),
            "reference_img_idxs": torch.from_numpy(q_slot).transpose(0, 1),
            "scales_relative": scales_rel.transpose(0, 1),
            "view_point_vector": view.transpose(0, 1),
            "query_img_ids": torch.from_numpy(ref_img),
            "query_pt2d_idxs": torch.from_numpy(ref_kpt),
            "reference_img_ids": torch.from_numpy(q_img).transpose(0, 1),
            "reference_pt2d_idxs": torch.from_numpy(q_kpt).transpose(0, 1),
        })
        return data

    def __getitem__(self, index):
        return self.bag_tensors(index)


class DeviceUpdatedQueryPts:
    """``UpdatedQueryPts`` (multiview_match_worker.py:85-108) as two dense tensors on ``device``: the refined location
    of every (image, keypoint) that a previous bag has already moved, and a flag.  Bags of one long track run in
    sequence and share it; probing / filling is one gather / one scatter per bag instead of a Python loop per track.

    Reference quirk, kept by default (``reference_lookup=True``): ``find_movable_and_update`` iterates
    ``data['query_pt2d_idxs'][0]`` as 0-dim torch tensors (:93) and tests them with ``in`` against a dict whose keys
    are the numpy integers ``update_query_pts`` stored; ``torch.Tensor.__hash__`` is identity based, so the lookup
    never hits: in the reference every query point stays movable and keeps its coarse location, whatever earlier bags
    refined.  A drop-in must produce the same tracks, so the default reproduces exactly that (the table is still
    filled); ``reference_lookup=False`` gives the evident intent (moved points are pinned to their refined location)."""

    def __init__(self, colmap_images, device="cpu", reference_lookup=True):
        self.reference_lookup = reference_lookup
        self.img_ids = sorted(colmap_images.keys())
        counts = [int(np.asarray(colmap_images[i].xys).shape[0]) for i in self.img_ids]
        self.device = torch.device(device)
        base = np.zeros(max(self.img_ids) + 2, dtype=np.int64)
        base[np.array(self.img_ids)] = np.cumsum([0] + counts[:-1])
        self.base = torch.from_numpy(base).to(self.device)
        self.xy = torch.zeros((sum(counts), 2), dtype=torch.float32, device=self.device)
        self.moved = torch.zeros((sum(counts),), dtype=torch.bool, device=self.device)

    def _key(self, img_ids, kpt_idxs):
        return self.base[img_ids.to(self.device).long()] + kpt_idxs.to(self.device).long()

    def find_movable_and_update(self, data: dict):
        """Replace already-moved query points by their refined location and mark them immovable (:89-104)."""
        key = self._key(data["query_img_ids"][0], data["query_pt2d_idxs"][0])
        moved = torch.zeros_like(self.moved[key]) if self.reference_lookup else self.moved[key]
        pts = data["query_points"][0].to(device=self.device, dtype=torch.float32)
        data.update({"query_points": torch.where(moved[:, None], self.xy[key], pts)[None],
                     "query_movable_mask": (~moved)[None]})

    def update_query_pts(self, kpts_refined, image_ids, pt2d_idxs):
        """Record refined locations (:106-108); on a repeated key the LAST row wins, like the dict assignment."""
        as_t = (lambda x: x if isinstance(x, torch.Tensor) else torch.from_numpy(np.asarray(x)))
        key = self._key(as_t(image_ids), as_t(pt2d_idxs))
        if key.numel() == 0:
            return
        val = as_t(kpts_refined).to(device=self.device, dtype=torch.float32)
        order = torch.argsort(key, stable=True)
        ks = key[order]
        last = torch.ones_like(ks, dtype=torch.bool)
        last[:-1] = ks[1:] != ks[:-1]
        sel = order[last]
        self.xy[key[sel]] = val[sel]
        self.moved[key[sel]] = True
