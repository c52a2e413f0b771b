"""Seeded synthetic inputs for the two hot-path halves.

No datasets or checkpoints are reachable offline, so benchmarks and parity tests use synthetic
frames of the reference's shapes and seeded random weights (``params.random_state_dict``).
Everything is generated on the CPU with explicit generators -> identical on every machine.
"""
import torch


def coarse_pair_batch(n_pairs: int, H: int = 480, W: int = 640, seed: int = 1000, shift=(1, 2)):
    """Config 2 input: image0 ~ U[0,1); image1 = image0 rolled by (8*dy, 8*dx) px + 0.02 N(0,1).
    Returns the plugin's ``data`` dict (image0/1 [N,1,H,W], scale0/1 [N,2])."""
    im0, im1 = [], []
    for p in range(n_pairs):
        g = torch.Generator().manual_seed(seed + p)
        a = torch.rand((1, 1, H, W), generator=g)
        b = torch.roll(a, shifts=(8 * shift[0], 8 * shift[1]), dims=(2, 3)) + 0.02 * torch.randn((1, 1, H, W), generator=g)
        im0.append(a)
        im1.append(b)
    return {"image0": torch.cat(im0), "image1": torch.cat(im1),
            "scale0": torch.ones(n_pairs, 2), "scale1": torch.ones(n_pairs, 2)}


def coarse_pair_padded(n_pairs: int, H: int = 96, W: int = 128, seed: int = 1000, shift=(1, 2), valid=None):
    """``coarse_pair_batch`` frames as a ``pad_to`` dataset hands them over (src/dataset/utils.py:104-121: the frame sits in
    the top-left corner of a zero canvas; the mask marks it) plus ``mask0`` / ``mask1`` [N, H/8, W/8] bool at the coarse
    resolution -- the optional inputs of LoFTR.forward (loftr.py:35-36, 61-63).  ``valid``: per pair ((h0, w0), (h1, w1)) in
    coarse cells; default: a different rectangle for each of the first pairs, the rest full frames."""
    data = coarse_pair_batch(n_pairs, H, W, seed, shift)
    hc, wc = H // 8, W // 8
    default = [((hc - 3, wc - 2), (hc - 1, wc)), ((hc, wc - 4), (hc - 2, wc - 1))]
    m0 = torch.zeros((n_pairs, hc, wc), dtype=torch.bool)
    m1 = torch.zeros((n_pairs, hc, wc), dtype=torch.bool)
    for p in range(n_pairs):
        (h0, w0), (h1, w1) = (valid[p] if valid is not None else default[p] if p < len(default) else ((hc, wc), (hc, wc)))
        m0[p, :h0, :w0] = True
        m1[p, :h1, :w1] = True
        data["image0"][p, :, 8 * h0:] = 0
        data["image0"][p, :, :, 8 * w0:] = 0
        data["image1"][p, :, 8 * h1:] = 0
        data["image1"][p, :, :, 8 * w1:] = 0
    data["mask0"], data["mask1"] = m0, m1
    return data


def coarse_pair_two_sizes(H0: int = 96, W0: int = 128, H1: int = 80, W1: int = 112, seed: int = 1000, shift=(1, 2)):
    """A pair whose two frames differ in size (LoFTR.forward's two-backbone-call branch, loftr.py:45-49):
    image1 = the (H1, W1) window of image0 that starts ``shift`` coarse cells in, + 0.02 N(0,1)."""
    g = torch.Generator().manual_seed(seed)
    a = torch.rand((1, 1, H0, W0), generator=g)
    y0, x0 = 8 * shift[0], 8 * shift[1]
    b = a[:, :, y0:y0 + H1, x0:x0 + W1] + 0.02 * torch.randn((1, 1, H1, W1), generator=g)
    return {"image0": a, "image1": b.contiguous(), "scale0": torch.ones(1, 2), "scale1": torch.ones(1, 2)}


def correlated_features(N: int, L: int, S: int, C: int = 256, seed: int = 7, noise: float = 0.1):
    """Kernel-level K3/K4 input: f0 ~ N(0,1), f1 = f0[perm] + noise*N(0,1) (random weights give
    ~0 matches at thr 0.2, so matching tests feed correlated features directly)."""
    g = torch.Generator().manual_seed(seed)
    f0 = torch.randn((N, L, C), generator=g)
    f1 = torch.empty((N, S, C))
    for n in range(N):
        perm = torch.randperm(max(L, S), generator=g)[:S] % L
        f1[n] = f0[n, perm] + noise * torch.randn((S, C), generator=g)
    return f0, f1


def refine_bag(T: int = 2000, V: int = 5, H: int = 480, W: int = 640, seed: int = 2000,
               variable_lengths: bool = False, scales=None):
    """Config 3 input: V RGB images U[0,1), T tracks; query ~ U([30,W-30]x[30,H-30]),
    reference = query + N(0, 2 px); view v of every track lives in image v.
    ``variable_lengths`` draws track lengths from {2..V} (sorted descending, padded with -1)."""
    g = torch.Generator().manual_seed(seed)
    images = [torch.rand((1, 3, H, W), generator=torch.Generator().manual_seed(seed + v)) for v in range(V)]
    q = torch.stack([torch.rand(T, generator=g) * (W - 60) + 30, torch.rand(T, generator=g) * (H - 60) + 30], -1)
    ref = q[None] + 2.0 * torch.randn((V - 1, T, 2), generator=g)
    if variable_lengths:
        lens = torch.randint(2, V + 1, (T,), generator=g).sort(descending=True)[0]
    else:
        lens = torch.full((T,), V)
    valid = torch.arange(1, V)[:, None] < lens[None]                                  # [V-1,T]
    ref_idx = torch.where(valid, torch.arange(1, V)[:, None].expand(V - 1, T), torch.full((V - 1, T), -1))
    data = {
        "images": images,
        "scales": (torch.ones(1, V, 2) if scales is None else scales),
        "query_points": q[None].float(),
        "reference_points_coarse": ref[None].float(),
        "query_img_idxs": torch.zeros(1, T, dtype=torch.long),
        "reference_img_idxs": ref_idx[None],
        "track_valid_mask": valid[None],
        "scales_relative": torch.ones(1, V, T),
        "view_point_vector": torch.zeros(1, V, T, 3),
        "query_movable_mask": torch.ones(1, T, dtype=torch.bool),
    }
    return data


class _Obj:
    def __init__(self, **kw):
        self.__dict__.update(kw)


class SyntheticSfMScene:
    """A COLMAP-shaped scene with the attributes ``MatchingMultiviewData`` reads from ``colmap_image_dataset``
    (src/dataset/coarse_sfm_refinement_dataset.py:76-115, 361-390): cameras on a ring around a point cloud, every 3D
    point observed by 2..max_views images (a few of them twice in one image, as COLMAP tracks can be), keypoints =
    noisy projections, the reference node of a track chosen by the 'middle' point-scale rule (:236-297)."""

    def __init__(self, n_images=10, n_points=200, seed=0, hw=(96, 128), max_views=12, with_scale=True):
        import numpy as np
        rng = np.random.default_rng(seed)
        H, W = hw
        f = 0.9 * W
        self.colmap_images, self.colmap_3ds, self.image_intrin_extrins = {}, {}, {}
        ids = list(range(1, n_images + 1))
        for i in ids:
            ang = 2 * np.pi * (i - 1) / n_images + 0.1 * rng.standard_normal()
            c = np.array([3.0 * np.cos(ang), 0.3 * rng.standard_normal(), 3.0 * np.sin(ang)])
            z = -c / np.linalg.norm(c)
            x = np.cross(np.array([0.0, 1.0, 0.0]), z)
            x /= np.linalg.norm(x)
            R = np.stack([x, np.cross(z, x), z])
            self.image_intrin_extrins[i] = {"intrin": np.array([[f, 0, W / 2], [0, f, H / 2], [0, 0, 1.0]]),
                                            "extrin": [R, -R @ c]}
        kpts = {i: [] for i in ids}
        obs_p3d = {i: [] for i in ids}
        self.point_cloud_assigned_imgID_kptID = {}
        for pid in range(1, n_points + 1):
            xyz = 0.6 * rng.standard_normal(3)
            nv = int(rng.integers(2, min(max_views, n_images) + 1))
            seen = [int(v) for v in rng.choice(ids, size=nv, replace=False)]
            if rng.random() < 0.1:
                seen.append(seen[int(rng.integers(0, len(seen)))])      # the same image twice in one track
            image_ids, p2d, scales = [], [], []
            for i in seen:
                K, (R, t) = self.image_intrin_extrins[i]["intrin"], self.image_intrin_extrins[i]["extrin"]
                cam = R @ xyz + t
                uv = (K @ cam)[:2] / (cam[2] + 1e-4) + rng.standard_normal(2)
                image_ids.append(i)
                p2d.append(len(kpts[i]))
                kpts[i].append(uv)
                obs_p3d[i].append(pid)
                scales.append(K[0, 0] / (cam[2] + 1e-4))
            self.colmap_3ds[pid] = _Obj(xyz=xyz, image_ids=np.array(image_ids, dtype=np.int32),
                                        point2D_idxs=np.array(p2d, dtype=np.int32))
            order = np.argsort(np.array(scales))
            a = int(order[len(order) // 2])
            self.point_cloud_assigned_imgID_kptID[pid] = (image_ids[a], p2d[a])
        self.keyframe_dict = {}
        for i in ids:
            self.colmap_images[i] = _Obj(xys=np.array(kpts[i], dtype=np.float64).reshape(-1, 2),
                                         point3D_ids=np.array(obs_p3d[i], dtype=np.int64))
            state = np.full(len(kpts[i]), -3, dtype=np.int64)
            for pid, (img, k) in self.point_cloud_assigned_imgID_kptID.items():
                if img == i:
                    state[k] = pid
            self.keyframe_dict[i] = state[state >= 0].astype(np.int32)
        self.colmapID2frameID_dict = {i: i - 1 for i in ids}
        self.colmap_cameras = {}                                   # read by the reference's constructor, unused
        g = torch.Generator().manual_seed(seed)
        self._items = [{"image": torch.rand((3, H, W), generator=g),
                        **({"scale": torch.tensor([1.0 + 0.25 * (k % 3), 1.0 + 0.5 * (k % 2)])} if with_scale else {})}
                       for k in range(n_images)]

    def __len__(self):
        return len(self._items)

    def __getitem__(self, frame_id):
        return self._items[frame_id]


def to_device(data: dict, device):
    out = {}
    for k, v in data.items():
        if isinstance(v, torch.Tensor):
            out[k] = v.to(device)
        elif isinstance(v, list) and v and isinstance(v[0], torch.Tensor):
            out[k] = [t.to(device) for t in v]
        else:
            out[k] = v
    return out
