"""Reference-shaped plugin builders (same names, argument meaning and error behaviour).

coarse   ``build_model(args) -> (detector, matcher)``, ``extract_preds``, ``extract_matches``
         mirror src/coarse_match/coarse_match_worker.py:21-99; selected with the NEW matcher names
         ``args['matcher'] == 'loftr_hip'`` / ``'matchformer_hip'`` / ``'aspanformer_hip'`` (``neuralsfm.NEUSFM_coarse_matcher``), so the
         reference's own 'loftr_official' / 'aspanformer' / 'matchformer' branches stay intact.
refine   ``build_refine_model(args, rewindow_size_factor, model_idx) -> matcher`` and
         ``extract_results`` mirror src/post_optimization/matcher_model/multiview_match_worker.py:16-82.

INTEGRATION.md shows the few lines a maintainer adds to the reference to route to these.
"""
import copy

import numpy as np
import torch
import torch.nn as nn

from . import ops
from .coarse import HipLoFTR
from .config import loftr_coarse_only_config, multiview_refinement_config
from .aspanformer import HipASpanFormer, aspanformer_coarse_only_config
from .matchformer import HipMatchformer, matchformer_coarse_only_config
from .refine import HipMultiviewMatcher


_COARSE_ONLY = "coarse_only"      # the literal of coarse_match_worker.py:134 (interned: identical to any literal 'coarse_only')


class DetectorWrapper(nn.Module):
    """No-op 'OnGrid' detector (src/coarse_match/utils/detector_wrapper.py:4-22)."""

    def __init__(self, detector=None, detector_type="OnGrid", fullcfg=None):
        super().__init__()
        if detector_type != "OnGrid" or detector is not None:
            raise NotImplementedError(detector_type)
        self.detector_type = detector_type

    @torch.no_grad()
    def forward(self, batch):
        return None


def build_model(args: dict):
    """args: {'matcher': 'loftr_hip', 'type': 'coarse_only', 'match_thr': float, 'seed': int,
    'loftr_hip': {'weight_path': path-or-None, 'cfg': optional lower-cased LoFTR config}}."""
    if "seed" in args:
        torch.manual_seed(args["seed"])
    if args["matcher"] == "matchformer_hip":            # the 'matchformer' branch of the reference (:62-76)
        if args.get("type", "coarse_only") != "coarse_only":
            raise NotImplementedError("matchformer_hip provides the coarse_only matcher")
        margs = args.get("matchformer_hip", {})
        cfg = margs.get("cfg") or matchformer_coarse_only_config(args["match_thr"])
        cfg["match_coarse"]["thr"] = args["match_thr"]
        matcher = HipMatchformer(config=cfg)
        if margs.get("weight_path") is not None:            # the MatchFormer checkpoints are bare state dicts (:72-73)
            matcher.load_state_dict(torch.load(margs["weight_path"], map_location="cpu"), strict=True)
        detector = DetectorWrapper()
        detector.eval()
        matcher.eval()
        return detector, matcher
    if args["matcher"] == "aspanformer_hip":            # the 'aspanformer' branch of the reference (:45-60)
        if args.get("type", "coarse_only") != "coarse_only":
            raise NotImplementedError("aspanformer_hip provides the coarse_only matcher")
        margs = args.get("aspanformer_hip", {})
        cfg = margs.get("cfg") or aspanformer_coarse_only_config(args["match_thr"])
        cfg["match_coarse"]["thr"] = args["match_thr"]
        matcher = HipASpanFormer(config=cfg, online_resize=True)
        if margs.get("weight_path") is not None:
            # strict=False like the reference (:56) -- but a drop-in must not silently run on half-loaded weights, so what
            # strict=False let through is checked: only the constant sampling pattern may be missing (it is not stored in
            # checkpoints), and unexpected keys are tolerated only outside the matcher (Lightning bookkeeping, loss buffers)
            res = matcher.load_state_dict(torch.load(margs["weight_path"], map_location="cpu")["state_dict"], strict=False)
            missing = [k for k in res.missing_keys if "sample_offset" not in k]
            unexpected = [k for k in res.unexpected_keys if k.startswith(("backbone.", "loftr_coarse.", "coarse_matching.",
                                                                             "pos_encoding.", "fine_", "loftr_fine."))]
            if missing or unexpected:
                raise RuntimeError(f"aspanformer_hip: checkpoint does not fit the matcher (missing {missing[:5]}"
                                   f"{'...' if len(missing) > 5 else ''}, unexpected {unexpected[:5]})")
        detector = DetectorWrapper()
        detector.eval()
        matcher.eval()
        return detector, matcher
    if args["matcher"] != "loftr_hip":
        raise NotImplementedError(args["matcher"])
    if args.get("type", "coarse_only") != "coarse_only":
        raise NotImplementedError("loftr_hip provides the coarse_only matcher")
    margs = args.get("loftr_hip", {})
    cfg = margs.get("cfg") or loftr_coarse_only_config(args["match_thr"])
    cfg["match_coarse"]["thr"] = args["match_thr"]
    cfg["coarse"]["temp_bug_fix"] = False
    matcher = HipLoFTR(config=cfg)
    weight_path = margs.get("weight_path")
    if weight_path is not None:
        state_dict = torch.load(weight_path, map_location="cpu")["state_dict"]
        matcher.load_state_dict(state_dict, strict=True)
    detector = DetectorWrapper()
    detector.eval()
    matcher.eval()
    return detector, matcher


def extract_preds(data):
    """extract predictions assuming bs==1 (coarse_match_worker.py:83-91)."""
    m_bids = data["m_bids"].cpu().numpy()
    assert (np.unique(m_bids) == 0).all()
    return data["mkpts0_f"].cpu().numpy(), data["mkpts1_f"].cpu().numpy(), data["mconf"].cpu().numpy()


@torch.no_grad()
def extract_matches(data, detector=None, matcher=None):
    detector(data)
    matcher(data)
    return extract_preds(data)


def match_table(data):
    """(M,5) rows [x0,y0,x1,y1,conf] as stored per pair (coarse_match_worker.py:139-141)."""
    mk0, mk1, mc = extract_preds(data)
    return np.concatenate([mk0, mk1, mc[:, None]], -1)


def build_refine_model(args: dict, rewindow_size_factor=None, model_idx=None):
    """args: {'cfg': optional model.multiview_refinement dict, 'weight_path': [path-or-None], 'seed': int}."""
    if "seed" in args:
        torch.manual_seed(args["seed"])
    cfg = multiview_refinement_config() if args.get("cfg") is None else copy.deepcopy(args["cfg"])
    if rewindow_size_factor is not None:     # window shrink per refinement iteration (:20-34)
        w = max(7, ((cfg["multiview_transform"]["window_size"] // 2) - rewindow_size_factor) * 2 + 1)
        cfg["backbone"]["s2dnet"]["window_size"] = w
        cfg["multiview_transform"]["window_size"] = w
        cfg["multiview_matching_test"]["window_size"] = w
        lw = cfg["multiview_matching_test"]["left_point_movement_window_size"]
        if lw is not None:
            cfg["multiview_matching_test"]["left_point_movement_window_size"] = max(
                3, ((lw // 2) - rewindow_size_factor) * 2 + 1)
    matcher = HipMultiviewMatcher(config=cfg, test=True).eval()
    paths = args.get("weight_path", [None])
    model_path = paths[model_idx] if model_idx is not None else paths[0]
    if model_path is not None:
        state_dict = torch.load(model_path, map_location="cpu")["state_dict"]
        for k in list(state_dict.keys()):          # multiview_match_worker.py:42-52
            if "matcher." in k:
                state_dict[k.replace("matcher.", "")] = state_dict.pop(k)
            else:
                state_dict.pop(k)
        for k in list(state_dict.keys()):
            if "loftr_coarse" in k:
                state_dict.pop(k)
            if "loftr_fine" in k:
                state_dict[k.replace("loftr_fine", "fine_transformer")] = state_dict.pop(k)
        matcher.load_state_dict(state_dict, strict=True)
    return matcher


@torch.no_grad()
def extract_results(data, matcher=None):
    """multiview_match_worker.py:59-82 (matcher=None: the forward pass has already been queued on ``data``)."""
    if matcher is not None:
        matcher(data)
    reference_points_refined = data["query_points_refined"].cpu().numpy()
    reference_img_ids = data["query_img_ids"].cpu().numpy()
    reference_pt2D_idxs = data["query_pt2d_idxs"].cpu().numpy()
    ref_movable_mask = data["query_movable_mask"].cpu().numpy()
    query_points_refined = data["reference_points_refined"][-1].cpu().numpy()
    query_img_ids = data["reference_img_ids"].cpu().numpy()
    query_pt2D_idxs = data["reference_pt2d_idxs"].cpu().numpy()
    mask = data["track_valid_mask"].cpu().numpy()
    assert query_points_refined.shape[0] == 1
    return ([query_points_refined[mask], query_img_ids[mask], query_pt2D_idxs[mask]],
            [reference_points_refined[ref_movable_mask], reference_img_ids[ref_movable_mask],
             reference_pt2D_idxs[ref_movable_mask]], data.get("time"))


def merge_match_tables(matches: dict, names: list, pair_name_split: str, device="cuda"):
    """Drop-in for the "Combine keypoints / Update matches / Post-processing keypoints" block of
    detector_free_coarse_matching (src/coarse_match/coarse_match.py:203-237: Match2Kpts + keypoint_worker +
    update_matches(merge=False) + transform_keypoints) on the device, one call for the whole scene.

    matches: {f"{name0}{split}{name1}": ndarray [N,5] (mkpts0, mkpts1, mconf)} as produced by match_worker;
    names: image list.  Returns (final_keypoints {name: [K,2] float32}, final_scores {name: [K] float32},
    updated_matches {pair: [N,2] int}) exactly like the reference's three dictionaries."""
    index = {n: i for i, n in enumerate(names)}
    keys = list(matches.keys())
    for k in keys:                                  # match_worker's tables are float32 (.cpu().numpy() of fp32 tensors, :139-141);
        if np.asarray(matches[k]).dtype == np.float64:   # float64 rows would be averaged differently by the reference
            raise TypeError(f"merge_match_tables: table {k!r} is float64; the device merge reproduces the reference on float32 tables")
    tabs = [np.asarray(matches[k], dtype=np.float32).reshape(-1, 5) for k in keys]
    lens = [t.shape[0] for t in tabs]
    if sum(lens) == 0:
        return ({n: np.empty((0, 2)) for n in names}, {n: np.empty((0,), np.float32) for n in names},
                {k: np.empty((0, 2)).astype(int) for k in keys})
    rows = torch.from_numpy(np.concatenate(tabs, 0)).to(device)
    pair_imgs = np.array([[index[a], index[b]] for a, b in (k.split(pair_name_split) for k in keys)], dtype=np.int32)
    rep = torch.repeat_interleave(torch.from_numpy(pair_imgs), torch.tensor(lens), dim=0).to(device)
    kpts, scores, offsets, ids = ops.merge_keypoints(rows, rep[:, 0], rep[:, 1], len(names))
    kpts, scores, offsets, ids = kpts.cpu().numpy(), scores.cpu().numpy(), offsets.cpu().numpy(), ids.cpu().numpy()
    final_keypoints, final_scores = {}, {}
    for i, n in enumerate(names):
        k = kpts[offsets[i]:offsets[i + 1]]
        final_keypoints[n] = k if len(k) else np.empty((0, 2))        # transform_keypoints' corner case (:264-266)
        final_scores[n] = scores[offsets[i]:offsets[i + 1]]
    updated, pos = {}, 0
    for k, n in zip(keys, lens):
        updated[k] = ids[pos:pos + n].astype(int)
        pos += n
    return final_keypoints, final_scores, updated


@torch.no_grad()
def match_scene_cached(matcher, images, pairs, batch=8, scales=None, to_host=True):
    """Exhaustive / covisible pair matching of one scene with the backbone evaluated ONCE per image.

    The reference's match_worker (src/coarse_match/coarse_match_worker.py:102-145) feeds every pair through
    detector+matcher, so an image that appears in k pairs pays for k backbone passes -- 55 % of the coarse step
    here.  Backbone tokens are a per-image quantity: they are computed once, kept on the device, and paired by
    index; positional encoding, transformer and matching run per pair exactly as in ``HipLoFTR.forward``.
    ``matcher``: HipLoFTR or HipASpanFormer (both expose ``image_tokens`` / ``match_tokens``; ASpanFormer's ResNet is per image
    too -- 1.5 of its 5.1 ms per pair -- and its ``match_tokens`` sends several pairs through the transformer per pass).  MatchFormer-LA's backbone interleaves cross attention between the two images of a pair,
    so it has no per-image part to cache.

    images: tensor [n_images,1,H,W] (same size); pairs: list of (i, j); scales: optional [n_images,2] (h, w scale).
    Only the images that occur in ``pairs`` are run through the backbone (a rank's shard of a scene).
    Returns {(i, j): [M,5]} rows (x0, y0, x1, y1, conf), the per-pair tables match_worker stores: numpy arrays, or
    device tensors with ``to_host=False`` (for the all-gather / device-side keypoint merge)."""
    dev = next(matcher.parameters()).device
    used = sorted({int(i) for p in pairs for i in p})
    slot = {img: k for k, img in enumerate(used)}
    toks, hw_c = [], None
    for lo in range(0, len(used), 2 * batch):
        idx = torch.tensor(used[lo:lo + 2 * batch])
        t, hw_c = matcher.image_tokens(images[idx].to(dev))
        toks.append(t)
    toks = torch.cat(toks, 0) if toks else None
    hw_i = tuple(images.shape[2:])
    out = {}
    defer = bool(getattr(matcher, "supports_defer", False))

    def collect(chunk, m):
        if defer:
            m = m.result()                        # the one host read of this batch -- after the NEXT batch has been launched
        rows = torch.cat([m["mkpts0_c"], m["mkpts1_c"], m["mconf"][:, None]], -1)
        counts = torch.bincount(m["b_ids"], minlength=len(chunk)).tolist()       # rows come in ascending b order
        parts = rows.split(counts)
        if to_host:
            parts = [p.cpu().numpy() for p in parts]
        for pr, tab in zip(chunk, parts):
            out[tuple(pr)] = tab

    waiting = None
    for lo in range(0, len(pairs), batch):
        chunk = pairs[lo:lo + batch]
        i0 = torch.tensor([slot[int(p[0])] for p in chunk], device=dev)
        i1 = torch.tensor([slot[int(p[1])] for p in chunk], device=dev)
        s0 = None if scales is None else scales[[int(p[0]) for p in chunk]]
        s1 = None if scales is None else scales[[int(p[1]) for p in chunk]]
        if defer:       # software pipeline of depth one: the device works on batch k + 1 while the host reads batch k's tables
            m = matcher.match_tokens(toks[i0], toks[i1], hw_c, hw_c, hw_i, s0, s1, defer=True)
            if waiting is not None:
                collect(*waiting)
            waiting = (chunk, m)
        else:
            collect(chunk, matcher.match_tokens(toks[i0], toks[i1], hw_c, hw_c, hw_i, s0, s1))
    if waiting is not None:
        collect(*waiting)
    return out


@torch.no_grad()
def match_scene_sharded(matcher: HipLoFTR, images, names, pair_name_split=" ", batch=8, scales=None, pairs=None, group=None,
                        root=None, shard="tiles"):
    """One scene on all ranks of the process group -- the analogue of the reference's Ray fan-out + merge
    (src/coarse_match/coarse_match.py:127-140, 203-237): every rank matches its shard of the pair list
    (``match_scene_cached``: backbone once per image), the match tables are collected with ONE payload collective
    (``dist.collect_tables``: RCCL over xGMI, gloo in the CPU tests), and the keypoint merge runs on the device.

    root=None (default): every rank receives the tables, merges and returns the dictionaries -- the contract of rounds 1-2.
    root=r (opt-in, a rank of ``group``): gather-to-root -- like the reference, whose driver process alone merges and writes
    the h5 files (coarse_match.py:203-254), only rank r receives the tables (an exact-size buffer) and runs the merge; the
    other ranks return None.
    shard="tiles" (default): ``dist.shard_pairs_tiled`` -- blocks of the (i, j) plane, so that a rank runs the backbone on
    ~ n/sqrt(world) images instead of nearly all of them; "contiguous": ``dist.shard_range`` of the pair list.  The result
    does not depend on the sharding (tables are put back in pair order).

    images [n_images,1,H,W]; names: image names in the same order; pairs: list of (i, j) (default: exhaustive, in the
    order of src/construct_pairs/pairs_exhaustive.py).  Returns the reference's dictionaries
    (matches {"name0<split>name1": [M,5]}, final_keypoints, final_scores, updated_matches)."""
    from . import dist as ddist
    pairs = ddist.exhaustive_pairs(len(names)) if pairs is None else [tuple(p) for p in pairs]
    world, rank = _world_rank(group)
    order = _pair_shards(pairs, len(names), world, shard)
    mine = match_scene_cached(matcher, images, [pairs[k] for k in order[rank]], batch=batch, scales=scales, to_host=False)
    tables = ddist.collect_tables([mine[pairs[k]] for k in order[rank]], group=group, root=root)
    if tables is None:
        return None
    assert len(tables) == len(pairs)
    flat = [k for o in order for k in o]                                       # tables arrive in rank order
    matches = {}
    by_pair = dict(zip(flat, tables))
    for k, (i, j) in enumerate(pairs):                                          # back to pair order
        matches[f"{names[i]}{pair_name_split}{names[j]}"] = by_pair[k].cpu().numpy()
    kp, sc, upd = merge_match_tables(matches, names, pair_name_split, device=next(matcher.parameters()).device)
    return matches, kp, sc, upd


def _world_rank(group):
    import torch.distributed as tdist
    world = tdist.get_world_size(group) if tdist.is_available() and tdist.is_initialized() else 1
    return world, (tdist.get_rank(group) if world > 1 else 0)


def _pair_shards(pairs, n_images, world, shard):
    """Per-rank index lists into ``pairs`` (identical on every rank)."""
    from . import dist as ddist
    if shard == "tiles":
        return ddist.shard_pairs_tiled(pairs, n_images, world)
    if shard == "contiguous":
        return [list(range(*ddist.shard_range(len(pairs), r, world))) for r in range(world)]
    raise ValueError(f"shard must be 'tiles' or 'contiguous', got {shard!r}")


# dataset rules of detector_free_coarse_matching per matcher (src/coarse_match/coarse_match.py:82-90)
_DATA_RULES = {"loftr_hip": {"df": 8, "pad_to": None}, "matchformer_hip": {"df": 8, "pad_to": -1},
               "aspanformer_hip": {"df": None, "pad_to": None}}


@torch.no_grad()
def match_worker(subset_ids, image_lists, covis_pairs_out, cfgs, device="cuda", frames=None, models=None):
    """``match_worker`` (src/coarse_match/coarse_match_worker.py:101-145) for any of the three HIP matchers, with
    ``CoarseMatchingDataset`` (src/dataset/coarse_matching_dataset.py:10-100) folded in: every frame of the subset is read
    once (the reference's ``img_preload``), resized / padded / converted on the device (``images.read_grayscale``, rules of
    coarse_match.py:82-90), then every pair goes through ``extract_matches``.  ``cfgs`` is the reference's dictionary
    (``cfgs['matcher']['model']`` -> ``build_model``, ``cfgs['data']['img_resize']``, ``cfgs['matcher']['pair_name_split']``).
    covis_pairs_out: list of "path0 path1" strings or a file of such lines.  ``frames``: optional {path: decoded uint8
    array} to skip the file decode.  Returns {path0<split>path1: ndarray [N,5]} like the reference.
    Grid rounding (:133-136): the reference's guard is ``args['model']['type'] is not 'coarse_only'`` -- a string IDENTITY
    test.  It is False for a type string that is the interned literal (a Python-level config) and True for an equal string
    that comes out of YAML / hydra, in which case the reference rounds even in coarse_only mode whenever
    ``round_matches_ratio`` is set.  The same identity test and the same rounding expression are used here, so a config
    behaves as it does in the reference either way (the shipped coarse_only configs set the ratio to null)."""
    from . import images
    margs = cfgs["matcher"]["model"]
    ratio = cfgs["matcher"].get("round_matches_ratio")
    detector, matcher = models if models is not None else build_model(margs)
    matcher.to(device)
    rule = dict(_DATA_RULES[margs["matcher"]])
    resize = cfgs["data"].get("img_resize")
    if isinstance(covis_pairs_out, (list, tuple)):
        pair_list = list(covis_pairs_out)
    else:
        with open(covis_pairs_out, "r") as f:
            pair_list = f.read().rstrip("\n").split("\n")
    split = cfgs["matcher"].get("pair_name_split", " ")
    cache, matches = {}, {}

    def read(path):
        if path not in cache:
            src = frames[path] if frames is not None else path
            cache[path] = images.read_grayscale(src, (resize,) if resize is not None else None, df=rule["df"],
                                                pad_to=rule["pad_to"], ret_scales=True, device=device)
        return cache[path]
    for pair_idx in subset_ids:
        p0, p1 = pair_list[pair_idx].split(" ")
        (img0, scale0, _), (img1, scale1, _) = read(p0), read(p1)
        data = {"image0": img0[None], "image1": img1[None], "scale0": scale0[None].to(device), "scale1": scale1[None].to(device),
                "pair_key": ([p0], [p1]), "frameID": pair_idx}
        mkpts0, mkpts1, mconfs = extract_matches(data, detector=detector, matcher=matcher)
        if margs.get("type", _COARSE_ONLY) is not _COARSE_ONLY and ratio is not None:      # identity, as in the reference
            sc0, sc1 = scale0[None].cpu().numpy()[:, [1, 0]], scale1[None].cpu().numpy()[:, [1, 0]]
            mkpts0 = np.round((mkpts0 / sc0) / ratio) * ratio * sc0
            mkpts1 = np.round((mkpts1 / sc1) / ratio) * ratio * sc1
        matches[split.join([p0, p1])] = np.concatenate([mkpts0, mkpts1, mconfs[:, None]], -1)
    return matches


@torch.no_grad()
def match_worker_sharded(image_lists, covis_pairs_out, cfgs, device="cuda", frames=None, models=None, group=None, root=None,
                         shard="tiles"):
    """The reference's Ray fan-out of ``match_worker`` over chunks of the pair list (coarse_match.py:127-140) as one process
    per GPU: rank r matches its shard (``shard="tiles"``: blocks of the (image, image) plane, so a rank reads and resizes
    few frames; "contiguous": a slice of the list), ONE payload collective (``dist.collect_tables``) hands every rank
    (root=None, default) or only rank ``root`` of the group (the others then return None) the scene's table dictionary in
    pair order."""
    from . import dist as ddist
    if isinstance(covis_pairs_out, (list, tuple)):
        pair_list = list(covis_pairs_out)
    else:
        with open(covis_pairs_out, "r") as f:
            pair_list = f.read().rstrip("\n").split("\n")
    world, rank = _world_rank(group)
    paths = {}
    for p in pair_list:
        for q in p.split(" "):
            paths.setdefault(q, len(paths))
    order = _pair_shards([tuple(paths[q] for q in p.split(" ")) for p in pair_list], len(paths), world, shard)
    mine = match_worker(order[rank], image_lists, pair_list, cfgs, device=device, frames=frames, models=models)
    split = cfgs["matcher"].get("pair_name_split", " ")
    keys = [split.join(p.split(" ")) for p in pair_list]
    tables = ddist.collect_tables([torch.from_numpy(mine[keys[k]]).to(torch.float32).to(device) for k in order[rank]],
                                  group=group, root=root)
    if tables is None:
        return None
    assert len(tables) == len(keys)
    by_pair = dict(zip([k for o in order for k in o], tables))
    return {keys[k]: by_pair[k].cpu().numpy() for k in range(len(keys))}


def _batched(bag: dict, device):
    """What the reference's DataLoader(batch_size=1) + dict_to_cuda hand to the matcher: a leading batch dimension on
    every tensor (multiview_match_worker.py:115,126)."""
    out = {}
    for k, v in bag.items():
        if isinstance(v, list):
            out[k] = [t[None].to(device) for t in v]
        elif isinstance(v, torch.Tensor):
            out[k] = v[None].to(device)
        else:
            out[k] = v
    return out


@torch.no_grad()
def match_tracks_worker(colmap_dataset, matcher, subset_track_idxs=None, dataset_cfgs=None, device="cuda",
                        reference_lookup=True):
    """``matchWorker`` (src/post_optimization/matcher_model/multiview_match_worker.py:111-141) on the device: bags from
    ``BagPlanner`` (this rank's track subset), already-refined query points in ``DeviceUpdatedQueryPts`` (no per-track
    Python loop between forward passes), one ``extract_results`` per bag.  Returns the reference's ``results_list``:
    one float64 ndarray [M,4] per bag, rows [x, y, image id, keypoint index] (queries of the valid slots first, then the
    movable reference nodes)."""
    from .bags import BagPlanner, DeviceUpdatedQueryPts
    planner = BagPlanner(colmap_dataset, dataset_cfgs, worker_split_idxs=subset_track_idxs)
    matcher.to(device)
    buf = DeviceUpdatedQueryPts(planner.colmap_images, device=device, reference_lookup=reference_lookup)
    results = []
    n = len(planner)
    nxt = planner.bag_tensors(0) if n else None
    for k in range(n):
        data = _batched(nxt, device)
        buf.find_movable_and_update(data)
        matcher(data)                       # kernels are queued; the device works on bag k ...
        # ... while the host builds the tensors of bag k + 1 (tools/bench_bags.py: the planner delivers 34 K tracks/s on one
        # core against 53 K tracks/s on the device, so it must not sit in front of the forward pass)
        nxt = planner.bag_tensors(k + 1) if k + 1 < n else None
        (q_pts, q_ids, q_idx), (r_pts, r_ids, r_idx), _ = extract_results(data, matcher=None)
        mov = data["query_movable_mask"]
        buf.update_query_pts(data["query_points_refined"][mov], data["query_img_ids"][mov], data["query_pt2d_idxs"][mov])
        pts = np.concatenate([q_pts, r_pts], axis=0)
        ids = np.concatenate([q_ids, r_ids], axis=0)
        idx = np.concatenate([q_idx, r_idx], axis=0)
        results.append(np.concatenate([pts, ids[:, None], idx[:, None]], axis=1))
    return results


@torch.no_grad()
def refine_scene_sharded(matcher, colmap_dataset, dataset_cfgs, seed=None, device="cuda", group=None, root=None):
    """One scene's feature tracks on all ranks of the process group -- the analogue of ``multiview_matcher`` with Ray
    (src/post_optimization/matcher_model/multiview_match.py:39-62): tracks are dealt to the ranks by index
    (``dist.shard_tracks``: all bags of one track on one rank), every rank runs ``match_tracks_worker`` on its subset,
    and the [M,4] result rows are collected with ONE payload collective of variable-length tables (RCCL over xGMI / gloo)
    as 16-byte rows: the two coordinates are fp32 values already (``.cpu().numpy()`` of fp32 tensors, widened to float64 only
    by the reference's ``np.concatenate`` with the integer columns, multiview_match_worker.py:136-139), so they travel as
    their fp32 bits, and the image id / keypoint index as int32.  Returns the concatenated list of per-bag float64 arrays
    (rank order, then bag order) on every rank (root=None, the default) or only on rank ``root`` of the group (opt-in
    gather-to-root; None elsewhere).  A rank whose rows do not fit the 16-byte form does not raise alone: it sends a
    one-column table, which makes ``collect_tables`` raise on every rank."""
    from . import dist as ddist
    world, rank = _world_rank(group)
    n = len(colmap_dataset.point_cloud_assigned_imgID_kptID)
    mine = match_tracks_worker(colmap_dataset, matcher, ddist.shard_tracks(n, rank, world, seed), dataset_cfgs, device)
    if world == 1:
        return mine
    dev = torch.device(device)
    words, bad = [], False
    for a in mine:
        xy = a[:, :2].astype(np.float32)
        ids = a[:, 2:]
        # NaN coordinates are legitimate fp32 values (equal_nan); the ids must be integers that fit int32
        if not (np.array_equal(xy.astype(np.float64), a[:, :2], equal_nan=True) and np.array_equal(np.rint(ids), ids) and
                (np.abs(ids) < 2 ** 31).all()):
            bad = True
            break
        words.append(torch.from_numpy(np.concatenate([xy.view(np.int32), ids.astype(np.int32)], 1)).to(dev))
    if bad:     # raise TOGETHER with the other ranks (a lone raise would leave them inside the collective)
        words = [torch.zeros((1, 1), dtype=torch.float64, device=dev)]
    try:
        tabs = ddist.collect_tables(words, group=group, root=root, dtype=torch.int32)
    except TypeError as e:
        raise ValueError("refine_scene_sharded: result rows are not (fp32 x, fp32 y, int32 image id, int32 keypoint index) "
                         f"on some rank ({e})") from e
    if tabs is None:
        return None
    out = []
    for t in tabs:
        w = t.cpu().numpy()
        out.append(np.concatenate([w[:, :2].copy().view(np.float32).astype(np.float64), w[:, 2:].astype(np.float64)], 1))
    return out
