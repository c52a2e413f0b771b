"""TEST INFRASTRUCTURE ONLY -- CPU restatement (oracle) of the match-table consumer stage.

Only ``tests/`` may import this file, as the checker.  The product never imports it.

What it restates (paths relative to /root/reference), in plain numpy on flat arrays:

  Match2Kpts.__getitem__        src/coarse_match/utils/merge_kpts.py:36-61   per image: the [x, y, conf] rows of every
                                                                              pair it appears in, pairs in table order
  agg_groupby_2d(agg="sum")     src/coarse_match/utils/merge_kpts.py:4-17    np.unique(axis=0) + np.bincount(weights)
  keypoint_worker               src/coarse_match/coarse_match_worker.py:151-175  int-truncated keypoints, summed
                                                                              confidence, ids by descending score
                                                                              (stable: ties keep (x, y) order)
  update_matches(merge=False)   src/coarse_match/coarse_match_worker.py:182-243  matches -> [id0, id1]
  transform_keypoints           src/coarse_match/coarse_match_worker.py:250-270  float32 keypoints / scores in id order

Pinned: ``oracle/make_golden.py`` runs the reference's own functions (``oracle.ref_import.
import_match_table_consumers``: their unchanged source) on seeded match tables and commits inputs + outputs as
``tests/golden/merge_keypoints.npz``; ``tests/test_oracle_golden.py`` checks this restatement bit-for-bit against
them, and live against the reference when /root/reference exists.
"""
import numpy as np


def merge_keypoints(rows, img0, img1, n_images):
    """rows [M,5] float32 = (x0, y0, x1, y1, conf) of every match of the scene, pairs concatenated in table order;
    img0/img1 [M] image index of each side.
    Returns (kpts [K,2] float32, scores [K] float32, offsets [n_images+1] int64, match_ids [M,2] int64): image i owns
    keypoints offsets[i]:offsets[i+1], listed in id order; match_ids[m] = (id of side 0 in img0[m], id of side 1 in img1[m])."""
    rows = np.asarray(rows, dtype=np.float32).reshape(-1, 5)
    M = rows.shape[0]
    kp_all, sc_all, offsets = [], [], [0]
    ids = np.zeros((M, 2), dtype=np.int64)
    for im in range(n_images):
        # Match2Kpts: rows of this image in pair order; side 0 and side 1 entries of a pair keep row order.
        # (an image never meets itself in a pair, so per pair exactly one side contributes)
        sel0, sel1 = np.nonzero(img0 == im)[0], np.nonzero(img1 == im)[0]
        ent = np.concatenate([np.stack([sel0, np.zeros_like(sel0)], 1), np.stack([sel1, np.ones_like(sel1)], 1)], 0)
        ent = ent[np.argsort(ent[:, 0], kind="stable")]
        if ent.shape[0] == 0:
            offsets.append(offsets[-1])
            continue
        xy = np.stack([rows[ent[:, 0], 2 * ent[:, 1]], rows[ent[:, 0], 2 * ent[:, 1] + 1]], 1).astype(int)
        conf = rows[ent[:, 0], 4]
        uniq, group = np.unique(xy, axis=0, return_inverse=True)          # lexicographic (x, y)
        group = group.reshape(-1)
        sums = np.bincount(group, weights=conf)                            # float64, input order
        order = sorted(range(len(sums)), key=lambda g: sums[g], reverse=True)   # stable, like sorted(dict.items())
        rank = np.empty(len(sums), dtype=np.int64)
        rank[np.asarray(order, dtype=np.int64)] = np.arange(len(sums))
        kp_all.append(uniq[order].astype(np.float32))
        sc_all.append(sums[order].astype(np.float32))
        ids[ent[:, 0], ent[:, 1]] = rank[group]
        offsets.append(offsets[-1] + len(sums))
    kpts = np.concatenate(kp_all, 0) if kp_all else np.zeros((0, 2), np.float32)
    scores = np.concatenate(sc_all, 0) if sc_all else np.zeros((0,), np.float32)
    return kpts, scores, np.asarray(offsets, dtype=np.int64), ids


def tables_to_flat(matches, names, split):
    """{pair name: [N,5]} (insertion order) -> flat (rows, img0, img1, pair_slices) for ``merge_keypoints``."""
    index = {n: i for i, n in enumerate(names)}
    rows, i0, i1, slices, pos = [], [], [], {}, 0
    for k, v in matches.items():
        a, b = k.split(split)
        v = np.asarray(v, dtype=np.float32).reshape(-1, 5)
        rows.append(v)
        i0.append(np.full(len(v), index[a], dtype=np.int32))
        i1.append(np.full(len(v), index[b], dtype=np.int32))
        slices[k] = (pos, pos + len(v))
        pos += len(v)
    cat = (lambda xs, d, w: np.concatenate(xs, 0) if xs else np.zeros(w, d))
    return cat(rows, np.float32, (0, 5)), cat(i0, np.int32, (0,)), cat(i1, np.int32, (0,)), slices


def synthetic_scene(n_images=6, n_pairs=9, seed=0, max_matches=400, grid=8, hw=(480, 640)):
    """Seeded match tables shaped like the coarse matcher's output: keypoints on the 1/8 grid times a per-image
    non-integer scale (so the int truncation matters), repeated keypoints across pairs, tied scores."""
    rng = np.random.RandomState(seed)
    names = [f"scene/img_{i:03d}.jpg" for i in range(n_images)]
    scales = 1.0 + rng.rand(n_images, 2) * 0.37
    pairs = [(a, b) for a in range(n_images) for b in range(a + 1, n_images)]
    rng.shuffle(pairs)
    matches = {}
    for a, b in pairs[:n_pairs]:
        n = int(rng.randint(0, max_matches))
        def pts(im):
            g = np.stack([rng.randint(0, hw[1] // grid, n), rng.randint(0, hw[0] // grid, n)], 1) * grid
            return (g * scales[im]).astype(np.float32)
        conf = rng.rand(n).astype(np.float32)
        conf[rng.rand(n) < 0.2] = np.float32(0.5)                    # exact ties
        matches[f"{names[a]} {names[b]}"] = np.concatenate([pts(a), pts(b), conf[:, None]], 1).astype(np.float32)
    return matches, names, " "
