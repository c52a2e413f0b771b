"""One-process-per-GPU sharding of the two embarrassingly parallel work lists and the single
collective of the path: an all-gather of variable-length match tables.

The reference fans pairs / tracks out over Ray tasks and merges pickled numpy results through the
object store (src/coarse_match/coarse_match.py:127-140, src/post_optimization/matcher_model/
multiview_match.py:39-62).  Here every rank takes a static contiguous shard (no data-path
collective), and the tables are collected with ``torch.distributed`` all-gathers (table count, row
counts, then one padded payload) -- RCCL over xGMI with backend "nccl" on MI355X, gloo on CPU for the tests.
"""
from typing import List, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_range(n_items: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous shard [lo, hi) of a work list; sizes differ by at most one."""
    base, rem = divmod(n_items, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_list(items: Sequence, rank: int, world_size: int):
    lo, hi = shard_range(len(items), rank, world_size)
    return items[lo:hi]


def shard_tracks(n_tracks: int, rank: int, world_size: int, seed=None) -> List[int]:
    """Indices (into ``point_cloud_assigned_imgID_kptID``) of the feature tracks one rank refines -- the analogue of
    ``chunk_index_balance(len(...), n_workers, shuffle=True)`` (src/utils/ray_utils.py:122-131, called at
    src/post_optimization/matcher_model/multiview_match.py:39-43): an (optionally seeded-shuffled) index array dealt
    round-robin.  The unit is the TRACK: each rank builds its bags from its own subset (``BagPlanner(...,
    worker_split_idxs=...)``), so the bags a long track is split into -- which depend on one another through
    ``UpdatedQueryPts`` -- always run on one rank, in order.  No data-path collective; results are collected with
    ``all_gather_tables`` ((M,4) rows [x, y, image id, keypoint index])."""
    idx = list(range(n_tracks))
    if seed is not None:
        import random
        random.Random(seed).shuffle(idx)
    return idx[rank::world_size]


def exhaustive_pairs(n_images: int) -> List[Tuple[int, int]]:
    """All i<j pairs in the order of src/construct_pairs/pairs_exhaustive.py:5-11."""
    return [(i, j) for i in range(n_images) for j in range(i + 1, n_images)]


def all_gather_tables(tables: List[torch.Tensor], group=None) -> List[torch.Tensor]:
    """Gather every rank's list of [M_k, W] float32 tables; returns the concatenated list in rank
    order on every rank.  Three small-to-one-large collectives regardless of the number of tables:
    (0) all-gather of (table count, width), (1) all-gather of the row counts (padded to the max table count),
    (2) all-gather of one flat payload per rank padded to the largest payload."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return list(tables)
    ws = dist.get_world_size(group)
    dev = tables[0].device if tables else torch.device(
        "cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    width = tables[0].shape[1] if tables else 0
    meta = torch.tensor([len(tables), width], dtype=torch.int64, device=dev)
    metas = [torch.zeros_like(meta) for _ in range(ws)]
    dist.all_gather(metas, meta, group=group)
    max_tables = max(int(m[0]) for m in metas)
    width = max(int(m[1]) for m in metas)
    rows = torch.zeros(max(max_tables, 1), dtype=torch.int64, device=dev)
    if tables:
        rows[:len(tables)] = torch.tensor([t.shape[0] for t in tables], dtype=torch.int64, device=dev)
    all_rows = [torch.zeros_like(rows) for _ in range(ws)]
    dist.all_gather(all_rows, rows, group=group)
    totals = [int(r.sum()) for r in all_rows]
    max_total = max(max(totals), 1)
    payload = torch.zeros((max_total, max(width, 1)), dtype=torch.float32, device=dev)
    if tables and totals[dist.get_rank(group)] > 0:
        payload[:totals[dist.get_rank(group)]] = torch.cat([t.to(torch.float32) for t in tables], 0)
    gathered = [torch.zeros_like(payload) for _ in range(ws)]
    dist.all_gather(gathered, payload, group=group)
    out = []
    for r in range(ws):
        off = 0
        for k in range(int(metas[r][0])):
            n = int(all_rows[r][k])
            out.append(gathered[r][off:off + n, :width])
            off += n
    return out
