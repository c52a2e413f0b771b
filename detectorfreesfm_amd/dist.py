"""One-process-per-GPU sharding of the two embarrassingly parallel work lists and the single
collective of the path: an all-gather of variable-length match tables.

The reference fans pairs / tracks out over Ray tasks and merges pickled numpy results through the
object store (src/coarse_match/coarse_match.py:127-140, src/post_optimization/matcher_model/
multiview_match.py:39-62).  Here every rank takes a static contiguous shard (no data-path
collective), and the tables are collected with ``collect_tables``: a three-integer metadata all-gather plus ONE payload
collective -- a flat ``all_gather_into_tensor`` when every rank needs the scene's tables, an exact-size gather-to-root
(``all_to_all_single`` with uneven splits) when only the merging rank does -- RCCL over xGMI with backend "nccl" on MI355X,
gloo on CPU for the tests.
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


def _group_device(tables, group):
    if tables:
        return tables[0].device
    if dist.get_backend(group) == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def _as_words(t: torch.Tensor) -> torch.Tensor:
    """A [M, W] table of a 4-byte dtype (float32 / int32) as int32 words, bit for bit."""
    if t.dtype not in (torch.float32, torch.int32):
        raise TypeError(f"collect_tables: tables travel as 4-byte words (float32 / int32), got {t.dtype}")
    return t.contiguous().view(torch.int32)


def collect_tables(tables: List[torch.Tensor], group=None, root=None, dtype=torch.float32):
    """Collect every rank's list of [M_k, W] tables (float32 or int32; all of one width) in rank order.

    root=None: every rank receives the full list (ONE metadata all-gather of three integers per rank, then ONE
    ``all_gather_into_tensor`` of a flat word buffer that carries the row counts in-band; a single [world, len] receive
    buffer, no per-rank Python lists, two host reads in total).
    root=r: gather-to-root -- the merge of a scene runs on one rank ("a gather-to-root suffices"), so only rank r
    allocates a receive buffer, of exactly the summed size (``all_to_all_single`` with uneven splits: nothing is padded to the
    largest rank and the other ranks receive nothing); they get ``None``.
    The reference moves the same tables as pickled numpy arrays through Ray's object store
    (src/coarse_match/coarse_match.py:127-140; multiview_match.py:39-62)."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return list(tables)
    ws, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = _group_device(tables, group)
    width = tables[0].shape[1] if tables else 0
    total = sum(t.shape[0] for t in tables)
    meta = torch.tensor([len(tables), width, total], dtype=torch.int64, device=dev)
    metas = torch.empty((ws * 3,), dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(metas, meta, group=group)
    metas = metas.view(ws, 3).tolist()                                            # host read 1: three integers per rank
    width = max(m[1] for m in metas)
    # this rank's words: [row counts (n_tables) | rows (total x width)]
    counts = torch.tensor([t.shape[0] for t in tables], dtype=torch.int32, device=dev)
    words = torch.cat([counts] + [_as_words(t).reshape(-1) for t in tables if t.shape[0]]) if tables else \
        torch.empty((0,), dtype=torch.int32, device=dev)
    lens = [m[0] + m[2] * m[1] for m in metas]
    assert words.numel() == lens[rank]

    def unpack(buf, offs):
        cnt = torch.cat([buf[offs[r]:offs[r] + metas[r][0]] for r in range(ws)]).tolist() if sum(m[0] for m in metas) else []
        out, k = [], 0                                                # host read 2: the row counts
        for r in range(ws):
            o = offs[r] + metas[r][0]
            for _ in range(metas[r][0]):
                n = cnt[k]
                k += 1
                out.append(buf[o:o + n * metas[r][1]].view(dtype).view(n, metas[r][1]) if metas[r][1] else
                           torch.empty((n, width), dtype=dtype, device=dev))
                o += n * metas[r][1]
        return out

    if root is None:
        pitch = max(max(lens), 1)
        send = torch.zeros((pitch,), dtype=torch.int32, device=dev)
        send[:words.numel()] = words
        recv = torch.empty((ws * pitch,), dtype=torch.int32, device=dev)
        dist.all_gather_into_tensor(recv, send, group=group)
        return unpack(recv, [r * pitch for r in range(ws)])
    in_split = [words.numel() if r == root else 0 for r in range(ws)]
    out_split = lens if rank == root else [0] * ws
    recv = torch.empty((sum(out_split),), dtype=torch.int32, device=dev)
    send = words if words.numel() else torch.empty((0,), dtype=torch.int32, device=dev)
    dist.all_to_all_single(recv, send, out_split, in_split, group=group)
    if rank != root:
        return None
    offs, o = [], 0
    for n in lens:
        offs.append(o)
        o += n
    return unpack(recv, offs)


def all_gather_tables(tables: List[torch.Tensor], group=None) -> List[torch.Tensor]:
    """``collect_tables(..., root=None)`` for float32 tables (the name the round-1/2 callers and tests use)."""
    return collect_tables(tables, group=group, root=None, dtype=torch.float32)
