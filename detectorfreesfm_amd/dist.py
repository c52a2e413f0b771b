"""One-process-per-GPU sharding of the two embarrassingly parallel work lists and the single
collective of the path: an all-gather of variable-length match tables.

The reference fans pairs / tracks out over Ray tasks and merges pickled numpy results through the
object store (src/coarse_match/coarse_match.py:127-140, src/post_optimization/matcher_model/
multiview_match.py:39-62).  Here every rank takes a static contiguous shard (no data-path
collective), and the tables are collected with ``collect_tables``: a four-integer metadata all-gather plus ONE payload
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


def shard_pairs_tiled(pairs: Sequence[Tuple[int, int]], n_images: int, world_size: int) -> List[List[int]]:
    """Indices into ``pairs`` for every rank such that a rank's pairs touch few images (the backbone runs once per image a
    rank sees, ``plugin.match_scene_cached``): the (i, j) plane is cut into g x g blocks of ceil(n/g) images, g = the
    smallest integer with g(g+1)/2 >= world_size (g ~ sqrt(2 world)) for which the deal below balances to 2 %; the non-empty
    upper-triangular blocks are dealt to the
    ranks heaviest first onto the least-loaded rank (ties: the rank that already owns a block in the same block row /
    column).  A rank then sees ~ 2n/g images (150 of 300 at world 8) instead of nearly all of them with a contiguous
    shard of the i<j list (src/construct_pairs/pairs_exhaustive.py order), and the pair counts stay balanced (300 images,
    world 8: 5550 ... 5625 pairs per rank).  Deterministic and identical on every rank; inside a rank the pairs keep their
    list order."""
    if world_size <= 1:
        return [list(range(len(pairs)))]

    def deal(g):
        B = max((n_images + g - 1) // g, 1)
        tiles = {}
        for k, (i, j) in enumerate(pairs):
            a, b = int(i) // B, int(j) // B
            tiles.setdefault((min(a, b), max(a, b)), []).append(k)
        load = [0] * world_size
        own = [set() for _ in range(world_size)]
        out = [[] for _ in range(world_size)]
        for key in sorted(tiles, key=lambda t: (-len(tiles[t]), t)):
            best = min(range(world_size), key=lambda r: (load[r], -len(own[r] & set(key)), r))
            load[best] += len(tiles[key])
            own[best] |= set(key)
            out[best].extend(tiles[key])
        return max(load), [sorted(o) for o in out]

    g0 = 1
    while g0 * (g0 + 1) // 2 < world_size:
        g0 += 1
    ideal = (len(pairs) + world_size - 1) // world_size
    best = None
    for g in range(g0, g0 + 4):                      # the coarsest block grid whose heaviest rank is within 2 % of the mean
        worst, out = deal(g)
        if best is None or worst < best[0]:
            best = (worst, out)
        if worst <= 1.02 * ideal:
            break
    worst, out = best
    if worst > 1.02 * ideal:
        # Sparse / star-shaped pair lists (covisibility graphs) can leave every block grid off balance: level the ranks by moving
        # single pairs from the heaviest ranks to the lightest ones.  A moved pair may bring one or two more images to its new rank;
        # the blocks themselves stay where they were dealt, so the image locality of the deal is kept up to those pairs.
        out = [list(o) for o in out]
        load = [len(o) for o in out]
        order = sorted(range(world_size), key=lambda r: -load[r])
        for r in order:
            while load[r] > ideal:
                to = min(range(world_size), key=lambda q: (load[q], q))
                if load[to] + 1 > ideal or to == r:
                    break
                n = min(load[r] - ideal, ideal - load[to])
                out[to].extend(out[r][-n:])
                del out[r][-n:]
                load[r] -= n
                load[to] += n
        out = [sorted(o) for o in out]
    return out


_WORD_DTYPES = (torch.float32, torch.int32)


def collect_tables(tables: List[torch.Tensor], group=None, root=None, dtype=torch.float32, packed=False):
    """Collect every rank's list of [M_k, W] tables (float32 or int32; all of one width) in rank order.

    root=None: every rank receives the full list (ONE metadata all-gather of four integers per rank, then ONE
    ``all_gather_into_tensor`` of a flat word buffer that carries the row counts in-band; a single [world, len] receive
    buffer, no per-rank Python lists).
    root=r (a rank OF ``group``, i.e. group-local): gather-to-root -- the merge of a scene runs on one rank ("a
    gather-to-root suffices"), so only rank r allocates a receive buffer, of exactly the summed size (``all_to_all_single``
    with uneven splits: nothing is padded to the largest rank and the other ranks receive nothing); they get ``None``.
    packed=True: returns ``(rows [sum M, W], counts int32 [n_tables])`` device tensors instead of a list of views -- no
    second host read and no per-table Python object (what a per-step caller such as bench.py wants; the list form splits
    ``rows`` by ``counts`` with one ``torch.split`` per rank).
    Every argument error (a dtype that is not a 4-byte word, tables of mixed width on one rank or of different widths on
    different ranks) is detected BEFORE the payload collective from the metadata every rank holds, so all ranks raise together
    instead of one rank raising while the others wait inside the collective.
    The reference moves the same tables as pickled numpy arrays through Ray's object store
    (src/coarse_match/coarse_match.py:127-140; multiview_match.py:39-62)."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        if packed:
            dev = tables[0].device if tables else torch.device("cpu")
            rows = torch.cat(list(tables)) if tables else torch.empty((0, 0), dtype=dtype, device=dev)
            return rows, torch.tensor([t.shape[0] for t in tables], dtype=torch.int32, device=dev)
        return list(tables)
    ws, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = _group_device(tables, group)
    err = 0
    widths = {int(t.shape[1]) for t in tables if t.dim() == 2}
    if any(t.dim() != 2 for t in tables) or len(widths) > 1:
        err = 2
    elif any(t.dtype not in _WORD_DTYPES for t in tables):
        err = 1
    width = widths.pop() if len(widths) == 1 else 0
    total = 0 if err else sum(int(t.shape[0]) for t in tables)
    meta = torch.tensor([0 if err else len(tables), width, total, err], dtype=torch.int64, device=dev)
    metas = torch.empty((ws * 4,), dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(metas, meta, group=group)
    metas = metas.view(ws, 4).tolist()                                            # host read 1: four integers per rank
    bad = [(r, m[3]) for r, m in enumerate(metas) if m[3]]
    if bad:                                                                       # every rank raises, none is left waiting
        what = {1: "tables travel as 4-byte words (float32 / int32)", 2: "tables must be [M, W] with one width per rank"}
        raise TypeError("collect_tables: " + "; ".join(f"rank {r}: {what[e]}" for r, e in bad))
    wmax = max(m[1] for m in metas)
    odd = [(r, m[1]) for r, m in enumerate(metas) if m[1] and m[1] != wmax]
    if odd:                                                                       # widths are per call, not per rank
        raise TypeError("collect_tables: table widths differ between ranks: " +
                        ", ".join(f"rank {r}: {w}" for r, w in odd) + f" vs {wmax}")
    # this rank's words: [row counts (n_tables) | rows (total x width)]
    counts = torch.tensor([t.shape[0] for t in tables], dtype=torch.int32, device=dev)
    words = torch.cat([counts] + [t.contiguous().view(torch.int32).reshape(-1) for t in tables if t.shape[0]]) if tables \
        else torch.empty((0,), dtype=torch.int32, device=dev)
    lens = [m[0] + m[2] * m[1] for m in metas]

    def unpack(buf, offs):
        cnts = [buf[offs[r]:offs[r] + metas[r][0]] for r in range(ws)]
        rows = [buf[offs[r] + metas[r][0]:offs[r] + lens[r]].view(dtype).view(metas[r][2], metas[r][1])
                for r in range(ws)]
        if packed:
            keep = [x for x, m in zip(rows, metas) if m[2]]                       # widths were checked equal above
            return (torch.cat(keep) if keep else torch.empty((0, wmax), dtype=dtype, device=dev)), torch.cat(cnts)
        host = torch.cat(cnts).tolist() if sum(m[0] for m in metas) else []      # host read 2: the row counts
        out, k = [], 0
        for r in range(ws):
            c = host[k:k + metas[r][0]]
            k += metas[r][0]
            if not c:
                continue
            if metas[r][1]:
                out.extend(rows[r].split(c))                                      # one call per rank, not one per table
            else:
                out.extend(torch.empty((n, wmax), dtype=dtype, device=dev) for n in c)
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
