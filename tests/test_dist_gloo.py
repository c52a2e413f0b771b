"""Multi-process path on CPU (gloo, world_size 2): static sharding + all-gather of match tables."""
import os

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from detectorfreesfm_amd import dist as ddist


def test_shard_ranges_partition_the_list():
    for n in (0, 1, 7, 8, 44850):
        for ws in (1, 2, 3, 8):
            spans = [ddist.shard_range(n, r, ws) for r in range(ws)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(ws - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    assert len(ddist.exhaustive_pairs(300)) == 44850
    assert ddist.exhaustive_pairs(3) == [(0, 1), (0, 2), (1, 2)]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pairs = ddist.shard_list(ddist.exhaustive_pairs(5), rank, world)
    g = torch.Generator().manual_seed(100)
    all_tables = [torch.rand((int(torch.randint(0, 6, (1,), generator=g)), 5), generator=g) for _ in range(10)]
    lo, hi = ddist.shard_range(10, rank, world)
    got = ddist.all_gather_tables(all_tables[lo:hi])
    ok = len(got) == 10 and all(torch.equal(a, b) for a, b in zip(got, all_tables))
    empty = ddist.all_gather_tables([] if rank == 0 else [torch.ones(2, 4)])
    ok = ok and len(empty) == 1 and torch.equal(empty[0], torch.ones(2, 4))
    q.put((rank, ok, len(pairs)))
    dist.barrier()
    dist.destroy_process_group()


def test_all_gather_tables_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok, _ in res)
    assert sum(n for _, _, n in res) == 10


def test_single_process_passthrough():
    t = [torch.ones(3, 5), torch.zeros(0, 5)]
    assert ddist.all_gather_tables(t)[0] is t[0]
