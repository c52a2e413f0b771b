"""Multi-process path on CPU (gloo, world_size 2 and 8): static sharding + the collectives of the match tables."""
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


def test_tiled_pair_shards_cover_and_balance():
    """shard_pairs_tiled: a partition of the pair list, balanced, and a rank touches ~ n/sqrt(world) images."""
    pairs = ddist.exhaustive_pairs(300)
    for ws in (1, 2, 3, 4, 8):
        sh = ddist.shard_pairs_tiled(pairs, 300, ws)
        assert len(sh) == ws and sorted(k for s in sh for k in s) == list(range(len(pairs)))
        assert all(s == sorted(s) for s in sh)
        assert max(len(s) for s in sh) <= 1.02 * (len(pairs) / ws) + 1
    imgs8 = [len({x for k in s for x in pairs[k]}) for s in ddist.shard_pairs_tiled(pairs, 300, 8)]
    assert max(imgs8) <= 150                                        # contiguous shards: 300, 263, ... images
    sparse = [(0, 5), (5, 0), (2, 3), (7, 1)]                       # any list, either order, also fewer pairs than ranks
    sh = ddist.shard_pairs_tiled(sparse, 8, 8)
    assert sorted(k for s in sh for k in s) == [0, 1, 2, 3]
    assert ddist.shard_pairs_tiled([], 4, 2) == [[], []]


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
    # gather-to-root: only the root receives (an exact-size buffer); int32 tables travel bit for bit
    for root in (0, 1):
        got = ddist.collect_tables(all_tables[lo:hi], root=root)
        ok = ok and ((got is None) if rank != root else
                     (len(got) == 10 and all(torch.equal(a, b) for a, b in zip(got, all_tables))))
    ints = [torch.arange(12, dtype=torch.int32).view(3, 4) * (rank + 1) - 7, torch.zeros((0, 4), dtype=torch.int32)]
    gi = ddist.collect_tables(ints, root=None, dtype=torch.int32)
    ok = ok and len(gi) == 4 and gi[0].dtype == torch.int32 and torch.equal(gi[2], torch.arange(12, dtype=torch.int32).view(3, 4) * 2 - 7)
    none = ddist.collect_tables([], root=0)
    ok = ok and (none == [] if rank == 0 else none is None)
    # packed form: one rows tensor + the counts, no list of views
    rows, cnt = ddist.collect_tables(all_tables[lo:hi], root=None, packed=True)
    ok = ok and torch.equal(rows, torch.cat(all_tables)) and cnt.tolist() == [t.shape[0] for t in all_tables]
    # an argument error on ONE rank (float64 table / mixed widths) raises on EVERY rank, before the payload collective
    for bad in ([torch.ones((2, 5), dtype=torch.float64)], [torch.ones(2, 5), torch.ones(2, 4)]):
        try:
            ddist.collect_tables(bad if rank == 1 else all_tables[lo:hi], root=None)
            ok = False
        except TypeError as e:
            ok = ok and "rank 1" in str(e)
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


def _scene_worker(rank, world, port, q):
    """plugin.match_scene_sharded on 2 gloo ranks (CPU stand-ins behind ``ops``) == the single-process scene."""
    import numpy as np
    from cpu_standins import cpu_ops
    from detectorfreesfm_amd import HipLoFTR, plugin, synth
    from detectorfreesfm_amd.config import loftr_coarse_only_config
    from detectorfreesfm_amd.params import loftr_param_spec, planted_loftr_state_dict
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = loftr_coarse_only_config(0.2)
    m = HipLoFTR(cfg).eval()
    m.load_state_dict(planted_loftr_state_dict(loftr_param_spec(cfg), 0), strict=True)
    base = synth.coarse_pair_batch(2, 64, 96, seed=1000)
    images = torch.cat([base["image0"], base["image1"]], 0)            # 4 images -> 6 exhaustive pairs, 3 per rank
    names = [f"scene/img{k}.jpg" for k in range(4)]
    with cpu_ops(), torch.no_grad():
        on_root = plugin.match_scene_sharded(m, images, names, " ", batch=2, root=0)   # opt-in: gather-to-root 0
        ok = (on_root is None) == (rank != 0)
        matches, kp, sc, upd = plugin.match_scene_sharded(m, images, names, " ", batch=2)   # default: on EVERY rank
        contig = plugin.match_scene_sharded(m, images, names, " ", batch=2, shard="contiguous")
        # the sharding changes which pairs share a transformer batch (summation order), never the table layout
        ok = ok and list(contig[0]) == list(matches) and all(
            np.array_equal(contig[0][k][:, :4], matches[k][:, :4]) and np.allclose(contig[0][k][:, 4], matches[k][:, 4], atol=1e-4)
            for k in matches)
        ok = ok and len(matches) == 6 and list(matches) == [f"{names[i]} {names[j]}" for i, j in ddist.exhaustive_pairs(4)]
        if rank == 0:          # the same scene in one process
            ok = ok and all(np.array_equal(on_root[0][k], matches[k]) for k in matches)
            full = plugin.match_scene_cached(m, images, ddist.exhaustive_pairs(4), batch=2)
            ref = {f"{names[i]} {names[j]}": t for (i, j), t in full.items()}
            kp1, sc1, upd1 = plugin.merge_match_tables(ref, names, " ", device="cpu")
            ok = ok and all(np.array_equal(matches[k][:, :4], ref[k][:, :4]) and
                            np.allclose(matches[k][:, 4], ref[k][:, 4], atol=1e-4) for k in ref)
            ok = ok and all(np.array_equal(kp[n], kp1[n]) for n in names) and all(np.array_equal(upd[k], upd1[k]) for k in ref)
            ok = ok and sum(len(t) for t in ref.values()) > 20
    q.put((rank, bool(ok), sum(len(t) for t in matches.values())))
    dist.barrier()
    dist.destroy_process_group()


def test_scene_sharded_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_scene_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok, _ in res), res
    assert res[0][2] == res[1][2] > 20          # every rank holds the full set of tables


def _refine_worker(rank, world, port, q):
    """plugin.refine_scene_sharded on 2 gloo ranks (CPU stand-ins behind ``ops``): every rank ends with all rows, and
    the rows equal the single-process scene track by track."""
    import numpy as np
    from cpu_standins import cpu_ops
    from detectorfreesfm_amd import HipMultiviewMatcher, plugin
    from detectorfreesfm_amd.config import multiview_refinement_config
    from detectorfreesfm_amd.params import multiview_param_spec, random_state_dict
    from detectorfreesfm_amd.synth import SyntheticSfMScene
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = multiview_refinement_config()
    m = HipMultiviewMatcher(cfg, test=True).eval()
    m.load_state_dict(random_state_dict(multiview_param_spec(cfg), 1), strict=True)
    scene = SyntheticSfMScene(n_images=5, n_points=24, seed=3, hw=(64, 96), max_views=4)
    dcfg = {"max_track_length": 16, "chunk": 6000}

    def table(results):
        rows = np.concatenate(results, 0)
        return {(int(r[2]), int(r[3])): r[:2] for r in rows}, rows.shape[0]
    with cpu_ops(), torch.no_grad():
        res_all = plugin.refine_scene_sharded(m, scene, dcfg, seed=2, device="cpu")               # default: on EVERY rank
        got, n_got = table(res_all)
        res_root = plugin.refine_scene_sharded(m, scene, dcfg, seed=2, device="cpu", root=0)      # opt-in: gather-to-root 0
        ok = (res_root is None) == (rank != 0)
        if rank == 0:
            ok = ok and len(res_root) == len(res_all) and all(a.dtype == np.float64 and np.array_equal(a, b)
                                                              for a, b in zip(res_root, res_all))
        if rank == 0:
            ref, n_ref = table(plugin.match_tracks_worker(scene, m, None, dcfg, device="cpu"))
            ok = ok and n_got == n_ref and set(got) == set(ref) and all(np.abs(got[k] - ref[k]).max() < 1e-3 for k in ref)
    q.put((rank, bool(ok), n_got))
    dist.barrier()
    dist.destroy_process_group()


def test_refine_scene_sharded_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + os.getpid() % 2000
    procs = [ctx.Process(target=_refine_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok, _ in res), res
    assert res[0][2] == res[1][2] > 40


def _worker8(rank, world, port, q):
    """collect_tables at the target world size (8 ranks, gloo): every form, ranks WITHOUT tables, both error paths."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(200)
    per_rank = [0 if r in (2, 6) else 1 + (r % 3) for r in range(world)]              # ranks 2 and 6 hold no table at all
    all_tables = [[torch.rand((int(torch.randint(0, 7, (1,), generator=g)), 5), generator=g) for _ in range(n)] for n in per_rank]
    flat = [t for ts in all_tables for t in ts]
    mine = all_tables[rank]
    ok = True
    got = ddist.collect_tables(mine, root=None)                                       # list form on every rank
    ok = ok and len(got) == len(flat) and all(torch.equal(a, b) for a, b in zip(got, flat))
    rows, cnt = ddist.collect_tables(mine, root=None, packed=True)                    # packed form on every rank
    ok = ok and torch.equal(rows, torch.cat(flat)) and cnt.tolist() == [t.shape[0] for t in flat]
    for root in (0, 5, 6):                                                            # gather-to-root, incl. a root without tables
        got = ddist.collect_tables(mine, root=root)
        ok = ok and ((got is None) if rank != root else (len(got) == len(flat) and all(torch.equal(a, b) for a, b in zip(got, flat))))
        pk = ddist.collect_tables(mine, root=root, packed=True)
        ok = ok and ((pk is None) if rank != root else (torch.equal(pk[0], torch.cat(flat)) and pk[1].tolist() == [t.shape[0] for t in flat]))
    ints = [torch.full((rank % 4, 4), rank, dtype=torch.int32)] if rank % 2 else []   # int32 rows, half the ranks empty
    gi = ddist.collect_tables(ints, root=None, dtype=torch.int32)
    ok = ok and len(gi) == 4 and all(t.dtype == torch.int32 for t in gi) and [int(t.shape[0]) for t in gi] == [1, 3, 1, 3] \
        and [int(t[0, 0]) for t in gi] == [1, 3, 5, 7]
    none = ddist.collect_tables([], root=None)                                        # nobody has anything
    ok = ok and none == []
    # error on ONE rank -> the same TypeError on ALL ranks, nobody left inside the payload collective
    for bad_rank, bad in ((3, [torch.ones((2, 5), dtype=torch.float64)]), (7, [torch.ones(2, 5), torch.ones(2, 4)])):
        for root in (None, 0):
            try:
                ddist.collect_tables(bad if rank == bad_rank else mine, root=root)
                ok = False
            except TypeError as e:
                ok = ok and f"rank {bad_rank}" in str(e)
    # widths that differ BETWEEN ranks (each rank consistent in itself): raised everywhere as well, in every form
    odd = [torch.ones(3, 4)] if rank == 4 else mine
    for kw in ({"root": None}, {"root": None, "packed": True}, {"root": 0}):
        try:
            ddist.collect_tables(odd, **kw)
            ok = False
        except TypeError as e:
            ok = ok and "rank 4: 4" in str(e)
    after = ddist.collect_tables(mine, root=None)                                     # the group is still usable afterwards
    ok = ok and len(after) == len(flat)
    q.put((rank, bool(ok), len(got) if got is not None else -1))
    dist.barrier()
    dist.destroy_process_group()


def _spawn(target, world, port, timeout):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=target, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=timeout) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    return res


def test_collect_tables_world8():
    res = _spawn(_worker8, 8, 35500 + os.getpid() % 2000, 300)
    assert len(res) == 8 and all(ok for _, ok, _ in res), res


def _scene_worker8(rank, world, port, q):
    """plugin.match_scene_sharded on 8 gloo ranks, 24 images / 276 exhaustive pairs, tiled shards (CPU stand-ins behind
    ``ops``): every rank ends with all 276 tables in pair-list order; the tiled and the contiguous sharding give the same
    rows; a sample of pairs equals the single-process path; a rank touches ~ half the images."""
    import numpy as np
    from cpu_standins import cpu_ops
    from detectorfreesfm_amd import HipLoFTR, plugin, synth
    from detectorfreesfm_amd.config import loftr_coarse_only_config
    from detectorfreesfm_amd.params import loftr_param_spec, planted_loftr_state_dict
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = loftr_coarse_only_config(0.2)
    m = HipLoFTR(cfg).eval()
    m.load_state_dict(planted_loftr_state_dict(loftr_param_spec(cfg), 0), strict=True)
    base = synth.coarse_pair_batch(12, 48, 64, seed=1000)
    images = torch.cat([base["image0"], base["image1"]], 0)            # 24 images -> 276 exhaustive pairs
    names = [f"scene/img{k:02d}.jpg" for k in range(24)]
    pairs = ddist.exhaustive_pairs(24)
    shards = ddist.shard_pairs_tiled(pairs, 24, world)
    ok = sorted(k for s in shards for k in s) == list(range(276)) and max(len(s) for s in shards) <= 36
    ok = ok and max(len({x for k in s for x in pairs[k]}) for s in shards) <= 16   # contiguous shards of this list: up to 24
    with cpu_ops(), torch.no_grad():
        matches, kp, sc, upd = plugin.match_scene_sharded(m, images, names, " ", batch=4)            # tiled, on EVERY rank
        on_root = plugin.match_scene_sharded(m, images, names, " ", batch=4, root=0, shard="contiguous")
        ok = ok and (on_root is None) == (rank != 0)
        ok = ok and list(matches) == [f"{names[i]} {names[j]}" for i, j in pairs]
        if rank == 0:
            c = on_root[0]
            ok = ok and list(c) == list(matches) and all(
                np.array_equal(c[k][:, :4], matches[k][:, :4]) and np.allclose(c[k][:, 4], matches[k][:, 4], atol=1e-4) for k in matches)
            sample = [pairs[k] for k in range(0, 276, 23)]
            ref = plugin.match_scene_cached(m, images, sample, batch=4)
            for (i, j), t in ref.items():
                g = matches[f"{names[i]} {names[j]}"]
                ok = ok and np.array_equal(g[:, :4], t[:, :4]) and np.allclose(g[:, 4], t[:, 4], atol=1e-4)
            ok = ok and all(len(kp[n]) > 0 for n in names)
    q.put((rank, bool(ok), sum(len(t) for t in matches.values())))
    dist.barrier()
    dist.destroy_process_group()


def test_scene_sharded_tiled_world8():
    res = _spawn(_scene_worker8, 8, 37500 + os.getpid() % 2000, 900)
    assert len(res) == 8 and all(ok for _, ok, _ in res), res
    assert len({n for _, _, n in res}) == 1 and res[0][2] > 3000       # every rank holds the same full set of tables


def test_tiled_pair_shards_level_sparse_lists():
    """A star-shaped covisibility list (one hub image paired with everything + a thin band) defeats every block grid; the
    pair-level levelling keeps the shards within one pair of the ideal."""
    n = 200
    pairs = [(0, j) for j in range(1, n)] + [(i, i + 1) for i in range(1, n - 1)] + [(i, i + 2) for i in range(1, n - 2)]
    for ws in (3, 8):
        sh = ddist.shard_pairs_tiled(pairs, n, ws)
        assert sorted(k for s in sh for k in s) == list(range(len(pairs)))
        ideal = -(-len(pairs) // ws)
        assert max(len(s) for s in sh) <= max(ideal, int(1.02 * ideal)), [len(s) for s in sh]


def test_single_process_passthrough():
    t = [torch.ones(3, 5), torch.zeros(0, 5)]
    assert ddist.all_gather_tables(t)[0] is t[0]


def test_bench_step_loop_dry_run_world8():
    """no 8-GPU node has run ``bench.py`` yet, so the exact step loops of ``bench.py --gpus 8`` (rank / world from the
    launcher's environment, shards, barriers, the per-step gather of match tables to rank 0, the max over ranks, ONE JSON line from rank
    0) are rehearsed here as 8 processes over gloo on the CPU stand-ins (DFSFM_BENCH_DRYRUN=cpu, toy frames), launched the way the
    driver launches it.  And ``--gpus N`` with another number of ranks refuses to report."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DFSFM_BENCH_DRYRUN="cpu", OMP_NUM_THREADS="1", MKL_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", "29547", os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--batch", "1",
           "--tracks", "6"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]                       # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["config"]["rccl_ranks"] == 8 and d["steps"] == 2 and d["scaling"] == "weak"
    assert d["data"].startswith("dry-run") and d["value"] > 0 and d["refinement_tracks_per_sec"] > 0
    assert abs(d["value"] - 8 * 1 * 1000.0 / d["ms_per_step"]) < 1e-6 * d["value"]          # whole-job rate: 8 ranks x batch 1 per step
    assert d["matches_last_step"] > 0                                # rank 0 received the gathered tables of all ranks
    # a line labelled N that N ranks did not produce must not exist
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=300, env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"), cwd=root)
    assert bad.returncode != 0 and "refusing to report" in (bad.stderr + bad.stdout)
    assert not any(l.strip().startswith("{") for l in bad.stdout.splitlines())
