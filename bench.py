#!/usr/bin/env python
"""Hot-path benchmark (driver contract: prints ONE JSON line on rank 0).

A *step* = one pass of the coarse matcher over one batch of 8 synthetic 640x480 pairs that are
already resident in HBM (BASELINE.json configs[1]); ``value`` = image pairs per second over all
ranks.  The same run also times the refinement head (configs[2]: 2000 tracks x 5 views) and
reports it under ``secondary``.  Pairs / track bags shard across ranks with no data-path
collective (weak scaling: every rank runs its own batch); with N>1 every step ends with the
path's one real exchange, the all-gather of the match tables (RCCL over xGMI).

Extra objects: ``roofline`` (dominant hand-written kernel, measured live with events on the
launch stream), ``rooflines`` (all hand-written kernels), ``breakdown`` (stage times), and on
rank 0 at N=1 ``cpu_baseline`` (the oracle = CPU port of the reference path, host cores).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from detectorfreesfm_amd import HipLoFTR, HipMultiviewMatcher, ops, synth  # noqa: E402
from detectorfreesfm_amd import dist as ddist  # noqa: E402
from detectorfreesfm_amd.config import loftr_coarse_only_config, multiview_refinement_config  # noqa: E402
from detectorfreesfm_amd.params import loftr_param_spec, multiview_param_spec, random_state_dict  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
MFMA_F32_PEAK_TF = 157.3     # fp32-input MFMA dense peak
MFMA_F16_PEAK_TF = 2500.0    # fp16/bf16 dense MFMA peak (spec; 2178-2382 TF measured micro-benchmarks)


def event_time_ms(fn, iters=10, warmup=2):
    """Average duration of fn() with events on the current (= launch) stream."""
    for _ in range(warmup):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) / iters


def timed_steps(step, steps, warmup, distributed):
    for _ in range(warmup):
        step()
    if distributed:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    if distributed:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def kernel_rooflines(dev, batch):
    """Hand-written kernels at the bench shapes, inputs resident, events on the launch stream."""
    out = []
    g = torch.Generator().manual_seed(0)
    # K6/K9/K2 conv_gemm_sf (fp16x2-split implicit GEMM on the fp16 matrix cores): the layer that dominates the
    # coarse step -- BasicBlock 3x3, 128->128 channels at 240x320, 2*batch images.  Algorithmic flops =
    # 2*M*Cout*kh*kw*Cin (counted once; the kernel issues 3 fp16 MFMAs per product for fp32-class accuracy).
    nimg = 2 * batch
    x = torch.randn((nimg, 240, 320, 128), generator=g).to(dev)
    xs = ops.SplitAct.empty(nimg, 240, 320, 128, dev)
    ops.split_rows(x, None, out_split=xs)
    pw = ops.PackedDense(torch.randn((128, 128, 3, 3), generator=g).to(dev) * 0.03, torch.zeros(128, device=dev), cin_pad=128,
                         tap_padded=True)
    ms = event_time_ms(lambda: ops.conv2d_nhwc(xs, pw, 1, 1, relu=True, out_split=True))
    flops = 2.0 * nimg * 240 * 320 * 128 * 9 * 128
    out.append({"kernel": "conv_gemm_sf_same_kernel<128,3> (3x3, 128->128 @240x320)", "bound": "mfma",
                "achieved": flops / ms / 1e9, "peak": MFMA_F16_PEAK_TF, "unit": "TFLOP/s",
                "frac": flops / ms / 1e9 / MFMA_F16_PEAK_TF, "traffic": None, "ms": ms, "units": f"{nimg} images",
                "mfma_flops_executed_frac": 3.0 * flops / ms / 1e9 / MFMA_F16_PEAK_TF,
                "step_share": "the conv_gemm_sf* kernels are ~70% of the coarse step and ~65% of the refinement step "
                              "(profiles/r01_*_step_kernel_stats.csv)"})
    del x, xs, pw
    # K3+K4+K5 at batch x (4800 x 4800 x 256): algorithmic flops 2*L*S*C per pair
    L = S = 4800
    f0, f1 = synth.correlated_features(batch, L, S, 256, 7, 0.1)
    f0, f1 = f0.to(dev), f1.to(dev)
    s0, s1 = ops.SplitAct.empty_rows((batch, L), 256, dev), ops.SplitAct.empty_rows((batch, S), 256, dev)
    ops.split_rows(f0, None, out_split=s0)
    ops.split_rows(f1, None, out_split=s1)
    ms = event_time_ms(lambda: ops.coarse_match(s0, s1, (60, 80), (60, 80), 0.2, 2, 0.1))
    flops = 2.0 * L * S * 256 * batch
    out.append({"kernel": "coarse_match_split (cm_gemm_sf x2 + select; the correlation is computed twice, 3 fp16 "
                          "MFMA products each)", "bound": "mfma", "achieved": flops / ms / 1e9,
                "peak": MFMA_F16_PEAK_TF, "unit": "TFLOP/s", "frac": flops / ms / 1e9 / MFMA_F16_PEAK_TF,
                "traffic": None, "ms": ms, "units": f"{batch} pairs",
                "mfma_flops_executed_frac": 6.0 * flops / ms / 1e9 / MFMA_F16_PEAK_TF})
    ms32 = event_time_ms(lambda: ops.coarse_match(f0, f1, (60, 80), (60, 80), 0.2, 2, 0.1))
    out.append({"kernel": "coarse_match_f32 (cm_gemm x2 + select, fp32 matrix path; generic entry point)",
                "bound": "mfma", "achieved": flops / ms32 / 1e9, "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
                "frac": flops / ms32 / 1e9 / MFMA_F32_PEAK_TF, "traffic": None, "ms": ms32, "units": f"{batch} pairs"})
    del s0, s1
    # K1 at the coarse shape, N = 2*batch (both images of every pair in one call): 4*N*L*H*D*4 bytes
    N = 2 * batch
    q, k, v = (torch.randn((N, L, 8, 32), generator=g).to(dev) for _ in range(3))
    ms = event_time_ms(lambda: ops.linear_attention(q, k, v))
    byts = 4.0 * N * L * 256 * 4
    out.append({"kernel": "linear_attention D=32", "bound": "hbm", "achieved": byts / ms / 1e6, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": byts / ms / 1e6 / HBM_PEAK_GBS, "traffic": None, "ms": ms,
                "units": f"{N} x 4800 tokens"})
    del q, k, v, f0, f1
    # K8: 2000 tracks x 5 views of 3x35x35 patches; algorithmic bytes = written + read per patch
    M = 10000
    img = torch.rand((1, 3, 480, 640), generator=g).to(dev)
    pts = (torch.rand((M, 2), generator=g) * torch.tensor([580.0, 420.0]) + 30).to(dev)
    boxes = torch.cat([pts - 17, pts + 17], -1)
    buf = torch.empty((M, 3, 35, 35), device=dev)
    ms = event_time_ms(lambda: ops.roi_align(img, boxes, 35, 35, out=buf))
    del buf
    byts = M * (3 * 35 * 35 * 4 + 3 * 36 * 36 * 4)
    out.append({"kernel": "roi_align 3x35x35", "bound": "hbm", "achieved": byts / ms / 1e6, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": byts / ms / 1e6 / HBM_PEAK_GBS, "traffic": None, "ms": ms,
                "units": f"{M} patches"})
    # K11+K12: 2000 tracks, 4 query views, W=15, C=128: ((V-1)*WW + WW)*C*4 bytes per track
    T, Vq, WW, C = 2000, 4, 225, 128
    ref = torch.randn((T, WW, C), generator=g).to(dev)
    qry = torch.randn((T, Vq, WW, C), generator=g).to(dev)
    mask = torch.ones((T, Vq), dtype=torch.bool, device=dev)
    ms = event_time_ms(lambda: ops.fine_match(ref, qry, mask, None, 15, 7))
    byts = T * (Vq + 1) * WW * C * 4.0
    out.append({"kernel": "fine_match W=15", "bound": "hbm", "achieved": byts / ms / 1e6, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": byts / ms / 1e6 / HBM_PEAK_GBS, "traffic": None, "ms": ms,
                "units": f"{T} tracks"})
    return out


def cpu_baseline(seconds_budget=25.0):
    """The oracle (CPU port of the reference PyTorch path, same weights/inputs) on the host cores:
    a bounded sample -- 640x480 pairs one at a time (the reference runs batch 1) and one bag of
    64 tracks x 5 views."""
    from oracle import restate
    cores = torch.get_num_threads()
    cfg = loftr_coarse_only_config(0.2)
    sd = random_state_dict(loftr_param_spec(cfg), 0)
    data = synth.coarse_pair_batch(1, 480, 640, seed=1000)
    with torch.no_grad():
        restate.loftr_coarse_forward(sd, cfg, data)      # warm-up
        n, t0 = 0, time.perf_counter()
        while n < 5 and (time.perf_counter() - t0) < seconds_budget * 0.6:
            restate.loftr_coarse_forward(sd, cfg, data)
            n += 1
        pairs_per_s = n / (time.perf_counter() - t0)
        rcfg = multiview_refinement_config()
        rsd = random_state_dict(multiview_param_spec(rcfg), 1)
        rdata = synth.refine_bag(T=64, V=5, H=480, W=640, seed=2000)
        t0 = time.perf_counter()
        restate.multiview_matcher_forward(rsd, rcfg, rdata)
        tracks_per_s = 64 / (time.perf_counter() - t0)
    return {"value": pairs_per_s, "unit": "image-pairs/s", "cores": cores, "kind": "port",
            "sample": f"{n} single 640x480 pairs through oracle.restate.loftr_coarse_forward (incl. the FPN "
                      "branch the reference computes and discards)",
            "secondary": {"value": tracks_per_s, "unit": "tracks/s",
                          "sample": "1 bag of 64 tracks x 5 views through oracle.restate.multiview_matcher_forward"}}


def _claim_stdout():
    """Route everything that any library writes to file descriptor 1 (RCCL prints a banner there when a process
    group is created, flushed at exit) to stderr and return a private handle on the real stdout, so that the ONE
    JSON line is the only -- and therefore the last -- thing on it."""
    sys.stdout.flush()
    real = os.dup(1)
    os.dup2(2, 1)
    return real


def _emit(real_stdout_fd, obj):
    os.write(real_stdout_fd, (json.dumps(obj) + "\n").encode())


def main():
    out_fd = _claim_stdout()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=8, help="pairs per step per GPU (BASELINE configs[1]: 8)")
    ap.add_argument("--tracks", type=int, default=2000, help="tracks per refinement bag (configs[2]: 2000)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-rooflines", action="store_true")
    ap.add_argument("--kernels-only", action="store_true",
                    help="only time the hand-written kernels at the bench shapes (compact rocprofv3 target)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # DFSFM_BENCH_FORCE_DIST=1 exercises the RCCL code path (init, barriers, max-reduce, table all-gather) with a
    # single rank; the driver's multi-GPU runs set WORLD_SIZE > 1
    distributed = world > 1 or os.environ.get("DFSFM_BENCH_FORCE_DIST") == "1"
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.distributed.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    if args.kernels_only:
        rl = kernel_rooflines(dev, args.batch)
        _emit(out_fd, {"rooflines": rl})
        return

    # ---- coarse matcher: configs[1] ---------------------------------------------------------------
    cfg = loftr_coarse_only_config(0.2)
    matcher = HipLoFTR(cfg)
    matcher.load_state_dict(random_state_dict(loftr_param_spec(cfg), 0), strict=True)
    matcher = matcher.eval().to(dev)
    data = synth.to_device(synth.coarse_pair_batch(args.batch, 480, 640, seed=1000 + 100 * rank), dev)
    n_matches = [0]

    def coarse_step():
        d = dict(data)
        matcher(d)
        table = torch.cat([d["mkpts0_f"], d["mkpts1_f"], d["mconf"][:, None]], -1)
        if distributed:
            gathered = ddist.all_gather_tables([table])
            n_matches[0] = sum(t.shape[0] for t in gathered)
        else:
            n_matches[0] = table.shape[0]

    dt = timed_steps(coarse_step, args.steps, args.warmup, distributed)
    pairs_per_s = args.batch * world * args.steps / dt

    # stage breakdown of one coarse step (events, rank 0 only, outside the timed region)
    breakdown = {}
    if rank == 0:
        P = matcher._packed or matcher._pack()
        imgs = torch.cat([data["image0"], data["image1"]], 0)
        with torch.no_grad():
            breakdown["backbone_ms"] = event_time_ms(lambda: matcher._backbone_hip(imgs, P), 5, 1)
            c = matcher._backbone_hip(imgs, P).flatten(1, 2)
            pe = matcher._pe_tokens((60, 80))
            f0, f1 = c[:args.batch], c[args.batch:]
            breakdown["transformer_ms"] = event_time_ms(lambda: matcher._transformer(f0, f1, P, pe, pe), 5, 1)
            matcher._transformer(f0, f1, P, pe, pe)
            g0, g1 = matcher._feat_split          # the split planes the last LayerNorm wrote
            breakdown["coarse_match_ms"] = event_time_ms(
                lambda: ops.coarse_match(g0, g1, (60, 80), (60, 80), 0.2, 2, 0.1), 5, 1)
        del imgs, c, f0, f1, g0, g1

    # ---- refinement head: configs[2] --------------------------------------------------------------
    rcfg = multiview_refinement_config()
    refiner = HipMultiviewMatcher(rcfg, test=True)
    refiner.load_state_dict(random_state_dict(multiview_param_spec(rcfg), 1), strict=True)
    refiner = refiner.eval().to(dev)
    rdata = synth.to_device(synth.refine_bag(args.tracks, 5, 480, 640, seed=2000 + 100 * rank), dev)

    def refine_step():
        d = dict(rdata)
        refiner(d)
        if distributed:
            rows = torch.cat([d["query_points_refined"][0], d["reference_points_refined"][-1][0].reshape(-1, 2)], 0)
            ddist.all_gather_tables([rows])

    r_steps = max(2, args.steps // 2)
    rdt = timed_steps(refine_step, r_steps, min(args.warmup, 2), distributed)
    tracks_per_s = args.tracks * world * r_steps / rdt

    result = {
        "metric": "coarse_image_pairs_per_sec", "value": pairs_per_s, "unit": "image-pairs/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1000.0 * dt / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32 (fp16x2-split operands, fp32 accumulate)", "data": "synthetic",
        "config": {"workload": f"configs[1]: LoFTR coarse_only, 640x480, batch {args.batch} pairs per GPU per step, "
                               "seeded random weights, inputs resident in HBM; match-table all-gather per step when N>1",
                   "parallelism": f"pairs sharded over {world} rank(s), no data-path collective"},
        "secondary": {"metric": "refinement_tracks_per_sec", "value": tracks_per_s, "unit": "tracks/s",
                      "steps": r_steps, "ms_per_step": 1000.0 * rdt / r_steps,
                      "workload": f"configs[2]: MultiviewMatcher, {args.tracks} tracks x 5 views, 640x480 RGB, W=15, crop 35"},
        "matches_last_step": n_matches[0],
        "breakdown": breakdown,
    }
    if rank == 0 and not args.no_rooflines:
        rl = kernel_rooflines(dev, args.batch)
        result["rooflines"] = rl
        dom = rl[0]     # conv_gemm_sf: by far the largest share of both steps (profiles/r01_*_step_kernel_stats.csv)
        result["roofline"] = {k: dom[k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic")}
        result["roofline"]["kernel"] = dom["kernel"]
        result["roofline"]["mfma_flops_executed_frac"] = dom["mfma_flops_executed_frac"]
        # HBM traffic of that launch cannot be counted from inside this process: it comes from the separate
        # rocprofv3 --pmc passes committed under profiles/ (same kernel, same shape), corrected as the MI355X
        # guide prescribes (FETCH_SIZE x2 for 16-byte/lane streaming reads on gfx950).
        pmc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_pmc_conv_gemm_sf_same.json")
        if os.path.exists(pmc):
            with open(pmc) as fh:
                pj = json.load(fh)
            result["roofline"]["traffic"] = pj["hbm_bytes_per_launch"]
            result["roofline"]["traffic_unit"] = "bytes per launch (PMC: 2*FETCH_SIZE + WRITE_SIZE)"
            result["roofline"]["algorithmic_bytes"] = pj["algorithmic_bytes_per_launch"]
            result["roofline"]["traffic_source"] = "profiles/r01_pmc_conv_gemm_sf_same.json"
            dom["traffic"] = pj["hbm_bytes_per_launch"]
    if rank == 0 and not args.no_rooflines:
        # per-kernel HBM traffic of the other hand-written kernels, from the committed PMC passes of
        # `bench.py --kernels-only` (2*FETCH_SIZE + WRITE_SIZE per launch, summed over the kernels of an entry point)
        pmc_all = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_pmc_kernels_only_traffic.json")
        if os.path.exists(pmc_all):
            with open(pmc_all) as fh:
                rows = json.load(fh)["kernels"]
            groups = {"linear_attention": ("la_kv_partial", "la_kv_finalize", "la_apply"), "roi_align": ("roi_align_kernel",),
                      "fine_match": ("fine_match_kernel",), "coarse_match_split": ("cm_gemm_sf", "cm_reduce_stats", "cm_select", "cm_compact"),
                      "coarse_match_f32": ("cm_gemm<",)}
            for r in result["rooflines"]:
                for key, subs in groups.items():
                    if r["kernel"].startswith(key):
                        mb = sum(k["fetch_MB_x2"] + k["write_MB"] for k in rows if any(s_ in k["kernel"] for s_ in subs))
                        r["traffic"] = mb * 1024 * 1024
                        r["traffic_unit"] = "bytes per call (PMC: 2*FETCH_SIZE + WRITE_SIZE, profiles/r01_pmc_kernels_only_traffic.json)"
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline()
    if rank == 0:
        _emit(out_fd, result)
    if distributed:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
