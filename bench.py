#!/usr/bin/env python
"""Hot-path benchmark (driver contract: prints ONE JSON line on rank 0).

Default workload (``--workload pairs``): a *step* = one pass of the coarse matcher over one batch of 8 synthetic
640x480 pairs that are already resident in HBM (BASELINE.json configs[1]); ``value`` = image pairs per second over
all ranks.  The same run also times the refinement head (configs[2]: 2000 tracks x 5 views) and reports it under
``secondary``.  Four distinct resident batches rotate through the steps; the weights are the seeded "planted" set
(params.planted_loftr_state_dict), so every step ends with a real match table (~3600 rows per pair at thr 0.2).
Pairs / track bags shard across ranks with no data-path collective (weak scaling: every rank runs its own batch);
with N>1 every step ends with the path's one real exchange, the gather of the match tables to the merging rank (RCCL over xGMI).

``--gpus N`` without a launcher (WORLD_SIZE unset) re-executes itself under ``torch.distributed.run`` with N ranks
on 127.0.0.1; under the driver's own torchrun it just reads RANK / LOCAL_RANK / WORLD_SIZE.

Other workloads (explicit, not the headline): ``--workload scene300`` = configs[3] (300 images, 44 850 exhaustive
pairs sharded over the ranks, backbone once per image, gather-to-root of the tables, keypoint merge on rank 0);
``--workload hires832`` = configs[4] (832x832 pairs + one 16 000-track refinement chunk); ``--workload eth3d1600`` /
``demo1200`` = the frame sizes the reference's shipped configs feed the matcher (1600x1064: L = 26 600; 1200x800: L = 15 000);
``--workload matchformer`` / ``aspanformer`` = the alternative coarse matchers of SURVEY 8(f) at the configs[1] frame size.

Extra objects: ``roofline`` (dominant hand-written kernel, measured live with events on the launch stream),
``rooflines`` (all hand-written kernels), ``step_roofline`` (whole-step algorithmic flops), ``breakdown`` (stage
times), and on rank 0 at N=1 ``cpu_baseline`` (the oracle = CPU port of the reference path, host cores).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from detectorfreesfm_amd import HipLoFTR, HipMultiviewMatcher, ops, plugin, synth  # noqa: E402
from detectorfreesfm_amd import dist as ddist  # noqa: E402
from detectorfreesfm_amd.config import loftr_coarse_only_config, multiview_refinement_config  # noqa: E402
from detectorfreesfm_amd.params import (loftr_param_spec, multiview_param_spec, planted_loftr_state_dict,  # noqa: E402
                                        random_state_dict)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
MFMA_F32_PEAK_TF = 157.3     # fp32-input MFMA dense peak
MFMA_F16_PEAK_TF = 2500.0    # fp16/bf16 dense MFMA peak (spec; 2178-2382 TF measured micro-benchmarks)
N_RESIDENT = 4               # distinct resident input batches rotated through the steps
# DFSFM_BENCH_DRYRUN=cpu: the SAME step loops (shards, barriers, table gather, max-reduce over ranks, the JSON line) over gloo on the CPU
# stand-ins of tests/cpu_standins.py with toy frame sizes -- a plumbing rehearsal of the multi-GPU launch for machines without GPUs
# (tests/test_dist_gloo.py runs it at world 8).  Its line says "dry-run" in `data` and is never a measurement.
DRYRUN = os.environ.get("DFSFM_BENCH_DRYRUN") == "cpu"
FRAME_HW = (96, 128) if DRYRUN else (480, 640)


def _sync():
    if not DRYRUN:
        torch.cuda.synchronize()

# algorithmic flops (hook-counted on the reference): per 640x480 image / pair / 5-view track
BACKBONE_FLOP_PER_IMAGE = 163.4e9      # ResNetFPN_8_2 without the dead FPN top-down branch (327 G per pair)
TRANSFORMER_FLOP_PER_PAIR = 103.0e9    # 16 encoder-layer applications x 6.45 G
MATCH_FLOP_PER_PAIR = 11.8e9           # 2*L*S*C, counted once
REFINE_FLOP_PER_TRACK = 6.6e9          # the REFERENCE's work: S2DNet 5.1 G + transformer 1.47 G + fine correlation 0.011 G
# what the kernels execute for the same result: S2DNet 0.61 G per patch x 5 (adaptation-0 only on the (W+4)^2 centre crop, bicubic
# window only at the W x W centre: output-identical dead-work skips) + transformer + correlation
REFINE_FLOP_PER_TRACK_EXECUTED = 5 * 0.61e9 + 1.47e9 + 0.011e9


def event_time_ms(fn, iters=10, warmup=2, rounds=3):
    """Average duration of fn() with events on the current (= launch) stream: the median of ``rounds`` averages over
    ``iters`` launches each, so that one allocator / page-fault hiccup (seen once: 60 ms inside a 10-launch window) does not
    become a kernel's reported time."""
    for _ in range(warmup):
        fn()
    avgs = []
    for _ in range(rounds):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            fn()
        e.record()
        e.synchronize()
        avgs.append(s.elapsed_time(e) / iters)
    return sorted(avgs)[len(avgs) // 2]


def timed_steps(step, steps, warmup, distributed, drain=None):
    """``drain``: called once after the last step INSIDE the timed region (a pipelined loop collects its last batch there)."""
    for i in range(warmup):
        step(i)
    if drain is not None:
        drain()
    if distributed:
        torch.distributed.barrier()
    _sync()
    t0 = time.perf_counter()
    for i in range(steps):
        step(warmup + i)
    if drain is not None:
        drain()
    if distributed:
        torch.distributed.barrier()
    _sync()
    dt = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if DRYRUN else "cuda")
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def _rl(kernel, bound, work, ms, units, **extra):
    """One roofline entry: ``work`` = algorithmic flops (mfma) or bytes (hbm) of the timed call."""
    if bound == "mfma":
        ach, peak, unit = work / ms / 1e9, extra.pop("peak", MFMA_F16_PEAK_TF), "TFLOP/s"
    else:
        ach, peak, unit = work / ms / 1e6, HBM_PEAK_GBS, "GB/s"
    d = {"kernel": kernel, "bound": bound, "achieved": ach, "peak": peak, "unit": unit, "frac": ach / peak,
         "traffic": None, "ms": ms, "units": units}
    d.update(extra)
    return d


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
    out.append(_rl("conv_gemm_sf_same_kernel<128,3> (3x3, 128->128 @240x320)", "mfma", flops, ms, f"{nimg} images",
                   mfma_flops_executed_frac=3.0 * flops / ms / 1e9 / MFMA_F16_PEAK_TF,
                   step_share="~47% of the coarse step and ~27% of the refinement step (profiles/r04_*_step_kernel_stats.csv)"))
    del x, xs, pw
    # K10 + K1 of the refinement head: one fused encoder layer (csrc/encoder_fused.hip) on the query tokens of a 2000-track x
    # 4-view bag: enc_kv_kernel (source tokens -> per-track attention state) + enc_apply_kernel (the rest of the layer).
    # Algorithmic work per token row: apply 2 * (128*128 q + 128*16 attention + 128*128 merge + 256*256 mlp.0 + 256*128 mlp.2)
    # = 266 240 flop and 1024 B (row read + row written as fp16x2 planes); kv 2 * 128 * 256 + 2 * 16 * 128 = 69 632 flop, 512 B.
    # The kernel is bound by MFMA issue (3 MFMAs per product, one wave per SIMD) long before HBM: priced against the MFMA peak.
    from detectorfreesfm_amd import coarse as _coarse
    T, Vq, WW, C = 2000, 4, 225, 128
    rows = T * Vq * WW
    wsd = {n: torch.randn(sh, generator=g) * sc for n, sh, sc in (("q_proj.weight", (C, C), .12), ("k_proj.weight", (C, C), .12),
           ("v_proj.weight", (C, C), .12), ("merge.weight", (C, C), .12), ("mlp.0.weight", (2 * C, 2 * C), .09),
           ("mlp.2.weight", (C, 2 * C), .09))}
    for nm in ("norm1", "norm2"):
        wsd[nm + ".weight"], wsd[nm + ".bias"] = torch.ones(C), torch.zeros(C)
    lw = _coarse.EncoderLayerWeights(lambda n: wsd[n].to(dev), "")
    qs = ops.SplitAct.empty_rows((T, Vq * WW), C, dev)
    ops.split_rows(torch.randn((T, Vq * WW, C), generator=g).to(dev), None, out_split=qs)
    qo = ops.SplitAct.empty_rows((T, Vq * WW), C, dev)
    st = ops.encoder_kv(qs, lw.fused)
    ms = event_time_ms(lambda: ops.encoder_apply(qs, lw.fused, st, Vq * WW, out_split=qo))
    out.append(_rl("enc_apply_kernel (fused encoder layer, d_model 128: q, attention, merge, LayerNorm, MLP, LayerNorm, residual)",
                   "mfma", rows * 266240.0, ms, f"{rows} token rows", hbm_GBps_of_rows=rows * 1024.0 / ms / 1e6,
                   mfma_flops_executed_frac=3.0 * rows * 266240.0 / ms / 1e9 / MFMA_F16_PEAK_TF,
                   step_share="~30% of the refinement step together with enc_kv_kernel (was 48%: five GEMM + three attention launches)"))
    ms = event_time_ms(lambda: ops.encoder_kv(qs, lw.fused))
    out.append(_rl("enc_kv_kernel (fused k|v projection + phi(K)^T V per track)", "mfma", rows * 69632.0, ms, f"{rows} token rows",
                   hbm_GBps_of_rows=rows * 512.0 / ms / 1e6))
    del qs, qo, st, lw
    # K9 front end (r06): conv1_1 -> conv1_2 -> centre window + max-pool of S2DNet in one launch, 10 000 patches of 35 x 35
    # (2000 tracks x 5 views).  Algorithmic work per patch: 1225 pixels x 2 x (27 x 64 + 576 x 64) flop; 14.7 KB read, 175 KB written.
    gp = torch.Generator().manual_seed(5)
    npatch = 2000 * 5
    px = (torch.randn((npatch, 35, 35, 3), generator=gp) * 1.3).to(dev)
    fw = ops.S2dFrontWeights((torch.randn((64, 3, 3, 3), generator=gp) * 0.27).to(dev), (torch.randn((64,), generator=gp) * 0.1).to(dev),
                             (torch.randn((64, 64, 3, 3), generator=gp) * 0.06).to(dev), (torch.randn((64,), generator=gp) * 0.1).to(dev))
    ms = event_time_ms(lambda: ops.s2d_front(px, fw, 8, 27))
    fl = npatch * 1225 * 2.0 * (27 * 64 + 576 * 64)
    out.append(_rl("s2d_front_kernel (S2DNet conv1_1 -> conv1_2 -> centre window + max-pool, one launch)", "mfma", fl, ms,
                   f"{npatch} patches of 35 x 35", mfma_flops_executed_frac=3.0 * fl / ms / 1e9 / MFMA_F16_PEAK_TF,
                   hbm_GBps_of_outputs=npatch * (19 * 19 + 18 * 18) * 64 * 4.0 / ms / 1e6,
                   step_share="~9% of the refinement step (the three launches it replaces: 16%)"))
    del px, fw
    # K2 + K1 of the COARSE transformer (d_model 256): one cross-layer application on batch x 4800 query tokens =
    # k|v projection (linear_gemm_sf_kernel, N = 512) + attention state (la_kv_partial_staged + enc256_image_kernel) +
    # enc256_apply_kernel.  Algorithmic work per query row: 2 * (256*256 q + 256*32 attention + 256*256 merge + 512*512 mlp.0
    # + 512*256 mlp.2) = 1 064 960 flop, 2048 B (row read + written as fp16x2 planes); per source row 2 * 256 * 512 = 262 144
    # flop for k|v (+ 2 * 32 * 256 for phi(K)^T V).  Beside it the five-GEMM + K1 path it replaces (coarse.FUSED_ENCODER256 = False).
    C = 256
    rows = batch * 4800
    wsd = {n: torch.randn(sh, generator=g) * sc for n, sh, sc in (("q_proj.weight", (C, C), .09), ("k_proj.weight", (C, C), .09),
           ("v_proj.weight", (C, C), .09), ("merge.weight", (C, C), .09), ("mlp.0.weight", (2 * C, 2 * C), .06),
           ("mlp.2.weight", (C, 2 * C), .06))}
    for nm in ("norm1", "norm2"):
        wsd[nm + ".weight"], wsd[nm + ".bias"] = torch.ones(C), torch.zeros(C)
    lw = _coarse.EncoderLayerWeights(lambda n: wsd[n].to(dev), "")
    xs2 = ops.SplitAct.empty_rows((batch, 4800), 2 * C, dev)
    ops.split_rows(torch.randn((batch, 4800, C), generator=g).to(dev), None, out_split=xs2.cols(0, C))
    src = ops.SplitAct.empty_rows((batch, 4800), C, dev)
    ops.split_rows(torch.randn((batch, 4800, C), generator=g).to(dev), None, out_split=src)
    xo = ops.SplitAct.empty_rows((batch, 4800), C, dev)
    layer_flop = rows * (1064960.0 + 262144.0 + 16384.0)
    if lw.fused256 is not None:
        kv = ops.linear(src, lw.pkv).view(batch, 4800, 2 * C)
        st = ops.encoder256_state(kv[..., :C], kv[..., C:])
        ms = event_time_ms(lambda: ops.encoder256_apply(xs2.cols(0, C), lw.fused256, st, 4800, out_split=xo))
        out.append(_rl("enc256_apply_kernel (fused encoder layer, d_model 256: q, attention, merge, LayerNorm, MLP, LayerNorm, residual)",
                       "mfma", rows * 1064960.0, ms, f"{rows} token rows", hbm_GBps_of_rows=rows * 2048.0 / ms / 1e6,
                       mfma_flops_executed_frac=3.0 * rows * 1064960.0 / ms / 1e9 / MFMA_F16_PEAK_TF))
        ms = event_time_ms(lambda: ops.encoder256_state(kv[..., :C], kv[..., C:]))
        out.append(_rl("encoder256_state (la_kv_partial_staged<32> + enc256_image_kernel)", "hbm", rows * 2048.0, ms,
                       f"{rows} source rows (k, v fp32 read once)"))
        ms = event_time_ms(lambda: ops.linear(src, lw.pkv))
        out.append(_rl("linear_gemm_sf_kernel (k|v projection, K = 256, N = 512)", "mfma", rows * 262144.0, ms, f"{rows} rows"))
        if lw.fused256.kv_stream is not None:
            ms = event_time_ms(lambda: ops.encoder256_kv(src, lw.fused256))
            out.append(_rl("enc256_kv_kernel + enc256_image_kernel (k|v projection fused with phi(K)^T V partial sums)", "mfma",
                           rows * (262144.0 + 16384.0), ms, f"{rows} source rows", hbm_GBps_of_rows=rows * 1024.0 / ms / 1e6))
        ms = event_time_ms(lambda: _coarse.encoder_layer_split(lw, xs2, src, None, xo, 8))
        out.append(_rl("coarse encoder layer application, fused (enc256_kv + image + enc256_apply: 3 launches)", "mfma", layer_flop, ms,
                       f"{rows} rows, cross layer"))
        del kv, st
    f256, lw.fused256 = lw.fused256, None
    ms = event_time_ms(lambda: _coarse.encoder_layer_split(lw, xs2, src, None, xo, 8))
    out.append(_rl("coarse encoder layer application, five GEMMs + K1 (8 launches; linear_gemm_sf_kernel, linear_ln160_kernel, la_*)",
                   "mfma", layer_flop, ms, f"{rows} rows, cross layer"))
    lw.fused256 = f256
    # the LayerNorm-fused linears of that path on their own (mlp.2 + norm2 + residual: K = 512, N = 256)
    hsp = ops.SplitAct.empty_rows((rows,), 2 * C, dev)
    ops.split_rows(torch.randn((rows, 2 * C), generator=g).to(dev), None, out_split=hsp)
    ms = event_time_ms(lambda: ops.linear_ln(hsp, lw.p2, lw.n2[0], lw.n2[1], residual=xs2.cols(0, C), out_split=xo))
    out.append(_rl("linear_ln160_kernel / conv_gemm_sf_same_kernel<256,1,2> (mlp.2 + LayerNorm2 + residual, K = 512)", "mfma",
                   rows * 2.0 * 512 * 256, ms, f"{rows} rows"))
    del xs2, src, xo, lw, hsp
    # the stride-2 3x3 convolution of the backbone still on the lock-step kernel (layer2.0.conv1: 128 -> 196 @240x320 -> 120x160)
    xs = ops.SplitAct.empty(nimg, 240, 320, 128, dev)
    ops.split_rows(torch.randn((nimg, 240, 320, 128), generator=g).to(dev), None, out_split=xs)
    pw = ops.PackedDense(torch.randn((196, 128, 3, 3), generator=g).to(dev) * 0.03, torch.zeros(196, device=dev), cin_pad=128)
    ms = event_time_ms(lambda: ops.conv2d_nhwc(xs, pw, 2, 1, relu=True, out_split=True))
    out.append(_rl("conv_gemm_sf_kernel<128> (3x3 stride 2, 128->196 @240x320)", "mfma", 2.0 * nimg * 120 * 160 * 196 * 9 * 128, ms,
                   f"{nimg} images"))
    del xs, pw
    # S2DNet conv1_2 (3x3, 64 -> 64 on 35x35 patches): the 512 x 64 tile
    M = 10000
    ps = ops.SplitAct.empty(M, 35, 35, 64, dev)
    ops.split_rows(torch.randn((M * 35 * 35, 64), device=dev), None,
                   out_split=ops.SplitAct(ps.hi.view(-1, 64), ps.lo.view(-1, 64), 64))
    pw = ops.PackedDense(torch.randn((64, 64, 3, 3), generator=g).to(dev) * 0.04, torch.zeros(64, device=dev), cin_pad=64, tap_padded=True)
    ms = event_time_ms(lambda: ops.conv2d_nhwc(ps, pw, 1, 1, relu=True, out_split=True), 5, 1)
    out.append(_rl("conv_gemm_sf_same_kernel<64,3,8> (S2DNet conv1_2: 3x3, 64->64 @35x35)", "mfma", 2.0 * M * 35 * 35 * 64 * 9 * 64, ms,
                   f"{M} patches", hbm_GBps=M * 35 * 35 * 64 * 4.0 * 2 / ms / 1e6))
    del ps, pw
    # K3+K4+K5 at batch x (4800 x 4800 x 256): algorithmic flops 2*L*S*C per pair
    L = S = 4800
    f0, f1 = synth.correlated_features(batch, L, S, 256, 7, 0.1)
    f0, f1 = f0.to(dev), f1.to(dev)
    s0, s1 = ops.SplitAct.empty_rows((batch, L), 256, dev), ops.SplitAct.empty_rows((batch, S), 256, dev)
    ops.split_rows(f0, None, out_split=s0)
    ops.split_rows(f1, None, out_split=s1)
    ms = event_time_ms(lambda: ops.coarse_match(s0, s1, (60, 80), (60, 80), 0.2, 2, 0.1))
    flops = 2.0 * L * S * 256 * batch
    out.append(_rl("coarse_match_split (split-plane correlation + dual-softmax + mutual-NN + compaction)", "mfma", flops, ms,
                   f"{batch} pairs"))
    ms32 = event_time_ms(lambda: ops.coarse_match(f0, f1, (60, 80), (60, 80), 0.2, 2, 0.1))
    out.append(_rl("coarse_match_f32 (cm_gemm x2 + select, fp32 matrix path; generic entry point)", "mfma", flops, ms32,
                   f"{batch} pairs", peak=MFMA_F32_PEAK_TF))
    del s0, s1
    # K1 at the coarse shape, N = 2*batch (both images of every pair in one call): 4*N*L*H*D*4 bytes
    N = 2 * batch
    q, k, v = (torch.randn((N, L, 8, 32), generator=g).to(dev) for _ in range(3))
    ms = event_time_ms(lambda: ops.linear_attention(q, k, v))
    out.append(_rl("linear_attention D=32", "hbm", 4.0 * N * L * 256 * 4, ms, f"{N} x 4800 tokens"))
    del q, k, v, f0, f1
    # K8: 2000 tracks x 5 views of 3x35x35 patches; algorithmic bytes = written + read per patch
    M = 10000
    img = torch.rand((1, 3, 480, 640), generator=g).to(dev)
    pts = (torch.rand((M, 2), generator=g) * torch.tensor([580.0, 420.0]) + 30).to(dev)
    boxes = torch.cat([pts - 17, pts + 17], -1)
    buf = torch.empty((M, 3, 35, 35), device=dev)
    ms = event_time_ms(lambda: ops.roi_align(img, boxes, 35, 35, out=buf))
    del buf
    out.append(_rl("roi_align 3x35x35", "hbm", M * (3 * 35 * 35 * 4 + 3 * 36 * 36 * 4.0), ms, f"{M} patches"))
    # K11+K12: 2000 tracks, 4 query views, W=15, C=128.  Bytes the algorithm needs per track: the (V-1) query windows
    # + the 49 candidate rows of the reference window (left=7) = (4*225 + 49)*128*4 = 486 KB  (SURVEY 8d quotes the
    # full reference window, 576 KB; the other 176 rows are never part of the result)
    T, Vq, WW, C = 2000, 4, 225, 128
    ref = torch.randn((T, WW, C), generator=g).to(dev)
    qry = torch.randn((T, Vq, WW, C), generator=g).to(dev)
    mask = torch.ones((T, Vq), dtype=torch.bool, device=dev)
    # the product feeds it the fp16x2-split planes the last transformer layer writes (same bytes per value as fp32)
    rs, qs = ops.SplitAct.empty_rows((T, WW), C, dev), ops.SplitAct.empty_rows((T, Vq, WW), C, dev)
    ops.split_rows(ref.view(-1, C), out_split=ops.SplitAct(rs.hi.view(-1, C), rs.lo.view(-1, C), C))
    ops.split_rows(qry.view(-1, C), out_split=ops.SplitAct(qs.hi.view(-1, C), qs.lo.view(-1, C), C))
    ms = event_time_ms(lambda: ops.fine_match(rs, qs, mask, None, 15, 7))
    out.append(_rl("fine_match W=15", "hbm", T * (Vq * WW + 49) * C * 4.0, ms, f"{T} tracks, split-plane input"))
    ms = event_time_ms(lambda: ops.fine_match(ref, qry, mask, None, 15, 7))
    out.append(_rl("fine_match W=15 (fp32 input)", "hbm", T * (Vq * WW + 49) * C * 4.0, ms, f"{T} tracks"))
    return out


def real_module_ratio():
    """Time of the oracle port divided by the time of the REAL reference modules on the same inputs, from the committed record of
    tools/time_reference_vs_restate.py (the reference tree is not on the GPU box): how far ``cpu_baseline`` (kind "port") is from
    what the reference itself would score on these cores."""
    import re
    for name in ("r06_reference_vs_restate_cpu.txt", "r05_reference_vs_restate_cpu.txt"):
        path = os.path.join(ROOT, "profiles", name)
        if os.path.exists(path):
            r = re.findall(r"port / reference time ratio\s+([0-9.]+)", open(path).read())
            if len(r) >= 2:
                return {"coarse": float(r[0]), "refinement": float(r[1]), "source": f"profiles/{name}",
                        "meaning": "oracle-port time / real-module time on the build container's 8 cores (< 1: the port is faster)"}
    return None


def cpu_baseline():
    """The oracle (CPU port of the reference PyTorch path, same weights / inputs) on the host cores, BASELINE.md section 2
    protocol: 1 warm-up + >= 5 timed 640x480 pairs one at a time (the reference runs batch 1) and one bag of 200 tracks x 5
    views.  The thread count is swept first (a 128-thread pool is slower than 16-32 threads on these shapes: the convolutions
    of one pair do not feed that many cores) and the fastest setting is the one reported, with its core count."""
    from oracle import restate
    all_threads = torch.get_num_threads()
    cand = sorted({t for t in (8, 16, 32, 64, all_threads) if t <= all_threads})
    cfg = loftr_coarse_only_config(0.2)
    sd = planted_loftr_state_dict(loftr_param_spec(cfg), 0)
    data = synth.coarse_pair_batch(1, 480, 640, seed=1000)
    rcfg = multiview_refinement_config()
    rsd = random_state_dict(multiview_param_spec(rcfg), 1)
    rprobe = synth.refine_bag(T=24, V=5, H=480, W=640, seed=2001)
    rdata = synth.refine_bag(T=200, V=5, H=480, W=640, seed=2000)

    def timed(fn):
        t0 = time.perf_counter()
        fn()
        return time.perf_counter() - t0
    sweep_c, sweep_r = {}, {}
    try:
        with torch.no_grad():
            restate.loftr_coarse_forward(sd, cfg, data)                          # warm-up (allocator, thread pool)
            for t in cand:
                torch.set_num_threads(t)
                sweep_c[t] = timed(lambda: restate.loftr_coarse_forward(sd, cfg, data))
            best_c = min(sweep_c, key=sweep_c.get)
            torch.set_num_threads(best_c)
            n, t0 = 0, time.perf_counter()
            while n < 5:
                restate.loftr_coarse_forward(sd, cfg, data)
                n += 1
            pairs_per_s = n / (time.perf_counter() - t0)
            restate.multiview_matcher_forward(rsd, rcfg, rprobe)
            for t in cand:
                torch.set_num_threads(t)
                sweep_r[t] = timed(lambda: restate.multiview_matcher_forward(rsd, rcfg, rprobe))
            best_r = min(sweep_r, key=sweep_r.get)
            torch.set_num_threads(best_r)
            tracks_per_s = 200 / timed(lambda: restate.multiview_matcher_forward(rsd, rcfg, rdata))
    finally:
        torch.set_num_threads(all_threads)
    return {"value": pairs_per_s, "unit": "image-pairs/s", "cores": best_c, "kind": "port",
            "sample": f"{n} single 640x480 pairs through oracle.restate.loftr_coarse_forward (incl. the FPN branch the "
                      f"reference computes and discards) at the fastest of {cand} threads; 1 warm-up",
            "thread_sweep_s_per_pair": {str(k): round(v, 3) for k, v in sweep_c.items()}, "host_threads": all_threads,
            "real_module_ratio": real_module_ratio(),
            "secondary": {"value": tracks_per_s, "unit": "tracks/s", "cores": best_r,
                          "sample": "1 bag of 200 tracks x 5 views through oracle.restate.multiview_matcher_forward "
                                    f"at the fastest of {cand} threads (swept on a 24-track bag)",
                          "thread_sweep_s_per_24_tracks": {str(k): round(v, 3) for k, v in sweep_r.items()}}}


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


def _self_spawn(n_gpus: int) -> int:
    """``bench.py --gpus N`` started without a launcher: run the same command line as N ranks of one node."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def build_coarse(dev, thr=0.2):
    cfg = loftr_coarse_only_config(thr)
    matcher = HipLoFTR(cfg)
    matcher.load_state_dict(planted_loftr_state_dict(loftr_param_spec(cfg), 0), strict=True)
    return matcher.eval().to(dev)


def build_refiner(dev):
    rcfg = multiview_refinement_config()
    refiner = HipMultiviewMatcher(rcfg, test=True)
    refiner.load_state_dict(random_state_dict(multiview_param_spec(rcfg), 1), strict=True)
    return refiner.eval().to(dev)


def load_pmc(result):
    """Attach the HBM traffic measured by the committed rocprofv3 --pmc passes (tools/pmc_collect.py writes
    profiles/r02_pmc_traffic.json; traffic cannot be counted from inside this process).  Corrected as the MI355X
    guide prescribes: 2*FETCH_SIZE (16-byte/lane streaming reads) + WRITE_SIZE, per launch."""
    for name in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json",
                 "r01_pmc_kernels_only_traffic.json"):
        path = os.path.join(ROOT, "profiles", name)
        if os.path.exists(path):
            break
    else:
        return
    with open(path) as fh:
        pmc = json.load(fh)
    rows = pmc["kernels"]
    # the PMC passes are a separate run (counters cannot be read from inside this process): say whether they were collected on
    # the library SOURCES that are being timed now (tools/pmc_collect.py stamps _lib.source_sha256() into the file; the .so itself
    # is not bit-reproducible across build directories)
    from detectorfreesfm_amd import _lib
    result["traffic_build_matches"] = bool(pmc.get("library_source_sha256")) and pmc.get("library_source_sha256") == _lib.source_sha256()
    result["traffic_file"] = f"profiles/{name}"
    groups = {"conv_gemm_sf_same_kernel<128,3>": ("conv_gemm_sf_same_kernel<128, 3", "conv_gemm_sf_same_kernel<128,3"),
              "enc_apply_kernel": ("enc_apply_kernel",), "enc_kv_kernel": ("enc_kv_kernel",), "s2d_front_kernel": ("s2d_front_kernel",),
              "enc256_apply_kernel": ("enc256_apply_kernel",), "enc256_kv_kernel": ("enc256_kv_kernel", "enc256_image_kernel"),
              "linear_attention": ("la_kv_partial", "la_kv_finalize", "la_apply"), "roi_align": ("roi_align_rgb_kernel", "roi_align_kernel"),
              # the two input forms are two template instances: one roofline entry each (r05 summed both under one key)
              "fine_match W=15 (fp32 input)": ("fine_match_kernel<128, false>", "fine_match_kernelILi128ELb0E"),
              "fine_match W=15": ("fine_match_kernel<128, true>", "fine_match_kernelILi128ELb1E"),
              "coarse_match_split": ("cm_gemm_sf", "cm_reduce_stats", "cm_select", "cm_compact", "cm_top", "cm_eval"),
              "coarse_match_f32": ("cm_gemm<",)}
    for r in result["rooflines"]:
        for key, subs in groups.items():            # first match wins: longer keys of one family come first
            if r["kernel"].startswith(key):
                sel = [k for k in rows if any(s_ in k["kernel"] for s_ in subs)]
                if sel:
                    r["traffic"] = sum(k["fetch_MB_x2"] + k["write_MB"] for k in sel) * 1024 * 1024
                    r["traffic_unit"] = f"bytes per call (PMC: 2*FETCH_SIZE + WRITE_SIZE, profiles/{name})"
                    r["traffic_source"] = "committed PMC pass (not measured by this run)"
                break


def run_pairs(args, dev, rank, world, distributed, out_fd):
    """configs[1] (+ configs[2] as ``secondary``)."""
    matcher = build_coarse(dev)
    batches = [synth.to_device(synth.coarse_pair_batch(args.batch, *FRAME_HW, seed=1000 + 1000 * rank + 10 * k), dev)
               for k in range(N_RESIDENT)]
    n_matches = [0]

    def coarse_step(i):
        d = dict(batches[i % N_RESIDENT])
        matcher(d)
        table = torch.cat([d["mkpts0_f"], d["mkpts1_f"], d["mconf"][:, None]], -1)
        if distributed:       # the path's one exchange: the tables go to the merging rank (gather-to-root)
            gathered = ddist.collect_tables([table], root=0, packed=True)      # (rows, counts): one host read per step
            n_matches[0] = gathered[0].shape[0] if gathered is not None else table.shape[0]
        else:
            n_matches[0] = table.shape[0]

    dt = timed_steps(coarse_step, args.steps, args.warmup, distributed)
    pairs_per_s = args.batch * world * args.steps / dt

    # The same K steps as a software pipeline of depth one (``HipLoFTR.forward(data, defer=True)``: the forward is queued without the
    # host read of the match count; step i's table is read after step i + 1 has been launched -- what plugin.match_scene_cached does
    # for the batches of a scene).  ``value`` stays the synchronous plugin call: the reference's caller reads every table right after
    # ``matcher(data)`` (coarse_match_worker.py:83-99); this second rate is what a caller that owns the loop gets (VERDICT r04 #8d).
    waiting = [None]
    piped_rows = [0]

    def collect():
        if waiting[0] is not None:
            d = waiting[0]()
            table = torch.cat([d["mkpts0_f"], d["mkpts1_f"], d["mconf"][:, None]], -1)
            if distributed:
                ddist.collect_tables([table], root=0, packed=True)
            piped_rows[0] = table.shape[0]
            waiting[0] = None

    def piped_step(i):
        d = dict(batches[i % N_RESIDENT])
        fin = matcher(d, defer=True)
        collect()
        waiting[0] = fin
    dt_p = timed_steps(piped_step, args.steps, args.warmup, distributed, drain=collect)

    # stage breakdown of one coarse step (events, rank 0 only, outside the timed region)
    breakdown = {}
    if rank == 0 and not DRYRUN:
        data = batches[0]
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
    refiner = build_refiner(dev)
    bags = [synth.to_device(synth.refine_bag(args.tracks, 5, *FRAME_HW, seed=2000 + 1000 * rank + 10 * k), dev)
            for k in range(N_RESIDENT)]

    def refine_step(i):
        d = dict(bags[i % N_RESIDENT])
        refiner(d)
        if distributed:
            rows = torch.cat([d["query_points_refined"][0], d["reference_points_refined"][-1][0].reshape(-1, 2)], 0)
            ddist.collect_tables([rows], root=0, packed=True)

    r_steps = max(2, args.steps // 2)
    rdt = timed_steps(refine_step, r_steps, min(args.warmup, 2), distributed)
    tracks_per_s = args.tracks * world * r_steps / rdt

    step_flops = args.batch * (2 * BACKBONE_FLOP_PER_IMAGE + TRANSFORMER_FLOP_PER_PAIR + MATCH_FLOP_PER_PAIR)
    result = {
        "metric": "coarse_image_pairs_per_sec", "value": pairs_per_s, "unit": "image-pairs/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1000.0 * dt / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32 (fp16x2-split operands, fp32 accumulate)",
        "data": "dry-run (CPU stand-ins over gloo, toy frames: plumbing only, not a measurement)" if DRYRUN else "synthetic",
        "config": {"workload": f"configs[1]: LoFTR coarse_only, 640x480, batch {args.batch} pairs per GPU per step, "
                               f"{N_RESIDENT} distinct resident batches rotating, seeded planted weights (real match tables "
                               "at thr 0.2), inputs resident in HBM; match-table gather to rank 0 per step when N>1",
                   "parallelism": f"pairs sharded over {world} rank(s), no data-path collective",
                   "rccl_ranks": world if distributed else 0},
        "secondary": {"metric": "refinement_tracks_per_sec", "value": tracks_per_s, "unit": "tracks/s",
                      "steps": r_steps, "ms_per_step": 1000.0 * rdt / r_steps,
                      "workload": f"configs[2]: MultiviewMatcher, {args.tracks} tracks x 5 views, 640x480 RGB, W=15, crop 35, "
                                  f"{N_RESIDENT} distinct resident bags rotating",
                      # priced with the flops the kernels EXECUTE for the reference's result; the reference's own count
                      # (which includes the 35x35 adaptation conv and full-size bicubic the kernels skip) beside it
                      "step_roofline": {"algorithmic_flops_per_step": args.tracks * REFINE_FLOP_PER_TRACK_EXECUTED,
                                        "achieved_tflops": args.tracks * REFINE_FLOP_PER_TRACK_EXECUTED * world * r_steps / rdt / 1e12,
                                        "frac_of_fp16_mfma_peak": args.tracks * REFINE_FLOP_PER_TRACK_EXECUTED * r_steps / rdt / 1e12 / MFMA_F16_PEAK_TF,
                                        "reference_flops_per_step": args.tracks * REFINE_FLOP_PER_TRACK,
                                        "frac_if_priced_with_reference_flops": args.tracks * REFINE_FLOP_PER_TRACK * r_steps / rdt / 1e12 / MFMA_F16_PEAK_TF}},
        # BASELINE's metric has two halves; the second one also at the top level of the line (the full record stays in `secondary`)
        "refinement_tracks_per_sec": tracks_per_s, "refinement_ms_per_step": 1000.0 * rdt / r_steps,
        "matches_last_step": n_matches[0],
        "pipelined": {"value": args.batch * world * args.steps / dt_p, "unit": "image-pairs/s", "ms_per_step": 1000.0 * dt_p / args.steps,
                      "matches_last_step": piped_rows[0],
                      "note": "same steps, same tables, forward(data, defer=True): step i's match count is read after step i+1 was "
                              "launched (no device idle on the host read); `value` is the synchronous plugin call"},
        "step_roofline": {"algorithmic_flops_per_step": step_flops,
                          "achieved_tflops_per_gpu": step_flops * args.steps / dt / 1e12,
                          "frac_of_fp16_mfma_peak": step_flops * args.steps / dt / 1e12 / MFMA_F16_PEAK_TF,
                          "frac_of_fp32_mfma_peak": step_flops * args.steps / dt / 1e12 / MFMA_F32_PEAK_TF,
                          "note": "algorithmic flops counted once (SURVEY 8a/8d); the split scheme executes 3 fp16 MFMA "
                                  "products per algorithmic product"},
        "breakdown": breakdown,
    }
    if rank == 0 and not args.no_rooflines:
        rl = kernel_rooflines(dev, args.batch)
        result["rooflines"] = rl
        load_pmc(result)
        dom = rl[0]     # conv_gemm_sf: by far the largest share of both steps (profiles/*_step_kernel_stats.csv)
        result["roofline"] = {k: dom[k] for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic")}
        result["roofline"]["mfma_flops_executed_frac"] = dom["mfma_flops_executed_frac"]
        result["roofline"]["algorithmic_bytes"] = 2 * args.batch * 240 * 320 * 128 * 4 * 2 + 9 * 128 * 128 * 4
        if "traffic_unit" in dom:
            result["roofline"]["traffic_unit"] = dom["traffic_unit"]
            result["roofline"]["traffic_source"] = dom.get("traffic_source")
        # the refinement step's own dominant hand-written kernels: the stride-1 3x3 convolutions of S2DNet run on the same
        # kernel as the entry above; the transformer is the fused encoder layer
        enc = next(r for r in rl if r["kernel"].startswith("enc_apply_kernel"))
        result["secondary"]["roofline"] = {k: enc[k] for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic",
                                                                 "mfma_flops_executed_frac", "hbm_GBps_of_rows")}
        result["secondary"]["roofline"]["algorithmic_bytes_per_row"] = 1024
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline()
    if rank == 0 and world == 1:
        result["feeding"] = feeding_rate(dev)
    if rank == 0:
        _emit(out_fd, result)


def feeding_rate(dev):
    """Outside the metric (which is defined on resident frames): the rate at which FILES become matcher inputs on the device --
    baseline JPEG decode (csrc/jpeg_decode.hip, batched: one set of launches per seven files) + LANCZOS resize / conversion (csrc/image_resize.hip) --
    beside libjpeg-turbo + Pillow's resize on one host core (what the reference's readers do, src/dataset/utils.py:123-160).
    Synthetic 1600x1200 4:2:0 JPEGs (Pillow writes them here), read at the pipeline's 640-pixel setting.  Never fails the
    bench: any problem is reported as a string."""
    try:
        import io
        import numpy as np
        from PIL import Image
        from detectorfreesfm_amd import images, jpeg
        rng = np.random.default_rng(0)
        bufs = []
        for k in range(4):
            y, x = np.mgrid[0:1200, 0:1600]
            img = (128 + 90 * np.sin(x / (13. + k)) * np.cos(y / (29. - k)) + rng.normal(0, 10, (1200, 1600))).clip(0, 255).astype(np.uint8)
            b = io.BytesIO()
            Image.fromarray(np.stack([img, img[::-1], img[:, ::-1]], -1)).save(b, "JPEG", quality=90, subsampling=2)
            bufs.append(b.getvalue())
        bufs = bufs * 8

        def device_pass():
            return [images.read_grayscale(f, resize=(640,), df=8, device=dev) for f in jpeg.decode_many(bufs, False, dev)]
        device_pass()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        device_pass()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        t0 = time.perf_counter()
        for b in bufs[:4]:
            im = Image.open(io.BytesIO(b))
            im.draft("L", im.size)
            np.asarray(im.resize((640, 480), resample=Image.LANCZOS), dtype=np.float32)
        host = (time.perf_counter() - t0) / 4
        # the decode alone, marker segments already parsed: ONE launching thread, one batched call (dfsfm_jpeg_decode_batch_u8), priced
        # with the bytes a file's decode must move -- scan in, compacted scan out + in, coefficients out + in, pixels out
        plans = [jpeg.plan(b) for b in bufs]
        jpeg.decode_batch(bufs, False, dev, plans=plans)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        jpeg.decode_batch(bufs, False, dev, plans=plans)
        torch.cuda.synchronize()
        dtb = (time.perf_counter() - t0) / len(bufs)
        pl = plans[0]
        nblocks = (1600 // 16) * (1200 // 16) * 6
        alg = 3 * pl.scan.size + 2 * nblocks * 128 + 1600 * 1200
        return {"frames_per_s": len(bufs) / dt,
                "jpeg_decode_batch": {"files_per_s": 1.0 / dtb, "ms_per_file": 1e3 * dtb, "launching_threads": 1,
                                      "roofline": {"bound": "hbm", "achieved": alg / dtb / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                                   "frac": alg / dtb / 1e9 / HBM_PEAK_GBS, "traffic": None,
                                                   "algorithmic_bytes_per_file": alg,
                                                   "note": "latency-bound integer work (a thread's next Huffman symbol depends on its last): the "
                                                           "fraction of HBM says how far from a bandwidth limit it is, not how good it is"},
                                      "huffman_chunks_per_s": float(pl.frame.nchunks) / dtb}, "host_one_core_frames_per_s": 1.0 / host, "frame": "1600x1200 4:2:0 q90 JPEG -> 640x480 fp32",
                "note": "file bytes in host memory -> [1,480,640] fp32 on the device: marker parse on the host, entropy decode + IDCT + "
                        "LANCZOS resize on the GPU (batches of 14 files on two streams, jpeg.decode_many); the host figure is libjpeg-turbo + "
                        "Pillow's resize on one core"}
    except Exception as e:          # noqa: BLE001 -- an extra, never the bench's failure
        return {"error": f"{type(e).__name__}: {e}"}


def run_scene(args, dev, rank, world, distributed, out_fd):
    """configs[3]: ETH3D-shaped scene, exhaustive pairs sharded over the ranks (the analogue of
    src/coarse_match/coarse_match.py:127-140): backbone once per image on every rank, the rank's tile shard of
    the pair list (dist.shard_pairs_tiled) matched from the cached tokens, ONE gather-to-root of the tables, keypoint merge on rank 0."""
    n_img = args.scene_images
    if args.scene_matcher == "aspanformer":       # backbone once per image + PAIRS_PER_PASS pairs per transformer pass
        from detectorfreesfm_amd.aspanformer import HipASpanFormer, aspanformer_coarse_only_config
        from detectorfreesfm_amd.params import aspanformer_param_spec, planted_aspanformer_state_dict
        cfg = aspanformer_coarse_only_config(0.4)
        matcher = HipASpanFormer(cfg)
        matcher.load_state_dict(planted_aspanformer_state_dict(aspanformer_param_spec(cfg), 0), strict=True)
        matcher = matcher.eval().to(dev)
        mname = "ASpanFormer (thr 0.4), "
    else:
        matcher = build_coarse(dev)
        mname = ""
    g = torch.Generator().manual_seed(4242)
    base = torch.rand((1, 1, 480, 640), generator=g)
    # a camera sweep: image k = the base texture rolled by k coarse cells (+ noise) -> every pair has a planted flow
    images = torch.cat([torch.roll(base, shifts=(8 * (k % 7), 8 * k), dims=(2, 3)) + 0.02 * torch.randn((1, 1, 480, 640), generator=g)
                        for k in range(n_img)], 0).to(dev)
    pairs = ddist.exhaustive_pairs(n_img)
    if args.scene_pairs:
        pairs = pairs[:args.scene_pairs]
    order = ddist.shard_pairs_tiled(pairs, n_img, world)      # blocks of the (i, j) plane: ~ n/sqrt(world) images per rank
    mine = [pairs[k] for k in order[rank]]
    torch.cuda.synchronize()
    if distributed:
        torch.distributed.barrier()
    t0 = time.perf_counter()
    tables = plugin.match_scene_cached(matcher, images, mine, batch=args.batch, to_host=False)
    torch.cuda.synchronize()
    t_match = time.perf_counter() - t0
    flat = [tables[p] for p in mine]
    gathered = ddist.collect_tables(flat, root=0, packed=True) if distributed or rank == 0 else None   # merge on rank 0 only
    torch.cuda.synchronize()
    t_gather = time.perf_counter() - t0 - t_match
    n_rows, n_kpts = (int(gathered[0].shape[0]) if gathered is not None else 0), 0
    if rank == 0:
        rows, lens = gathered                                 # rows of all ranks in rank order, row counts per pair
        pid = torch.tensor([pairs[k] for o in order for k in o], dtype=torch.int32, device=dev)
        rep = torch.repeat_interleave(pid, lens.to(torch.int64), dim=0)
        kpts, scores, offsets, ids = ops.merge_keypoints(rows, rep[:, 0].contiguous(), rep[:, 1].contiguous(), n_img)
        n_kpts = int(kpts.shape[0])
    torch.cuda.synchronize()
    if distributed:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        _emit(out_fd, {
            "metric": "coarse_image_pairs_per_sec", "value": len(pairs) / dt, "unit": "image-pairs/s", "n_gpus": world,
            "steps": 1, "warmup": 0, "ms_per_step": 1000.0 * dt, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32 (fp16x2-split operands, fp32 accumulate)", "data": "synthetic",
            "config": {"workload": f"configs[3]: {mname}{n_img} images 640x480, {len(pairs)} exhaustive pairs, backbone once per image, "
                                   "tables gathered to rank 0 once, keypoint merge on rank 0",
                       "parallelism": f"tiled pair shards (blocks of the image x image plane) over {world} rank(s); one gather-to-root of match tables",
                       "rccl_ranks": world if distributed else 0},
            "phases_s": {"match_rank0": t_match, "gather_rank0": t_gather, "total_max_over_ranks": dt},
            "match_rows": n_rows, "keypoints": n_kpts})


# --workload name -> (H, W, pairs per step at --batch 8, refinement chunk, label)
HIRES = {"hires832": (832, 832, 4, 16000, "configs[4]"),
         # what the reference's shipped configs feed the matcher: img_resize 1600 (hydra_configs/eth3d_sfm/dfsfm.yaml:76;
         # ETH3D frames are 3:2 -> 1600x1064 after the df=8 rounding), refinement chunks of 2000 tracks (:65)
         "eth3d1600": (1064, 1600, 2, 2000, "production frame size of hydra_configs/eth3d_sfm/dfsfm.yaml"),
         # img_resize 1200 (src/coarse_match/coarse_match.py:15, hydra_configs/demo/dfsfm.yaml:48)
         "demo1200": (800, 1200, 4, 2000, "production frame size of hydra_configs/demo/dfsfm.yaml")}


def run_hires(args, dev, rank, world, distributed, out_fd):
    """configs[4] (832x832) and the reference's production frame sizes (1600x1064, 1200x800) through the LoFTR coarse
    matcher + one refinement chunk on frames of that size."""
    H, W, nb8, T, label = HIRES[args.workload]
    L = (H // 8) * (W // 8)
    matcher = build_coarse(dev)
    nb = max(1, args.batch * nb8 // 8)
    batches = [synth.to_device(synth.coarse_pair_batch(nb, H, W, seed=500 + 1000 * rank + 10 * k), dev) for k in range(2)]
    n_matches = [0]

    def coarse_step(i):
        d = dict(batches[i % 2])
        matcher(d)
        n_matches[0] = int(d["mconf"].shape[0])
    dt = timed_steps(coarse_step, args.steps, args.warmup, distributed)
    breakdown = {}
    if rank == 0 and not DRYRUN:
        data = batches[0]
        P = matcher._packed or matcher._pack()
        imgs = torch.cat([data["image0"], data["image1"]], 0)
        with torch.no_grad():
            breakdown["backbone_ms"] = event_time_ms(lambda: matcher._backbone_hip(imgs, P), 3, 1)
            c = matcher._backbone_hip(imgs, P).flatten(1, 2)
            pe = matcher._pe_tokens((H // 8, W // 8))
            f0, f1 = c[:nb], c[nb:]
            breakdown["transformer_ms"] = event_time_ms(lambda: matcher._transformer(f0, f1, P, pe, pe), 3, 1)
            matcher._transformer(f0, f1, P, pe, pe)
            g0, g1 = matcher._feat_split
            breakdown["coarse_match_ms"] = event_time_ms(
                lambda: ops.coarse_match(g0, g1, (H // 8, W // 8), (H // 8, W // 8), 0.2, 2, 0.1), 3, 1)
        del imgs, c, f0, f1, g0, g1
    refiner = build_refiner(dev)
    bag = synth.to_device(synth.refine_bag(T, 5, H, W, seed=2500 + rank), dev)

    def refine_step(i):
        refiner(dict(bag))
    r_steps = max(2, args.steps // 4)
    rdt = timed_steps(refine_step, r_steps, 1, distributed)
    if rank == 0:
        flops = nb * (2 * BACKBONE_FLOP_PER_IMAGE * (H * W) / (640 * 480) + 16 * 6.45e9 * L / 4800 + 2.0 * L * L * 256)
        _emit(out_fd, {
            "metric": "coarse_image_pairs_per_sec", "value": nb * world * args.steps / dt, "unit": "image-pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1000.0 * dt / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 (fp16x2-split operands, fp32 accumulate)", "data": "synthetic",
            "config": {"workload": f"{label}: LoFTR coarse_only at {W}x{H} (L = S = {L}), batch {nb} pairs per GPU per step; "
                                   f"refinement chunk_size {T} x 5 views", "parallelism": f"{world} rank(s), weak"},
            "matches_last_step": n_matches[0], "breakdown_ms": breakdown,
            "step_roofline": {"algorithmic_flops_per_step": flops, "achieved_tflops_per_gpu": flops * args.steps / dt / 1e12,
                              "frac_of_fp16_mfma_peak": flops * args.steps / dt / 1e12 / MFMA_F16_PEAK_TF},
            "secondary": {"metric": "refinement_tracks_per_sec", "value": T * world * r_steps / rdt, "unit": "tracks/s",
                          "steps": r_steps, "ms_per_step": 1000.0 * rdt / r_steps,
                          "workload": f"{label}: one {T}-track x 5-view chunk, {W}x{H} RGB frames"}})


def run_alt_matcher(args, dev, rank, world, distributed, out_fd):
    """The reference's two alternative coarse matchers at the configs[1] frame size.  MatchFormer-LA
    takes a batch of pairs per step; ASpanFormer takes one pair per call like the reference (aspanformer.py:43), a step is
    ``--batch`` calls."""
    if args.workload == "matchformer":
        from detectorfreesfm_amd.matchformer import HipMatchformer, matchformer_coarse_only_config
        from detectorfreesfm_amd.params import matchformer_param_spec, planted_matchformer_state_dict
        cfg = matchformer_coarse_only_config(0.4)
        m = HipMatchformer(cfg)
        m.load_state_dict(planted_matchformer_state_dict(matchformer_param_spec(), 0), strict=True)
        per_call, name = args.batch, "MatchFormer-LA large, coarse_only, thr 0.4"
    else:
        from detectorfreesfm_amd.aspanformer import HipASpanFormer, aspanformer_coarse_only_config
        from detectorfreesfm_amd.params import aspanformer_param_spec, planted_aspanformer_state_dict
        cfg = aspanformer_coarse_only_config(0.4)
        m = HipASpanFormer(cfg)
        m.load_state_dict(planted_aspanformer_state_dict(aspanformer_param_spec(cfg), 0), strict=True)
        per_call, name = 1, "ASpanFormer, coarse_only, online_resize, thr 0.4, one pair per call"
    m = m.eval().to(dev)
    calls = args.batch // per_call
    fh, fw = (int(v) for v in args.alt_frame.lower().split("x"))
    batches = [[synth.to_device(synth.coarse_pair_batch(per_call, fh, fw, seed=1000 + 1000 * rank + 100 * k + c), dev)
                for c in range(calls)] for k in range(N_RESIDENT)]
    n_matches = [0]

    def step(i):
        n = 0
        for b in batches[i % N_RESIDENT]:
            d = dict(b)
            m(d)
            n += int(d["mconf"].shape[0])
        n_matches[0] = n
    dt = timed_steps(step, args.steps, args.warmup, distributed)
    if rank == 0:
        _emit(out_fd, {
            "metric": "coarse_image_pairs_per_sec", "value": args.batch * world * args.steps / dt, "unit": "image-pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1000.0 * dt / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 (fp16x2-split operands, fp32 accumulate)", "data": "synthetic",
            "config": {"workload": f"{name}; {fw}x{fh}, {args.batch} pairs per GPU per step, {N_RESIDENT} distinct resident batches "
                                   "rotating, seeded weights on a planted backbone", "parallelism": f"{world} rank(s), weak"},
            "matches_last_step": n_matches[0]})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=8, help="pairs per step per GPU (BASELINE configs[1]: 8)")
    ap.add_argument("--tracks", type=int, default=2000, help="tracks per refinement bag (configs[2]: 2000)")
    ap.add_argument("--workload", choices=("pairs", "scene300", "hires832", "eth3d1600", "demo1200", "matchformer", "aspanformer"), default="pairs")
    ap.add_argument("--scene-images", type=int, default=300)
    ap.add_argument("--scene-pairs", type=int, default=0, help="truncate the exhaustive pair list (0 = all)")
    ap.add_argument("--scene-matcher", choices=("loftr", "aspanformer"), default="loftr", help="coarse matcher of --workload scene300")
    ap.add_argument("--alt-frame", default="480x640", help="HxW of the frames of --workload matchformer / aspanformer (832x832: configs[4])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-rooflines", action="store_true")
    ap.add_argument("--kernels-only", action="store_true",
                    help="only time the hand-written kernels at the bench shapes (compact rocprofv3 target)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(_self_spawn(args.gpus))

    out_fd = _claim_stdout()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # DFSFM_BENCH_FORCE_DIST=1 exercises the RCCL code path (init, barriers, max-reduce, table all-gather) with a
    # single rank; multi-GPU runs have WORLD_SIZE > 1
    distributed = world > 1 or os.environ.get("DFSFM_BENCH_FORCE_DIST") == "1"
    if world != args.gpus:
        # the driver computes scaling from `value` per N: a line labelled N that N ranks did not produce must not exist
        raise SystemExit(f"bench.py --gpus {args.gpus} but {world} rank(s) were launched (WORLD_SIZE): refusing to report")
    if DRYRUN:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from cpu_standins import cpu_ops
        dev = torch.device("cpu")
        if distributed:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
            torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
        args.no_rooflines = args.no_cpu_baseline = True
        with cpu_ops(), torch.no_grad():
            if args.workload != "pairs":
                raise SystemExit("the dry run rehearses --workload pairs")
            run_pairs(args, dev, rank, world, distributed, out_fd)
        if distributed:
            torch.distributed.barrier()
            torch.distributed.destroy_process_group()
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    if world > torch.cuda.device_count():
        raise SystemExit(f"{world} ranks on a node with {torch.cuda.device_count()} GPU(s): one process per GPU")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        torch.distributed.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        probe = torch.ones(1, device=dev)
        torch.distributed.all_reduce(probe)                 # RCCL really is up: the sum must equal the world size
        if int(probe.item()) != world:
            raise SystemExit(f"RCCL all-reduce over {world} rank(s) returned {int(probe.item())}: refusing to report")
        # one process per GPU: the ranks must sit on distinct devices
        ids = [None] * world
        torch.distributed.all_gather_object(ids, (local_rank, str(torch.cuda.get_device_properties(dev).uuid) if hasattr(
            torch.cuda.get_device_properties(dev), "uuid") else str(local_rank)))
        if len({i[1] for i in ids}) != world:
            raise SystemExit(f"ranks share a GPU: {ids}")

    if args.kernels_only:
        _emit(out_fd, {"rooflines": kernel_rooflines(dev, args.batch)})
    elif args.workload == "pairs":
        run_pairs(args, dev, rank, world, distributed, out_fd)
    elif args.workload == "scene300":
        run_scene(args, dev, rank, world, distributed, out_fd)
    elif args.workload in ("matchformer", "aspanformer"):
        run_alt_matcher(args, dev, rank, world, distributed, out_fd)
    else:
        run_hires(args, dev, rank, world, distributed, out_fd)
    if distributed:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
