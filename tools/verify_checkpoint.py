#!/usr/bin/env python3
"""Maintainer-side check of REAL checkpoints (missing #4): every parity number in this repository is on seeded or
planted weights, because the reference's weights are a Drive download (/root/reference/INSTALL.md:31-32) this build environment
cannot reach.  A person who holds ``outdoor_ds.ckpt`` (LoFTR) and / or ``multiview_matcher.ckpt`` runs

    python tools/verify_checkpoint.py --loftr-ckpt weight/outdoor_ds.ckpt --refine-ckpt weight/multiview_matcher.ckpt \\
        --images SfM_dataset/example_dataset/example_scene/images [--resize 640] [--thr 0.2] [--tracks 96]

on an MI355X and gets, per checkpoint, a PARITY report (HIP plugin vs the CPU oracle under the north_star rules of
tests/parity.py: match indices identical, confidences / refined coordinates within 1e-4, per-entry exemptions listed) and a RANGE
report (the abs-max of every split-plane tensor the first forward produced, incl. the fused encoder layers' register-only
intermediates through their five-GEMM shadow pass -- the fp16x2 representation holds |v| < 65504, and real weights have never been
through it).  Exit code 0 = both reports clean.

What runs:
* coarse: the first two frames of ``--images`` (sorted), decoded to grey and LANCZOS-resized on the host exactly as the reference's
  reader does (src/dataset/utils.py:123-160; longest side ``--resize``, both sides rounded down to multiples of 8), through
  ``plugin.build_model`` (checkpoint layout of loftr.py:83-87: ``state_dict`` with the ``matcher.`` prefix) -> ``detector(data);
  matcher(data)`` -> compared with ``oracle.restate.loftr_coarse_forward`` on the same tensors -- and with the REAL reference
  module (``oracle.ref_import.import_loftr``) as well when the reference tree is present (``DFSFM_REFERENCE_ROOT``).
* refinement: one bag over the first ``--views`` RGB frames: tracks are the coarse matches of the pair when the coarse step ran
  and found enough (real geometry), otherwise seeded synthetic tracks; ``plugin.build_refine_model`` (layout of
  multiview_match_worker.py:40-53) -> compared with ``oracle.restate.multiview_matcher_forward``.

``--cpu-standins`` replaces the HIP kernels by the CPU emulation of tests/cpu_standins.py: it exists so that THIS SCRIPT's
plumbing can be tested without a GPU (tests/test_verify_checkpoint_cpu.py); it verifies nothing about the kernels.
This is test infrastructure: it imports ``oracle`` and ``tests/parity.py`` and is not part of the product path.
"""
import argparse
import contextlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import parity  # noqa: E402
from detectorfreesfm_amd import images as dimages, ops, plugin  # noqa: E402
from detectorfreesfm_amd.config import loftr_coarse_only_config, multiview_refinement_config  # noqa: E402
from oracle import ref_import, restate  # noqa: E402

EXTS = (".jpg", ".jpeg", ".png", ".JPG", ".JPEG", ".PNG")


def list_frames(image_dir):
    names = sorted(f for f in os.listdir(image_dir) if f.endswith(EXTS))
    if len(names) < 2:
        raise SystemExit(f"{image_dir}: need at least two frames ({EXTS})")
    return [os.path.join(image_dir, n) for n in names]


def read_frame(path, resize, color):
    """Host-side restatement of read_grayscale / read_rgb (src/dataset/utils.py:80-160): decode, resize the longest side to
    ``resize`` with PIL LANCZOS, both sides rounded down to multiples of 8, /255.  Returns ([C,h,w] float32, scale (h_o/h, w_o/w))."""
    from PIL import Image, ImageOps
    with Image.open(path) as im:
        im = ImageOps.exif_transpose(im).convert("RGB" if color else "L")
        w, h = im.size
        w_new, h_new = dimages.process_resize(w, h, (resize,), df=8)
        im = im.resize((w_new, h_new), resample=Image.LANCZOS)
        a = np.asarray(im, dtype=np.float32) / 255.0
    t = torch.from_numpy(a)
    t = t.permute(2, 0, 1) if color else t[None]
    return t.contiguous(), torch.tensor([h / h_new, w / w_new], dtype=torch.float32)


def range_report(seen, label, out):
    """seen: [(producer, abs-max)] of a range sweep.  Prints the five largest and the headroom to the split-plane limit."""
    if not seen:
        out(f"  [{label}] range: no split-plane producer reported (sweep off?)")
        return True
    worst = {}
    for n, v in seen:
        v = float(v) if float(v) == float(v) else float("inf")              # NaN: out of range
        worst[n] = max(worst.get(n, 0.0), v)
    top = sorted(worst.items(), key=lambda kv: -kv[1])[:5]
    mx = top[0][1]
    out(f"  [{label}] range: {len(seen)} split-plane tensors checked, max |v| = {mx:.4g} "
        f"(limit 65504, headroom x{65504.0 / max(mx, 1e-30):.3g})")
    for n, v in top:
        out(f"      {v:12.5g}  {n}")
    return mx < 65504.0


def to_dev(d, dev):
    return {k: (v.to(dev) if isinstance(v, torch.Tensor) else [x.to(dev) for x in v] if isinstance(v, list) else v)
            for k, v in d.items()}


def verify_coarse(ckpt, frames, resize, thr, dev, out):
    out(f"== coarse matcher: {ckpt}")
    cfg = loftr_coarse_only_config(thr)
    det, matcher = plugin.build_model({"matcher": "loftr_hip", "type": "coarse_only", "match_thr": thr, "seed": 666,
                                       "loftr_hip": {"weight_path": ckpt, "cfg": cfg}})
    sd = {k.replace("matcher.", "", 1) if k.startswith("matcher.") else k: v
          for k, v in torch.load(ckpt, map_location="cpu")["state_dict"].items()}
    (im0, s0), (im1, s1) = read_frame(frames[0], resize, False), read_frame(frames[1], resize, False)
    data = {"image0": im0[None], "image1": im1[None], "scale0": s0[None], "scale1": s1[None]}
    out(f"  frames {os.path.basename(frames[0])} {tuple(im0.shape[1:])}, {os.path.basename(frames[1])} {tuple(im1.shape[1:])}")
    matcher = matcher.to(dev) if dev.type == "cuda" else matcher
    d = to_dev(data, dev)
    seen = []
    with torch.no_grad(), ops.range_sweep("verify_checkpoint(coarse)", report=seen):
        det(d)
        matcher(d)
    ok = range_report(seen, "coarse", out)
    with torch.no_grad():
        o = restate.loftr_coarse_forward(sd, cfg, data, with_fine_backbone=False)
        conf = restate.dual_softmax_conf(o["feat_c0"], o["feat_c1"], cfg["match_coarse"]["dsmax_temperature"])
    n_h, n_o = int(d["i_ids"].numel()), int(o["i_ids"].numel())
    try:
        ex = parity.check_coarse(d, o, conf, thr)
        parity.check_coarse_rows(d, o, ex)
        both = {(int(b), int(i)): float(c) for b, i, c in zip(o["b_ids"], o["i_ids"], o["mconf"])}
        dc = max((abs(float(c) - both[(int(b), int(i))]) for b, i, c in zip(d["b_ids"].cpu(), d["i_ids"].cpu(), d["mconf"].cpu())
                  if (int(b), int(i)) in both), default=0.0)
        out(f"  [coarse] parity vs oracle: OK -- {n_h} matches (oracle {n_o}), indices identical, max |conf diff| {dc:.2e} "
            f"(tolerance {parity.TOL_CONF:g}), exempted entries: {ex}")
    except AssertionError as e:
        ok = False
        out(f"  [coarse] parity vs oracle: FAILED -- {n_h} matches (oracle {n_o}): {str(e)[:600]}")
    if ref_import.reference_available():
        LoFTR, _ = ref_import.import_loftr()
        ref = LoFTR(cfg).eval()
        ref.load_state_dict(sd, strict=True)
        r = dict(data)
        with torch.no_grad():
            ref(r)
        same = all(torch.equal(r[k], o[k]) for k in ("b_ids", "i_ids", "j_ids")) and torch.equal(r["mconf"], o["mconf"])
        out(f"  [coarse] oracle vs the REAL reference module on this checkpoint: {'bit-identical' if same else 'DIFFERENT'} "
            f"({int(r['i_ids'].numel())} matches)")
        ok = ok and same
    else:
        out("  [coarse] reference tree not present: the oracle stands alone (it is pinned bit-for-bit to the real module on "
            "seeded weights, tests/test_oracle_golden.py)")
    if n_o == 0:
        out("  [coarse] WARNING: the oracle found no match at this threshold -- the parity statement is vacuous; try --thr 0.05")
    table = None
    if n_h:
        table = np.concatenate([d["mkpts0_f"].cpu().numpy(), d["mkpts1_f"].cpu().numpy()], 1)
    return ok, table


def build_bag(frames, resize, views, tracks, table, seed):
    """One bag in the layout of construct_matching_data.py:456-475 (batch dim 1)."""
    imgs, scales = zip(*(read_frame(f, resize, True) for f in frames[:views]))
    V = len(imgs)
    g = torch.Generator().manual_seed(seed)
    h0, w0 = imgs[0].shape[1:]
    real = table is not None and V == 2 and len(table) >= 8
    if real:      # the pair's coarse matches as two-view tracks (original-image pixels)
        sel = torch.randperm(len(table), generator=g)[:tracks].sort()[0].numpy()
        q = torch.from_numpy(table[sel, :2]).float()
        r = torch.from_numpy(table[sel, 2:4]).float()[None]
        T = len(sel)
    else:
        T = tracks
        sc0 = scales[0][[1, 0]]
        q = (torch.rand((T, 2), generator=g) * torch.tensor([w0 - 60.0, h0 - 60.0]) + 30.0) * sc0
        r = torch.stack([(q / sc0 + 2.0 * torch.randn((T, 2), generator=g)) * scales[v][[1, 0]] for v in range(1, V)])
    data = {
        "images": [im[None] for im in imgs], "scales": torch.stack(scales)[None],
        "query_points": q[None], "reference_points_coarse": r[None],
        "query_img_idxs": torch.zeros((1, T), dtype=torch.long),
        "reference_img_idxs": torch.arange(1, V)[None, :, None].expand(1, V - 1, T).contiguous(),
        "track_valid_mask": torch.ones((1, V - 1, T), dtype=torch.bool),
        "scales_relative": torch.ones((1, V, T)), "view_point_vector": torch.zeros((1, V, T, 3)),
        "query_movable_mask": torch.ones((1, T), dtype=torch.bool),
    }
    return data, ("coarse matches of the pair" if real else "seeded synthetic tracks")


def verify_refine(ckpt, frames, resize, views, tracks, table, dev, out):
    out(f"== refinement matcher: {ckpt}")
    cfg = multiview_refinement_config()
    matcher = plugin.build_refine_model({"weight_path": [ckpt], "seed": 666}, None, 0)
    sd = dict(matcher.state_dict())
    data, what = build_bag(frames, resize, views, tracks, table, 7)
    T, V = data["query_points"].shape[1], len(data["images"])
    out(f"  bag: {T} tracks x {V} views ({what})")
    matcher = matcher.to(dev) if dev.type == "cuda" else matcher
    d = to_dev(data, dev)
    seen = []
    with torch.no_grad(), ops.range_sweep("verify_checkpoint(refine)", report=seen):
        matcher(d)
    ok = range_report(seen, "refine", out)
    with torch.no_grad():
        o = restate.multiview_matcher_forward({k: v.cpu() for k, v in sd.items()}, cfg, data)
    left = cfg["multiview_matching_test"]["left_point_movement_window_size"]
    qs = data["scales"][0, data["query_img_idxs"][0]][:, [1, 0]]
    try:
        flips = parity.check_refine(d["query_points_refined"][0], d["reference_points_refined"][-1][0], d["std"][-1][0],
                                    o["query_points_refined"][0], o["reference_points_refined"][0], o["std"][0],
                                    data["track_valid_mask"][0], data["query_points"][0], qs, o["cand_score"], left)
        dq = float((d["query_points_refined"][0].cpu() - o["query_points_refined"][0]).abs().max())
        out(f"  [refine] parity vs oracle: OK -- {T} tracks within {parity.TOL_PX:g} px (max |dq| {dq:.2e} incl. flipped tracks), "
            f"argmin flips between candidates the oracle itself scores within {parity.TOL_SCORE:g}: {flips}")
    except AssertionError as e:
        ok = False
        out(f"  [refine] parity vs oracle: FAILED: {str(e)[:600]}")
    return ok


def make_seeded(out_dir, H=480, W=640):
    """Writes what a maintainer would bring -- here from seeds: three frames (the synthetic pair of BASELINE configs[1] + one more)
    and the two checkpoints in the Lightning layouts the reference's loaders read (planted LoFTR weights, seeded refinement
    weights) -- so that the script can be run end to end on a GPU box that has neither the Drive weights nor the example scene."""
    from PIL import Image
    from detectorfreesfm_amd import synth
    from detectorfreesfm_amd.params import loftr_param_spec, multiview_param_spec, planted_loftr_state_dict, random_state_dict
    os.makedirs(out_dir, exist_ok=True)
    pair = synth.coarse_pair_batch(1, H, W, seed=1000)
    third = synth.coarse_pair_batch(1, H, W, seed=1001)["image1"]
    for k, im in enumerate((pair["image0"], pair["image1"], third)):
        a = (im[0, 0].clamp(0, 1) * 255).round().byte().numpy()
        Image.fromarray(np.stack([a, a, a], -1)).save(os.path.join(out_dir, f"frame{k}.png"))
    sd = planted_loftr_state_dict(loftr_param_spec(loftr_coarse_only_config(0.2)), 0)
    lck = os.path.join(out_dir, "seeded_loftr.ckpt")
    torch.save({"state_dict": {"matcher." + k: v for k, v in sd.items()}}, lck)
    rsd = random_state_dict(multiview_param_spec(multiview_refinement_config()), 1)
    rck = os.path.join(out_dir, "seeded_multiview_matcher.ckpt")
    torch.save({"state_dict": {("matcher." + k.replace("fine_transformer", "loftr_fine")): v for k, v in rsd.items()}}, rck)
    return lck, rck


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--loftr-ckpt")
    ap.add_argument("--refine-ckpt")
    ap.add_argument("--images", help="directory of frames (the reference's example scene works)")
    ap.add_argument("--make-seeded", metavar="DIR", help="write seeded frames + checkpoints to DIR and verify THOSE (a dry run of "
                    "this script for boxes without the real weights)")
    ap.add_argument("--resize", type=int, default=640, help="longest side after the resize (the shipped configs use 1200 / 1600)")
    ap.add_argument("--thr", type=float, default=0.2)
    ap.add_argument("--views", type=int, default=2, help="views of the refinement bag (2: the pair's own coarse matches as tracks)")
    ap.add_argument("--tracks", type=int, default=96)
    ap.add_argument("--device", default="cuda:0")
    ap.add_argument("--cpu-standins", action="store_true", help="plumbing test only: CPU emulation instead of the HIP kernels")
    args = ap.parse_args(argv)
    if args.make_seeded:
        args.loftr_ckpt, args.refine_ckpt = make_seeded(args.make_seeded)
        args.images = args.make_seeded
    if not args.loftr_ckpt and not args.refine_ckpt:
        ap.error("give --loftr-ckpt and / or --refine-ckpt (or --make-seeded DIR)")
    if not args.images:
        ap.error("--images is required")
    lines = []

    def out(s):
        lines.append(s)
        print(s, flush=True)
    frames = list_frames(args.images)
    if args.cpu_standins:
        from cpu_standins import cpu_ops
        ctx, dev = cpu_ops(), torch.device("cpu")
        out("!! --cpu-standins: the HIP kernels are NOT exercised (plumbing test of this script)")
    else:
        if not torch.cuda.is_available():
            raise SystemExit("verify_checkpoint needs an MI355X (or --cpu-standins for a plumbing test)")
        ctx, dev = contextlib.nullcontext(), torch.device(args.device)
    ok, table = True, None
    with ctx:
        if args.loftr_ckpt:
            c_ok, table = verify_coarse(args.loftr_ckpt, frames, args.resize, args.thr, dev, out)
            ok = ok and c_ok
        if args.refine_ckpt:
            ok = verify_refine(args.refine_ckpt, frames, args.resize, max(2, args.views), args.tracks, table, dev, out) and ok
    out(f"== verify_checkpoint: {'ALL CLEAN' if ok else 'PROBLEMS FOUND (see above)'}")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
