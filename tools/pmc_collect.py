"""HBM traffic of the hand-written kernels from rocprofv3 PMC passes (run on the GPU box, through gpurun).

usage: python tools/pmc_collect.py <out.json> [--target "<python command>"] [--extra COUNTER,COUNTER ...]

Runs the target (default: ``python bench.py --kernels-only``) once per counter group -- FETCH_SIZE and WRITE_SIZE
cannot share a pass on gfx950 (TCC has 4 slots; FETCH_SIZE takes 3, WRITE_SIZE 2; MI355X_MICROARCH.md, "rocprofv3
PMC slots") -- with ``--pmc <group> --kernel-trace`` only (no sys/hip/hsa trace domains), then averages the counters
per launch for every kernel of libdfsfm_hip.so and writes one JSON:

  fetch_MB_raw  = FETCH_SIZE (KB) / 1024
  fetch_MB_x2   = 2 x that: on gfx950 FETCH_SIZE tallies the 128-B requests of 16-B/lane streaming reads at 64 B
                  (guide, HBM section) -- every kernel here streams with buffer_load_dwordx4 / LDS-DMA
  write_MB      = WRITE_SIZE (KB) / 1024 (matches the algorithmic output bytes where those are known)
"""
import argparse
import csv
import glob
import json
import os
import shlex
import subprocess
import sys
import time
from collections import defaultdict

OURS = ("conv_gemm", "cm_", "la_", "roi_align", "fine_match", "layernorm", "split_rows", "direct_conv", "maxpool",
        "add_scatter", "mk_", "mlp_", "bag_", "dfsfm", "enc_", "enc256", "jd_", "s2d_front")


def source_sha256():
    """Identity of the library build the passes ran on (bench.py compares it with the sources it times: ``traffic_build_matches``)."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from detectorfreesfm_amd import _lib
    return _lib.source_sha256()


def ours(name: str) -> bool:
    return "at::native" not in name and "rocprim" not in name and any(k in name for k in OURS)


def run_pass(target, counters, outdir):
    os.makedirs(outdir, exist_ok=True)
    cmd = ["rocprofv3", "--pmc", *counters, "--kernel-trace", "--output-format", "csv", "-d", outdir, "--"] + shlex.split(target)
    env = dict(os.environ, TMPDIR="/tmp")
    if subprocess.call(cmd, env=env, stdout=subprocess.DEVNULL) != 0:       # e.g. a counter this part does not have: skip the pass
        print(f"pass {counters} failed", file=sys.stderr)
        return defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    csv.field_size_limit(1 << 30)
    for path in glob.glob(os.path.join(outdir, "**", "*_counter_collection.csv"), recursive=True):
        with open(path, newline="") as fh:
            for row in csv.DictReader(fh):
                name = row["Kernel_Name"]
                if ours(name):
                    a = acc[name][row["Counter_Name"]]
                    a[0] += float(row["Counter_Value"])
                    a[1] += 1
    return acc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--target", default=f"{sys.executable} bench.py --kernels-only")
    ap.add_argument("--extra", nargs="*", default=[], help="additional counter groups, comma separated per pass")
    ap.add_argument("--scratch", default="gpurun_out/pmc_scratch")
    args = ap.parse_args()
    groups = [["FETCH_SIZE"], ["WRITE_SIZE"]] + [g.split(",") for g in args.extra]
    merged = defaultdict(dict)
    for gi, g in enumerate(groups):
        acc = run_pass(args.target, g, os.path.join(args.scratch, f"pass{gi}"))
        for name, ctrs in acc.items():
            for c, (tot, n) in ctrs.items():
                merged[name][c] = tot / max(n, 1)
                merged[name]["launches"] = n
    rows = []
    for name in sorted(merged):
        m = merged[name]
        row = {"kernel": name, "launches": m.get("launches", 0)}
        if "FETCH_SIZE" in m:
            row["fetch_MB_raw"] = round(m["FETCH_SIZE"] / 1024.0, 3)
            row["fetch_MB_x2"] = round(2.0 * m["FETCH_SIZE"] / 1024.0, 3)
        if "WRITE_SIZE" in m:
            row["write_MB"] = round(m["WRITE_SIZE"] / 1024.0, 3)
        for c, v in m.items():
            if c not in ("FETCH_SIZE", "WRITE_SIZE", "launches"):
                row[c] = v
        rows.append(row)
    with open(args.out, "w") as fh:
        json.dump({"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, --kernel-trace only) of `" + args.target +
                           "`, averages per launch; FETCH_SIZE doubled per the MI355X guide (16 B/lane streaming reads)",
                   "library_source_sha256": source_sha256(), "collected": time.strftime("%Y-%m-%d %H:%M:%S"),
                   "kernels": rows}, fh, indent=1)
    print(f"{args.out}: {len(rows)} kernels")


if __name__ == "__main__":
    main()
