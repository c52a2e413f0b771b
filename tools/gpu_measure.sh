#!/bin/bash
# Measurement session: explicit workloads, RCCL single-rank path, rocprofv3 kernel stats of both steps, PMC traffic.
tag=${1:-m1}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
timeout 400 python bench.py --workload scene300 > $out/scene300.json 2> $out/scene300.err; echo "scene rc=$?"
timeout 400 python bench.py --workload hires832 --steps 6 --warmup 2 > $out/hires832.json 2> $out/hires832.err; echo "hires rc=$?"
DFSFM_BENCH_FORCE_DIST=1 timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-rooflines > $out/dist1.json 2> $out/dist1.err; echo "dist rc=$?"
for w in coarse refine; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$w -- python tools/profile_step.py $w 4 > $out/prof_$w.log 2>&1
  f=$(find $out/prof_$w -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/${w}_step_kernel_stats.csv
  rm -rf $out/prof_$w
done
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_k -- python bench.py --kernels-only > $out/prof_k.log 2>&1
f=$(find $out/prof_k -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/kernels_only_kernel_stats.csv; rm -rf $out/prof_k
timeout 600 python tools/pmc_collect.py $out/pmc_traffic.json --scratch $out/pmc_s > $out/pmc.log 2>&1; rm -rf $out/pmc_s
ls -la $out
