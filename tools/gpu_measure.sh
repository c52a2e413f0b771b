#!/bin/bash
# Measurement session of the round: explicit workloads, RCCL single-rank path, rocprofv3 kernel stats of both steps and of
# bench.py --kernels-only, PMC traffic (separate FETCH_SIZE / WRITE_SIZE passes, --kernel-trace only).
exec < /dev/null
tag=${1:-m1}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
root=$PWD
quick=${2:-}          # "quick": skip the workloads whose kernels did not change since the last full session
timeout 400 python bench.py --workload scene300 --no-cpu-baseline > $out/scene300.json 2> $out/scene300.err; echo "scene rc=$?"
if [ -z "$quick" ]; then
timeout 300 python bench.py --workload scene300 --scene-images 60 > $out/scene60.json 2> $out/scene60.err; echo "scene60 rc=$?"
DFSFM_BENCH_FORCE_DIST=1 timeout 300 python bench.py --workload scene300 --scene-images 60 > $out/scene60_dist.json 2> $out/scene60_dist.err; echo "scene60 dist rc=$?"
fi
timeout 400 python bench.py --workload hires832 --steps 6 --warmup 2 --no-cpu-baseline > $out/hires832.json 2> $out/hires832.err; echo "hires rc=$?"
timeout 300 python bench.py --workload eth3d1600 --steps 5 --warmup 2 --no-cpu-baseline > $out/eth3d1600.json 2> $out/eth3d1600.err; echo "eth3d1600 rc=$?"
timeout 300 python bench.py --workload demo1200 --steps 5 --warmup 2 --no-cpu-baseline > $out/demo1200.json 2> $out/demo1200.err; echo "demo1200 rc=$?"
timeout 300 python bench.py --workload matchformer --steps 6 --warmup 2 --no-cpu-baseline > $out/matchformer.json 2> $out/mf.err; echo "mf rc=$?"
timeout 300 python bench.py --workload aspanformer --steps 6 --warmup 2 --no-cpu-baseline > $out/aspanformer.json 2> $out/as.err; echo "as rc=$?"
if [ -z "$quick" ]; then
timeout 300 python bench.py --workload aspanformer --alt-frame 832x832 --batch 4 --steps 4 --warmup 1 > $out/aspanformer832.json 2> $out/as832.err; echo "as832 rc=$?"
timeout 300 python bench.py --workload scene300 --scene-matcher aspanformer --scene-images 40 > $out/scene40_aspanformer.json 2> $out/scene40_as.err; echo "scene40 aspan rc=$?"
fi
cd /tmp
for w in coarse refine; do
  timeout 600 env PYTHONPATH=$root rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out/prof_$w -o $w -- python $root/tools/profile_step.py $w 10 > $root/$out/prof_$w.log 2>&1
  f=$(find $root/$out/prof_$w -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $root/$out/${w}_step_kernel_stats.csv
  rm -rf $root/$out/prof_$w
done
cd $root
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out/prof_k -o k -- python bench.py --kernels-only > $out/kernels_only.json 2> $out/prof_k.log
f=$(find $out/prof_k -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/kernels_only_kernel_stats.csv
rm -rf $out/prof_k
timeout 900 python tools/pmc_collect.py $out/pmc_traffic.json --scratch $out/pmc_s > $out/pmc.log 2>&1; rm -rf $out/pmc_s
ls $out
