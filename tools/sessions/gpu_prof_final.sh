#!/bin/bash
# rocprofv3 kernel stats of the final build: one warm coarse / refinement step (tools/profile_step.py) and bench.py --kernels-only.
exec < /dev/null
tag=${1:-pf}; out=gpurun_out/$tag; mkdir -p $out
root=$PWD; cd /tmp; export TMPDIR=/tmp
for w in coarse refine; do
  timeout 600 env PYTHONPATH=$root rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out/prof_$w -o $w -- python $root/tools/profile_step.py $w 4 > $root/$out/prof_$w.log 2>&1
  f=$(find $root/$out/prof_$w -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $root/$out/${w}_step_kernel_stats.csv
  rm -r $root/$out/prof_$w
done
cd $root
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out/prof_k -o k -- python bench.py --kernels-only > $out/kernels_only.json 2> $out/prof_k.log
f=$(find $out/prof_k -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/kernels_only_kernel_stats.csv
rm -r $out/prof_k
head -4 $out/kernels_only_kernel_stats.csv | cut -c1-140; head -3 $out/coarse_step_kernel_stats.csv | cut -c1-140
