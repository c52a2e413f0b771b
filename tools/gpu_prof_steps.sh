#!/bin/bash
# rocprofv3 kernel stats of one warm coarse and one warm refinement step (tools/profile_step.py).
exec < /dev/null
tag=${1:-p1}; out=gpurun_out/$tag; mkdir -p $out
root=$PWD; cd /tmp; export TMPDIR=/tmp
for w in coarse refine; do
  timeout 600 env PYTHONPATH=$root rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out/prof_$w -o $w -- python $root/tools/profile_step.py $w 4 > $root/$out/prof_$w.log 2>&1
  f=$(find $root/$out/prof_$w -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $root/$out/${w}_step_kernel_stats.csv
  rm -rf $root/$out/prof_$w
done
cd $root; head -28 $out/refine_step_kernel_stats.csv | cut -c1-130
