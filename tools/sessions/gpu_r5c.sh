#!/bin/bash
# r05 session c: warm step profiles (no one-time work) of both halves
exec < /dev/null
out=gpurun_out/r5c; mkdir -p $out
export TMPDIR=/tmp
root=$PWD
cd /tmp
for w in coarse refine; do
  timeout 600 env PYTHONPATH=$root rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out/prof_$w -o $w -- python $root/tools/profile_step.py $w 10 > $root/$out/prof_$w.log 2>&1
  f=$(find $root/$out/prof_$w -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $root/$out/${w}_step_kernel_stats.csv
  rm -rf $root/$out/prof_$w
  tail -1 $root/$out/prof_$w.log
done
