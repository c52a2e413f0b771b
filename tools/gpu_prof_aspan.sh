#!/bin/bash
exec < /dev/null
out=gpurun_out/${1:-asp}; mkdir -p $out
root=$PWD; cd /tmp; export TMPDIR=/tmp
timeout 600 env PYTHONPATH=$root rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out/prof -o as -- python $root/tools/bench_aspanformer.py > $root/$out/prof.log 2>&1
cd $root
f=$(find $out/prof -name "*kernel_stats.csv" 2>/dev/null | head -1); [ -n "$f" ] && head -8 "$f" | cut -c1-150
