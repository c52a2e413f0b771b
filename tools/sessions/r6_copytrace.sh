#!/bin/bash
exec < /dev/null
export TMPDIR=/tmp
root=$PWD; out=$root/gpurun_out/copytrace; mkdir -p $out
cd /tmp
timeout 600 env PYTHONPATH=$root rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $out/t -o c -- python $root/tools/profile_step.py coarse 4 > $out/log.txt 2>&1
find $out/t -name "*.csv" | head
f=$(find $out/t -name "*memory_copy_trace.csv" | head -1); [ -n "$f" ] && cp $f $out/memcpy.csv
f=$(find $out/t -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && cp $f $out/kernels.csv
rm -rf $out/t
ls -la $out
