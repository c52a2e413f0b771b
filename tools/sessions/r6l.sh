#!/bin/bash
exec < /dev/null
tag=${1:-r6l}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
R=$PWD
python tools/profile_jpeg_batch.py 2>&1 | grep -v amdgpu.ids
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof -o jb -- python $R/tools/profile_jpeg_batch.py > $R/$out/prof.log 2>&1)
f=$(find $out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/jpeg_batch_kernel_stats.csv && head -20 $out/jpeg_batch_kernel_stats.csv | cut -c1-160
rm -rf $out/prof
