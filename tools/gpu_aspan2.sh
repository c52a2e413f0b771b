#!/bin/bash
# ASpanFormer: timing + kernel stats, then the GPU test file three more times (stress for the one-off device fault of the first run).
exec < /dev/null
out=gpurun_out/${1:-asb}; mkdir -p $out
timeout 300 python tools/bench_aspanformer.py > $out/bench.log 2>&1
root=$PWD; cd /tmp; export TMPDIR=/tmp
timeout 600 env PYTHONPATH=$root rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out/prof -o as -- python $root/tools/bench_aspanformer.py > $root/$out/prof.log 2>&1
cd $root
for r in 1 2 3; do timeout 600 python -m pytest tests/test_gpu_aspan.py -q 2>&1 | tail -3 > $out/stress$r.log; done
tail -2 $out/bench.log; cat $out/stress*.log | grep -i "passed\|failed\|fault\|abort"
f=$(find $out/prof -name "*kernel_stats.csv" 2>/dev/null | head -1); [ -n "$f" ] && head -16 "$f" | cut -c1-150
