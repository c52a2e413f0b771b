#!/bin/bash
# rocprofv3 kernel stats of the default bench command on the final build (the profile of "the same command" the bench line comes from).
exec < /dev/null
cd "$(dirname "$0")/.." || exit 1
root=$PWD; mkdir -p gpurun_out/final; export TMPDIR=/tmp
cd /tmp
timeout 200 env PYTHONPATH=$root rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fprof -o b -- python $root/bench.py --no-cpu-baseline > $root/gpurun_out/final/bench_under_rocprof.json 2> $root/gpurun_out/final/bench_under_rocprof.err
f=$(find /tmp/fprof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $root/gpurun_out/final/bench_kernel_stats.csv
head -6 $root/gpurun_out/final/bench_kernel_stats.csv | cut -c1-150
tail -c 600 $root/gpurun_out/final/bench_under_rocprof.json
