#!/bin/bash
# GPU session of the device JPEG decoder: parity tests, throughput beside libjpeg-turbo, kernel stats.
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
if [ "$1" != "notests" ]; then
timeout 600 python -m pytest tests/test_gpu_jpeg.py -x -q > gpurun_out/jpeg_tests.txt 2>&1; echo "tests rc=$?" >> gpurun_out/jpeg_tests.txt
tail -5 gpurun_out/jpeg_tests.txt
fi
timeout 300 python tools/bench_jpeg.py --reps 20 > gpurun_out/jpeg_bench.txt 2>&1; tail -30 gpurun_out/jpeg_bench.txt
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/jprof -o jpeg -- python $OLDPWD/tools/profile_jpeg.py 20 rgb > $OLDPWD/gpurun_out/jpeg_prof.log 2>&1 )
cp /tmp/jprof/*kernel_stats.csv gpurun_out/jpeg_kernel_stats.csv 2>/dev/null || find /tmp/jprof -name "*kernel_stats*" -exec cp {} gpurun_out/jpeg_kernel_stats.csv \;
head -16 gpurun_out/jpeg_kernel_stats.csv
