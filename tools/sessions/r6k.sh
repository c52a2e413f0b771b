#!/bin/bash
# r06 session k: batched JPEG decode -- tests + bench
exec < /dev/null
tag=${1:-r6k}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_jpeg.py -q -x > $out/pytest_jpeg.log 2>&1; tail -12 $out/pytest_jpeg.log
timeout 600 python tools/bench_jpeg.py > $out/bench_jpeg.log 2>&1; tail -12 $out/bench_jpeg.log
