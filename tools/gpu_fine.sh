#!/bin/bash
# fine_match: kernel tests (both input forms), refinement end-to-end tests, timing of both forms at 2000 x 4 views.
exec < /dev/null
tag=${1:-fm1}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -k "fine_match" > $out/pytest_k.log 2>&1; tail -3 $out/pytest_k.log
timeout 300 python tools/bench_fine.py > $out/bench_fine.log 2>&1; tail -3 $out/bench_fine.log
if [ -z "$QUICK" ]; then
timeout 900 python -m pytest tests/test_gpu_e2e.py -q -k "refine or multiview or bag or tensor or W11 or views" > $out/pytest_e.log 2>&1; tail -3 $out/pytest_e.log
fi
