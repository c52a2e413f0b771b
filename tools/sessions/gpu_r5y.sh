#!/bin/bash
# r05 last check of the final tree: full GPU suite + smoke
exec < /dev/null
out=gpurun_out/r5y; mkdir -p $out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q --durations=12 > $out/pytest_gpu.log 2>&1; tail -18 $out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > $out/smoke.log 2>&1; tail -2 $out/smoke.log
