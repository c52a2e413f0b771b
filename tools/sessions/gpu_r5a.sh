#!/bin/bash
# r05 session a: refactor validation (K3 / conv / new e2e tests) + same-box A/B of the K3 tile traversal
exec < /dev/null
out=gpurun_out/r5a; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "coarse or cm_ or match or conv or linear" > $out/pytest_kernels.log 2>&1; tail -3 $out/pytest_kernels.log
timeout 1200 python -m pytest tests/test_gpu_e2e.py -q -x -k "benched or flattened or chunks_are_invisible or 640x480_planted" -s > $out/pytest_e2e.log 2>&1; tail -3 $out/pytest_e2e.log
timeout 300 python -m pytest tests/test_gpu_aspan.py -q -x -k "span_attention" > $out/pytest_span.log 2>&1; tail -2 $out/pytest_span.log
for r in 1 2; do
  for v in base product; do
    if [ $v = base ]; then export DFSFM_LIB_PATH=$PWD/detectorfreesfm_amd/csrc/abl/lib_base.so; else unset DFSFM_LIB_PATH; fi
    timeout 300 python tools/bench_cm.py --big --split-only 2>&1 | grep "ms / call" | tee -a $out/bench_cm.txt
  done
done
