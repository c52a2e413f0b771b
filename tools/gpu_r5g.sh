#!/bin/bash
# r05 session g: cm_eval with the column tiles dealt to four waves -- K3 tests + same-box A/B against the previous build
exec < /dev/null
out=gpurun_out/r5g; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "coarse or cm_ or match" > $out/pytest_k3.log 2>&1; tail -2 $out/pytest_k3.log
timeout 900 python -m pytest tests/test_gpu_e2e.py -q -x -k "planted or masked or golden" > $out/pytest_e2e.log 2>&1; tail -2 $out/pytest_e2e.log
for r in 1 2; do
  for v in base product; do
    if [ $v = base ]; then export DFSFM_LIB_PATH=$PWD/detectorfreesfm_amd/csrc/abl/lib_base.so; else unset DFSFM_LIB_PATH; fi
    timeout 300 python tools/bench_cm.py --big --split-only 2>&1 | grep "ms / call" | tee -a $out/bench_cm.txt
  done
done
