#!/bin/bash
out=gpurun_out/${1:-lin1}; mkdir -p $out
DFSFM_LIN2=3 timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "conv or linear or ln or epilogue or split" > $out/pytest.log 2>&1; echo "rc=$?" >> $out/pytest.log
DFSFM_LIN2=3 timeout 600 python -m pytest tests/test_gpu_e2e.py -q -x >> $out/pytest.log 2>&1; echo "rc=$?" >> $out/pytest.log
for m in 1 3; do DFSFM_LIN2=$m timeout 300 python tools/bench_linear.py > $out/lin_$m.log 2>&1; done
grep -n "passed\|failed\|rc=\|Error" $out/pytest.log | head; paste -d'\n' $out/lin_1.log $out/lin_3.log | cut -c1-130
