#!/bin/bash
out=gpurun_out/${1:-lin1}; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "conv or linear or ln or epilogue or split" > $out/pytest.log 2>&1; echo "rc=$?" >> $out/pytest.log
for m in 0 1 2; do DFSFM_LIN2=$m timeout 300 python tools/bench_linear.py > $out/lin_$m.log 2>&1; done
tail -4 $out/pytest.log; paste -d'\n' $out/lin_0.log $out/lin_1.log $out/lin_2.log | cut -c1-130
