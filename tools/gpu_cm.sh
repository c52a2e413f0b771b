#!/bin/bash
out=gpurun_out/${1:-cm1}; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "cm_" > $out/pytest.log 2>&1; echo "rc=$?" >> $out/pytest.log
timeout 300 python -m pytest tests/test_gpu_e2e.py -q -x -k "loftr or coarse or scene" >> $out/pytest.log 2>&1; echo "rc=$?" >> $out/pytest.log
python tools/bench_cm.py > $out/cm_single.log 2>&1
DFSFM_CM_TWOPASS=1 python tools/bench_cm.py > $out/cm_two.log 2>&1
tail -6 $out/pytest.log; cat $out/cm_single.log $out/cm_two.log | grep -v amdgpu
