#!/bin/bash
exec < /dev/null
out=gpurun_out/${1:-k2}; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "coarse_match or conv2d or linear or cm_" > $out/pytest.log 2>&1
echo "pytest rc=$?" >> $out/pytest.log; grep -E "passed|failed|rc=" $out/pytest.log | tail -3
timeout 300 python tools/bench_conv_layers.py same 2>&1 | grep -v amdgpu | grep "conv1_2\|adap\|l1 3x3" | tee $out/conv.txt
