#!/bin/bash
out=gpurun_out/${1:-mf1}; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "la_ or dwconv or bilinear or layernorm_matchformer or leaky or padding_masks or cm_" > $out/pytest_k.log 2>&1; echo "rc=$?" >> $out/pytest_k.log
timeout 600 python -m pytest tests/test_gpu_e2e.py -q -x -s -k "matchformer" > $out/pytest_e.log 2>&1; echo "rc=$?" >> $out/pytest_e.log
grep -n "passed\|failed\|rc=\|Error\|error" $out/pytest_k.log | head -20; tail -30 $out/pytest_e.log | cut -c1-300
