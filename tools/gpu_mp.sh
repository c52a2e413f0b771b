#!/bin/bash
exec < /dev/null
out=gpurun_out/${1:-mp1}; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -k "maxpool or split_rows" 2>&1 | tail -5 > $out/pytest_k.log
timeout 900 python -m pytest tests/test_gpu_e2e.py -q -k "multiview or refine" 2>&1 | tail -5 > $out/pytest_e.log
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $out/bench.json 2> $out/bench.err
cat $out/pytest_k.log $out/pytest_e.log | grep "passed\|failed"
grep -o "\"value\": [0-9.]*\|\"ms_per_step\": [0-9.]*" $out/bench.json | head -4
