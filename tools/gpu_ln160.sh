#!/bin/bash
exec < /dev/null
tag=${1:-ln1}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -k "linear_ln" > $out/pytest.log 2>&1; tail -4 $out/pytest.log
timeout 300 python tools/bench_ln160.py 2>&1 | grep -v amdgpu.ids | tee $out/bench_ln160.log
