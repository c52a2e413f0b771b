#!/bin/bash
# r06 session j: front kernel with conv1_1 on the matrix cores -- tests, timing, refine e2e, bench
exec < /dev/null
tag=${1:-r6j}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -k "s2d_front" -x > $out/pytest_front.log 2>&1; tail -12 $out/pytest_front.log
python tools/bench_s2d_front.py 2>&1 | grep -v amdgpu.ids | tee $out/s2d_front.log
timeout 900 python -m pytest tests/test_gpu_e2e.py -q -k "refine or multiview or bag" > $out/pytest_e2e.log 2>&1; tail -3 $out/pytest_e2e.log
timeout 600 python bench.py --no-cpu-baseline > $out/bench.json 2> $out/bench.err; echo "bench rc=$?" >> $out/bench.err
grep -o "\"value\": [0-9.]*\|\"ms_per_step\": [0-9.]*" $out/bench.json | head -6; tail -2 $out/bench.err
