#!/bin/bash
# r05 session d: the tests added since session b + the in-loop stage timeline
exec < /dev/null
out=gpurun_out/r5d; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_encoder256.py tests/test_gpu_encoder_fused.py -q -x -k "bit_reproducible" > $out/pytest_repro.log 2>&1; tail -3 $out/pytest_repro.log
timeout 900 python -m pytest tests/test_gpu_e2e.py -q -x -s -k "batch2_1600" > $out/pytest_1600.log 2>&1; tail -4 $out/pytest_1600.log
timeout 300 python tools/step_timeline.py > $out/step_timeline.txt 2>&1; tail -6 $out/step_timeline.txt
