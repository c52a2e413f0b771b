#!/bin/bash
# r05 session i: the two oracle checks added last (configs[3] scene path at 640x480, configs[4] 16000-track chunk sample)
exec < /dev/null
out=gpurun_out/r5i; mkdir -p $out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_e2e.py -q -x -s -k "scene_path_640x480 or chunk16000" > $out/pytest.log 2>&1; grep "scene pair\|chunk 16000\|passed\|failed\|Error" $out/pytest.log | cut -c1-220
