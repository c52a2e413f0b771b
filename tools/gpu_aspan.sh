#!/bin/bash
# ASpanFormer session: kernel + e2e GPU tests, then a timing of the 640x480 pair.
exec < /dev/null
out=gpurun_out/${1:-as1}; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_aspan.py -q -s > $out/pytest.log 2>&1; echo "rc=$?" >> $out/pytest.log
grep -n "passed\|failed\|rc=\|^\[aspan\|Error\|assert " $out/pytest.log | head -40
