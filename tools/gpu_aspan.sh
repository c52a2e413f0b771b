#!/bin/bash
# ASpanFormer session: tests/test_gpu_aspan.py.
exec < /dev/null
out=gpurun_out/${1:-as1}; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_aspan.py -q -s 2>&1 | grep -v "^  File\|pluggy\|_pytest" | tail -40 > $out/pytest_all.log
grep -n "passed\|failed\|^\[aspan\|Error\|assert \|fault" $out/pytest_all.log | cut -c1-420 | head -20
