#!/bin/bash
# ASpanFormer session: tests/test_gpu_aspan.py in one process, then (a device fault must not hide the rest) test by test.
exec < /dev/null
out=gpurun_out/${1:-as1}; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_aspan.py -q -s 2>&1 | grep -v "^  File\|pluggy\|_pytest" | tail -40 > $out/pytest_all.log
: > $out/pytest.log
for t in avgpool full_attention span_attention layernorm2d upsample flow_decode e2e_golden 480x640 plugin_surface; do
  echo "=== $t" >> $out/pytest.log
  timeout 600 python -m pytest tests/test_gpu_aspan.py -q -s -k $t 2>&1 | grep -v "^  File\|pluggy\|_pytest" | tail -25 >> $out/pytest.log
done
grep -n "passed\|failed\|^\[aspan\|Error\|assert \|fault" $out/pytest_all.log | cut -c1-420 | head -20
grep -n "===\|passed\|failed\|^\[aspan\|Error\|assert \|fault" $out/pytest.log | cut -c1-420 | head -60
