#!/bin/bash
# fused encoder layer: kernel tests (bounded: a barrier mismatch would hang) + timing (+ stagger sweep)
exec < /dev/null
out=gpurun_out/${1:-enc1}
mkdir -p $out
timeout 300 python -m pytest tests/test_gpu_encoder_fused.py -q -s --maxfail=30 > $out/pytest.log 2>&1
echo "pytest rc=$?" >> $out/pytest.log
grep -E "passed|failed|Error|error|rc=" $out/pytest.log | head -10
for st in ${STAGGERS:-2}; do
  echo "== DFSFM_ENC_STAGGER=$st"
  DFSFM_ENC_STAGGER=$st timeout 120 python tools/bench_encoder_fused.py 2>&1 | grep -v amdgpu.ids | tee -a $out/bench_enc.txt
done
