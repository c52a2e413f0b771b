#!/bin/bash
# r06 session p: enc_kv -- flat 64-step stream with hand-placed fragment reads, 1 / S, ballot mask -- vs the ring version (same box)
exec < /dev/null
tag=${1:-r6p}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_encoder_fused.py -q -x > $out/pytest_enc.log 2>&1; tail -5 $out/pytest_enc.log
for r in 1 2; do
  for v in old new; do
    if [ $v = old ]; then export DFSFM_LIB_PATH=$PWD/detectorfreesfm_amd/csrc/abl/lib_enc_old.so; else unset DFSFM_LIB_PATH; fi
    echo "== $v" >> $out/enc_ab.log
    timeout 300 python tools/bench_encoder_fused.py 2>&1 | grep -v amdgpu.ids | tail -6 | cut -c1-200 | grep -v "stage medians" >> $out/enc_ab.log
  done
done
unset DFSFM_LIB_PATH
cat $out/enc_ab.log
timeout 300 python tools/kv_image_hash.py 2>&1 | grep -v amdgpu.ids | tail -4
timeout 1200 python -m pytest tests/test_gpu_e2e.py -q -x -k "multiview or refine" > $out/pytest_e2e.log 2>&1; tail -5 $out/pytest_e2e.log
bash tools/gpu_ab.sh enc_old 2 2>&1 | tee $out/bench_ab.log
