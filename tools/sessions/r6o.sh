#!/bin/bash
# r06 session o: enc_kv with resident weights vs the ring version (same box): image hashes, encoder tests, kernel timing, step A/B; S2DNet per-layer times
exec < /dev/null
tag=${1:-r6o}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
for v in old new; do
  if [ $v = old ]; then export DFSFM_LIB_PATH=$PWD/detectorfreesfm_amd/csrc/abl/lib_enc_old.so; else unset DFSFM_LIB_PATH; fi
  echo "== $v" >> $out/kv_hash.log
  timeout 300 python tools/kv_image_hash.py 2>&1 | grep -v amdgpu.ids | tail -6 >> $out/kv_hash.log
done
cat $out/kv_hash.log
for r in 1 2; do
  for v in old new; do
    if [ $v = old ]; then export DFSFM_LIB_PATH=$PWD/detectorfreesfm_amd/csrc/abl/lib_enc_old.so; else unset DFSFM_LIB_PATH; fi
    echo "== $v" >> $out/enc_ab.log
    timeout 300 python tools/bench_encoder_fused.py 2>&1 | grep -v amdgpu.ids | tail -6 | cut -c1-400 >> $out/enc_ab.log
  done
done
unset DFSFM_LIB_PATH
cat $out/enc_ab.log
timeout 900 python -m pytest tests/test_gpu_encoder_fused.py -q > $out/pytest_enc.log 2>&1; tail -3 $out/pytest_enc.log
timeout 300 python tools/bench_s2d_layers.py > $out/s2d_layers.txt 2>&1; cat $out/s2d_layers.txt | grep -v amdgpu.ids
bash tools/gpu_ab.sh enc_old 2 2>&1 | tee $out/bench_ab.log
