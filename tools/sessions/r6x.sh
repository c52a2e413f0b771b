#!/bin/bash
# r06 session x: enc_apply with hand-placed look-ahead fragment reads across the ring barrier vs the previous build (abl/lib_enc_prev.so), same box
exec < /dev/null
tag=${1:-r6x}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_encoder_fused.py -q 2>&1 | tail -4
for r in 1 2; do
  for v in enc_prev prod; do
    if [ $v = prod ]; then unset DFSFM_LIB_PATH; else export DFSFM_LIB_PATH=$PWD/detectorfreesfm_amd/csrc/abl/lib_$v.so; fi
    echo "== $v" >> $out/enc_ab.log
    timeout 300 python tools/bench_encoder_fused.py 2>&1 | grep -v amdgpu.ids | tail -6 | cut -c1-420 | grep "fused \|enc_kv\|stage" >> $out/enc_ab.log
  done
done
unset DFSFM_LIB_PATH
cat $out/enc_ab.log
