#!/bin/bash
# r06 session u: enc_kv variants, same box: head (builtin MFMAs, fragment reads 2 ahead), head3 (3 ahead), prod (asm MFMAs, 3 ahead)
exec < /dev/null
tag=${1:-r6u}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_encoder_fused.py -q 2>&1 | tail -3
DFSFM_LIB_PATH=$PWD/detectorfreesfm_amd/csrc/abl/lib_kv_head3.so timeout 900 python -m pytest tests/test_gpu_encoder_fused.py -q 2>&1 | tail -3
for r in 1 2; do
  for v in kv_head kv_head3 prod; do
    if [ $v = prod ]; then unset DFSFM_LIB_PATH; else export DFSFM_LIB_PATH=$PWD/detectorfreesfm_amd/csrc/abl/lib_$v.so; fi
    echo "== $v" >> $out/enc_ab.log
    timeout 300 python tools/bench_encoder_fused.py 2>&1 | grep -v amdgpu.ids | tail -6 | cut -c1-200 | grep "fused \|enc_kv" >> $out/enc_ab.log
  done
done
unset DFSFM_LIB_PATH
cat $out/enc_ab.log
