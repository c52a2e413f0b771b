#!/bin/bash
# r06 session aa: the fused encoder files built with -mllvm -amdgpu-mfma-vgpr-form (MFMA results in the vector file: fewer v_accvgpr moves) vs the product build
exec < /dev/null
tag=${1:-r6aa}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
for v in vformboth; do
  DFSFM_LIB_PATH=$PWD/detectorfreesfm_amd/csrc/abl/lib_$v.so timeout 900 python -m pytest tests/test_gpu_encoder_fused.py tests/test_gpu_encoder256.py -q 2>&1 | tail -3
done
for r in 1 2; do
  for v in prod vform128 vformboth; do
    if [ $v = prod ]; then unset DFSFM_LIB_PATH; else export DFSFM_LIB_PATH=$PWD/detectorfreesfm_amd/csrc/abl/lib_$v.so; fi
    echo "== $v" >> $out/enc_ab.log
    timeout 300 python tools/bench_encoder_fused.py 2>&1 | grep -v amdgpu.ids | tail -6 | cut -c1-200 | grep "fused \|query rows" >> $out/enc_ab.log
    timeout 300 python tools/bench_enc256.py 2>&1 | grep -v amdgpu.ids | tail -8 | cut -c1-220 >> $out/enc_ab.log
  done
done
unset DFSFM_LIB_PATH
cat $out/enc_ab.log
bash tools/gpu_ab.sh vformboth 2 2>&1 | tee $out/bench_ab.log
