#!/bin/bash
# r06 session ab: scheduler-strategy builds of the two fused-encoder files vs the product build, same box
exec < /dev/null
tag=${1:-r6ab}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
for r in 1 2; do
  for v in prod ef_ilp ef_mem; do
    if [ $v = prod ]; then unset DFSFM_LIB_PATH; else export DFSFM_LIB_PATH=$PWD/detectorfreesfm_amd/csrc/abl/lib_$v.so; fi
    echo "== $v" >> $out/ab.log
    timeout 300 python tools/bench_encoder_fused.py 2>&1 | grep -v amdgpu.ids | tail -6 | cut -c1-200 | grep "fused \|query rows" >> $out/ab.log
  done
  for v in prod e256_ilp e256_vf e256_trk; do
    if [ $v = prod ]; then unset DFSFM_LIB_PATH; else export DFSFM_LIB_PATH=$PWD/detectorfreesfm_amd/csrc/abl/lib_$v.so; fi
    echo "== $v" >> $out/ab.log
    timeout 300 python tools/bench_enc256.py 2>&1 | grep -v amdgpu.ids | tail -12 | cut -c1-400 >> $out/ab.log
  done
done
unset DFSFM_LIB_PATH
cat $out/ab.log
