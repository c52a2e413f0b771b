#!/bin/bash
# r06 session q: enc_kv with sched_group_barrier interleave (1 MFMA : n VALU) vs the product build, same box
exec < /dev/null
tag=${1:-r6q}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
for r in 1 2; do
  for v in prod kv_il8 kv_il11; do
    if [ $v = prod ]; then unset DFSFM_LIB_PATH; else export DFSFM_LIB_PATH=$PWD/detectorfreesfm_amd/csrc/abl/lib_$v.so; fi
    echo "== $v" >> $out/enc_ab.log
    timeout 300 python tools/bench_encoder_fused.py 2>&1 | grep -v amdgpu.ids | tail -6 | cut -c1-200 | grep "fused \|enc_kv" >> $out/enc_ab.log
    timeout 300 python tools/kv_image_hash.py 2>&1 | grep -v amdgpu.ids | tail -4 | cut -c1-60 | tr "\n" " " >> $out/enc_ab.log; echo >> $out/enc_ab.log
  done
done
unset DFSFM_LIB_PATH
cat $out/enc_ab.log
