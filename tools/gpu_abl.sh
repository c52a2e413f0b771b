#!/bin/bash
out=gpurun_out/${1:-abl1}; mkdir -p $out; rm -f $out/abl.log
for v in "" ROW128 NOEPI; do
  if [ -z "$v" ]; then unset DFSFM_LIB_PATH; else export DFSFM_LIB_PATH=$PWD/detectorfreesfm_amd/csrc/abl/lib_$v.so; fi
  echo "== variant ${v:-full}" >> $out/abl.log
  timeout 200 python tools/bench_linear.py 2>&1 | grep -v amdgpu.ids >> $out/abl.log
done
cat $out/abl.log
