#!/bin/bash
# r06 session f: does the front kernel overlap its two workgroups per CU at all?  One workgroup per CU (LDS pad) vs two, with ablations.
exec < /dev/null
tag=${1:-r6f}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
for pad in 0 8192; do
  for v in product NO_CONV1 NO_MFMA NO_EPI; do
    if [ $v = product ]; then lp=""; else lp="DFSFM_LIB_PATH=$PWD/detectorfreesfm_amd/csrc/abl/lib_s2d_$v.so"; fi
    env $lp DFSFM_S2D_LDS_PAD=$pad DFSFM_S2D_SKEW=0 python tools/bench_s2d_front.py 2>&1 | grep -v amdgpu.ids | sed "s/^/pad $pad $v: /" >> $out/s2d_residency.log
  done
done
cat $out/s2d_residency.log
