#!/bin/bash
# r06 session i: wave priority in the front kernel (none / in the MFMA bursts / in the conv1_1 phases), with start skew
exec < /dev/null
tag=${1:-r6i}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
for v in product PRIO1 PRIO2; do
  for k in 0 2; do
    if [ $v = product ]; then lp=""; else lp="DFSFM_LIB_PATH=$PWD/detectorfreesfm_amd/csrc/abl/lib_s2d_$v.so"; fi
    env $lp DFSFM_S2D_SKEW=$k python tools/bench_s2d_front.py 2>&1 | grep -v amdgpu.ids | sed "s/^/$v skew $k: /" >> $out/s2d_prio.log
  done
done
cat $out/s2d_prio.log
