#!/bin/bash
# r06 session y: diagnostics of enc_apply's slab stream (results of the variants are wrong by construction): no refill DMA, no ring barrier, neither
exec < /dev/null
tag=${1:-r6y}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
for v in prod enc_nodma enc_nobar enc_nobar_nodma; do
    if [ $v = prod ]; then unset DFSFM_LIB_PATH; else export DFSFM_LIB_PATH=$PWD/detectorfreesfm_amd/csrc/abl/lib_$v.so; fi
    echo "== $v" >> $out/enc_diag.log
    timeout 300 python tools/bench_encoder_fused.py 2>&1 | grep -v amdgpu.ids | tail -6 | cut -c1-420 | grep "query rows\|stage" >> $out/enc_diag.log
done
cat $out/enc_diag.log
