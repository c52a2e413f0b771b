#!/bin/bash
# r06 session n: encoder_fused slab_mma with in-k-step fragment prefetch vs the previous build (same box): kernel timing, tests
exec < /dev/null
tag=${1:-r6n}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
for r in 1 2; do
  for v in old new; do
    if [ $v = old ]; then export DFSFM_LIB_PATH=$PWD/detectorfreesfm_amd/csrc/abl/lib_enc_old.so; else unset DFSFM_LIB_PATH; fi
    echo "== $v" >> $out/enc_ab.log
    python tools/bench_encoder_fused.py 2>&1 | grep -v amdgpu.ids | tail -6 >> $out/enc_ab.log
  done
done
unset DFSFM_LIB_PATH
cat $out/enc_ab.log
timeout 900 python -m pytest tests/test_gpu_encoder_fused.py -q > $out/pytest_enc.log 2>&1; tail -3 $out/pytest_enc.log
bash tools/gpu_ab.sh enc_old 2 2>&1 | tee $out/bench_ab.log
