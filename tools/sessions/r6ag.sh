#!/bin/bash
# r06 session ag: saturating splits of the fused encoder kernels through MODE.FP16_OVFL instead of v_pk_min / v_pk_max pairs: tests, kernel timing, step A/B vs the previous build
exec < /dev/null
tag=${1:-r6ag}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_encoder_fused.py tests/test_gpu_encoder256.py -q 2>&1 | tail -3
DFSFM_LIB_PATH=$PWD/detectorfreesfm_amd/csrc/abl/lib_prev.so timeout 300 python -m pytest tests/test_gpu_encoder_fused.py -q -k saturates 2>&1 | tail -2
for r in 1 2; do
  for v in prev prod; do
    if [ $v = prod ]; then unset DFSFM_LIB_PATH; else export DFSFM_LIB_PATH=$PWD/detectorfreesfm_amd/csrc/abl/lib_$v.so; fi
    echo "== $v" >> $out/ab.log
    timeout 300 python tools/bench_encoder_fused.py 2>&1 | grep -v amdgpu.ids | tail -6 | cut -c1-200 | grep "fused \|query rows" >> $out/ab.log
    timeout 300 python tools/bench_enc256.py 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-220 >> $out/ab.log
  done
done
unset DFSFM_LIB_PATH
cat $out/ab.log
bash tools/gpu_ab.sh prev 2 2>&1 | tee $out/bench_ab.log
