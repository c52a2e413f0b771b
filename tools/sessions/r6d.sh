#!/bin/bash
# r06 session d: op_sel hazard map, timing-only ablations of the fused front kernel, the new JPEG tests on the device.
exec < /dev/null
tag=${1:-r6d}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
timeout 200 tools/ubench/pk_opsel_inplace 4000 > $out/pk_opsel.log 2>&1; echo "rc=$?" >> $out/pk_opsel.log; grep -v amdgpu.ids $out/pk_opsel.log
python tools/bench_s2d_front.py > $out/s2d_front_abl.log 2>&1
for v in NO_CONV1 NO_MFMA NO_EPI NO_CONV1_NO_EPI NO_CONV1_NO_EPI_NO_MFMA; do
  DFSFM_LIB_PATH=$PWD/detectorfreesfm_amd/csrc/abl/lib_s2d_$v.so python tools/bench_s2d_front.py >> $out/s2d_front_abl.log 2>&1
done
grep -v amdgpu.ids $out/s2d_front_abl.log
timeout 900 python -m pytest tests/test_gpu_jpeg.py -q -x > $out/pytest_jpeg.log 2>&1; tail -5 $out/pytest_jpeg.log
timeout 300 python tools/bench_jpeg.py > $out/bench_jpeg.log 2>&1; tail -25 $out/bench_jpeg.log
