#!/bin/bash
# r06 session g: front kernel after the latency fixes (tests, ablations at one / two workgroups per CU), then the whole GPU suite + bench.
exec < /dev/null
tag=${1:-r6g}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -k "s2d_front" -x > $out/pytest_front.log 2>&1; tail -3 $out/pytest_front.log
for pad in 0 8192; do
  for v in product NO_CONV1 NO_MFMA NO_EPI; do
    if [ $v = product ]; then lp=""; else lp="DFSFM_LIB_PATH=$PWD/detectorfreesfm_amd/csrc/abl/lib_s2d_$v.so"; fi
    env $lp DFSFM_S2D_LDS_PAD=$pad python tools/bench_s2d_front.py 2>&1 | grep -v amdgpu.ids | sed "s/^/pad $pad $v: /" >> $out/s2d_residency.log
  done
done
cat $out/s2d_residency.log
for k in 0 2 4; do DFSFM_S2D_SKEW=$k python tools/bench_s2d_front.py 2>&1 | grep -v amdgpu.ids | sed "s/^/skew $k: /" >> $out/s2d_skew.log; done; cat $out/s2d_skew.log
timeout 1500 python -m pytest tests -m gpu -q --maxfail=20 > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
grep -E "passed|failed|error" $out/pytest.log | tail -5; grep -E "^FAILED|^ERROR" $out/pytest.log | head
timeout 300 python __graft_entry__.py smoke > $out/smoke.log 2>&1; echo "smoke rc=$?" >> $out/smoke.log; tail -2 $out/smoke.log
timeout 600 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?" >> $out/bench.err
grep -o "\"value\": [0-9.]*\|\"ms_per_step\": [0-9.]*" $out/bench.json | head -6; tail -2 $out/bench.err
