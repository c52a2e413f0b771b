#!/bin/bash
# r06 session e: op_sel hazard map incl. long runs of the clean forms, the SLP build with only the op_sel forms rewritten,
# the front kernel with packed conv1_1 + first-generation skew (tests, skew sweep, bench).
exec < /dev/null
tag=${1:-r6e}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp FINE_BISECT_DIR=/tmp/fine_bisect
timeout 600 tools/ubench/pk_opsel_inplace 4000 > $out/pk_opsel.log 2>&1; echo "rc=$?" >> $out/pk_opsel.log; grep -v amdgpu.ids $out/pk_opsel.log | grep -v "     got"
timeout 600 python tools/studies/fine_bisect.py opsel > $out/fine_opsel.log 2>&1; grep -v amdgpu.ids $out/fine_opsel.log
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -k "s2d_front" -x > $out/pytest_front.log 2>&1; tail -3 $out/pytest_front.log
for k in 0 1 2 3 4 6; do DFSFM_S2D_SKEW=$k python tools/bench_s2d_front.py 2>&1 | grep -v amdgpu.ids | sed "s/^/skew $k: /" >> $out/s2d_skew.log; done; cat $out/s2d_skew.log
timeout 900 python -m pytest tests/test_gpu_e2e.py -q -k "refine or multiview or bag" > $out/pytest_e2e.log 2>&1; tail -3 $out/pytest_e2e.log
timeout 600 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?" >> $out/bench.err
grep -o "\"value\": [0-9.]*\|\"ms_per_step\": [0-9.]*" $out/bench.json | head -6; tail -2 $out/bench.err
