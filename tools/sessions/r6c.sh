#!/bin/bash
# r06 session c: the fused S2DNet front end (tests, refine e2e, step profile, bench) + the packed op_sel reproducer.
exec < /dev/null
tag=${1:-r6c}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
R=$(pwd)
timeout 120 tools/ubench/pk_opsel_inplace 4000 > $out/pk_opsel.log 2>&1; echo "rc=$?" >> $out/pk_opsel.log; cat $out/pk_opsel.log | grep -v amdgpu.ids
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -k "s2d_front" -x > $out/pytest_front.log 2>&1; tail -15 $out/pytest_front.log
timeout 900 python -m pytest tests/test_gpu_e2e.py -q -k "refine or multiview or bag or tensor or W11 or views or flattened" > $out/pytest_e2e.log 2>&1; tail -5 $out/pytest_e2e.log
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/$out/prof_refine -- python $R/tools/profile_step.py refine > $R/$out/prof_refine.log 2>&1)
f=$(find $out/prof_refine -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/refine_step_kernel_stats.csv && head -12 $out/refine_step_kernel_stats.csv | cut -c1-150
tail -2 $out/prof_refine.log
timeout 600 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?" >> $out/bench.err
grep -o "\"value\": [0-9.]*\|\"ms_per_step\": [0-9.]*" $out/bench.json | head -6; tail -2 $out/bench.err
