#!/bin/bash
exec < /dev/null
out=gpurun_out/${1:-mf1}; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -k "la_ or dwconv or bilinear or layernorm_matchformer or leaky or padding_masks or cm_" > $out/pytest_k.log 2>&1; echo "rc=$?" >> $out/pytest_k.log
timeout 300 python tools/bench_matchformer.py 8 > $out/bench_mf.log 2>&1
cd /tmp; export TMPDIR=/tmp
timeout 600 env PYTHONPATH=$GRAFT_REPO_ROOT rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$out/prof -o mf -- python $GRAFT_REPO_ROOT/tools/bench_matchformer.py 8 > $GRAFT_REPO_ROOT/$out/prof.log 2>&1
cd $GRAFT_REPO_ROOT
grep -n "passed\|failed\|rc=\|Error\|error" $out/pytest_k.log | head -20; cat $out/bench_mf.log | tail -3
f=$(find $out/prof -name "*kernel_stats.csv" 2>/dev/null | head -1); [ -n "$f" ] && head -25 "$f" | cut -c1-160
