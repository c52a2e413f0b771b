#!/bin/bash
exec < /dev/null
out=gpurun_out/${1:-la}; mkdir -p $out
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -x -k "linear_attention or la_" > $out/pytest.log 2>&1
echo "pytest rc=$?" >> $out/pytest.log; grep -E "passed|failed|rc=" $out/pytest.log | tail -3
for w in 512; do echo "== DFSFM_LA_WGS=$w"; DFSFM_LA_WGS=$w timeout 120 python tools/bench_la.py 2>&1 | grep coarse; done | tee $out/la.txt
cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof -- python $GRAFT_REPO_ROOT/tools/bench_la.py > /dev/null 2>&1
f=$(find $GRAFT_REPO_ROOT/$out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -8 $f | cut -c1-150
rm -rf $GRAFT_REPO_ROOT/$out/prof
