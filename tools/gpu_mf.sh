#!/bin/bash
# MatchFormer-LA session: its kernel tests, the e2e parity tests, tools/bench_matchformer.py and a kernel-stats profile of it.
exec < /dev/null
out=gpurun_out/${1:-mf1}; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -k "la_ or dwconv or bilinear or layernorm or leaky or padding_masks or cm_" > $out/pytest_k.log 2>&1; echo "rc=$?" >> $out/pytest_k.log
timeout 600 python -m pytest tests/test_gpu_e2e.py -q -x -s -k "matchformer or refine_scene_worker" > $out/pytest_e.log 2>&1; echo "rc=$?" >> $out/pytest_e.log
timeout 300 python tools/bench_matchformer.py 8 > $out/bench_mf.log 2>&1
root=$PWD; cd /tmp; export TMPDIR=/tmp
timeout 600 env PYTHONPATH=$root rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out/prof -o mf -- python $root/tools/bench_matchformer.py 8 > $root/$out/prof.log 2>&1
cd $root
grep -n "passed\|failed\|rc=\|Error\|error" $out/pytest_k.log | head -20; tail -8 $out/pytest_e.log | cut -c1-300; tail -3 $out/bench_mf.log
f=$(find $out/prof -name "*kernel_stats.csv" 2>/dev/null | head -1); [ -n "$f" ] && head -25 "$f" | cut -c1-160
