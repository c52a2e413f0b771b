#!/bin/bash
# One gpurun call: GPU tests, smoke, bench; logs under gpurun_out/<tag>/
exec < /dev/null
tag=${1:-s1}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --maxfail=20 -s > $out/pytest.log 2>&1
echo "pytest rc=$?" >> $out/pytest.log
timeout 300 python __graft_entry__.py smoke > $out/smoke.log 2>&1
echo "smoke rc=$?" >> $out/smoke.log
timeout 600 python bench.py --steps 10 --warmup 2 > $out/bench.json 2> $out/bench.err
echo "bench rc=$?" >> $out/bench.err
grep -E "passed|failed|error" $out/pytest.log | tail -5
grep -E "^\[" $out/pytest.log | tail -40
tail -3 $out/smoke.log
grep -o "\"value\": [0-9.]*\|\"ms_per_step\": [0-9.]*\|\"secondary\": {[^}]*}" $out/bench.json | head -5; tail -2 $out/bench.err
