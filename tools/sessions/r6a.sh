#!/bin/bash
# r06 session a: baseline of the round on this box (full GPU suite, smoke, bench) + the assembly-level fine_match bisect.
exec < /dev/null
tag=${1:-r6a}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
export FINE_BISECT_DIR=/tmp/fine_bisect
timeout 900 python tools/studies/fine_bisect.py auto > $out/fine_bisect.log 2>&1; echo "bisect rc=$?" >> $out/fine_bisect.log
tail -40 $out/fine_bisect.log
timeout 900 python -m pytest tests -m gpu -q --maxfail=20 > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
grep -E "passed|failed|error" $out/pytest.log | tail -5
timeout 300 python __graft_entry__.py smoke > $out/smoke.log 2>&1; echo "smoke rc=$?" >> $out/smoke.log; tail -2 $out/smoke.log
timeout 600 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?" >> $out/bench.err
grep -o "\"value\": [0-9.]*\|\"ms_per_step\": [0-9.]*\|\"secondary\": {[^}]*}" $out/bench.json | head -5; tail -2 $out/bench.err
