#!/bin/bash
# r05 session h: the refinement bench-bag parity test; a second default bench line on another box of the pool
exec < /dev/null
out=gpurun_out/r5h; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_e2e.py -q -x -s -k "benched_bag" > $out/pytest_bag.log 2>&1; grep "bench bag\|passed\|failed" $out/pytest_bag.log
timeout 600 python bench.py --no-cpu-baseline > $out/bench_n1.json 2> $out/bench.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/r5h/bench_n1.json'))
print({k:d[k] for k in ('value','ms_per_step','breakdown','traffic_build_matches')}, d['pipelined']['value'], d['secondary']['value'], d['roofline']['frac'])
PY
