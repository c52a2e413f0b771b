#!/bin/bash
# r05 final session: full GPU suite, smoke, default bench line (with cpu_baseline), then tools/gpu_measure.sh (workloads, step profiles, PMC)
exec < /dev/null
out=gpurun_out/r5m; mkdir -p $out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1; tail -3 $out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > $out/smoke.log 2>&1; tail -3 $out/smoke.log
timeout 900 python bench.py > $out/bench_n1.json 2> $out/bench_n1.err; tail -c 400 $out/bench_n1.json; echo
bash tools/gpu_measure.sh r5m
timeout 600 python bench.py --no-cpu-baseline > $out/bench_n1_after_pmc.json 2> $out/bench_n1_after_pmc.err
python - <<'PY'
import json
for f in ('bench_n1','bench_n1_after_pmc'):
    d=json.load(open(f'gpurun_out/r5m/{f}.json'))
    print(f,{k:d.get(k) for k in ('value','ms_per_step','breakdown','traffic_build_matches')}, d['pipelined']['value'], d['secondary']['value'], d['roofline']['frac'])
PY
