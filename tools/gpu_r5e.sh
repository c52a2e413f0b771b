#!/bin/bash
# r05 session e: gather schedule -- conv shape tests, refinement / coarse e2e, per-layer A/B against the flattened-K kernel, bench
exec < /dev/null
out=gpurun_out/r5e; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "conv" > $out/pytest_conv.log 2>&1; tail -3 $out/pytest_conv.log
timeout 900 python -m pytest tests/test_gpu_e2e.py -q -x -k "golden or planted or flattened or multiview or refine" > $out/pytest_e2e.log 2>&1; tail -3 $out/pytest_e2e.log
for m in split gather split gather; do
  timeout 200 python tools/bench_conv_layers.py $m s2 2>&1 | grep -v amdgpu.ids | tee -a $out/conv_layers.txt
  timeout 200 python tools/bench_conv_layers.py $m adap0 2>&1 | grep -v amdgpu.ids | tee -a $out/conv_layers.txt
done
timeout 600 python bench.py --no-cpu-baseline --no-rooflines > $out/bench.json 2> $out/bench.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/r5e/bench.json'))
print({k:d[k] for k in ('value','ms_per_step','breakdown','pipelined') if k in d}, d['secondary']['value'], d['secondary']['ms_per_step'])
PY
