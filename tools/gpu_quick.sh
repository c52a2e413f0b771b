#!/bin/bash
# LoFTR end-to-end GPU tests + the default bench line (no CPU baseline)
exec < /dev/null
tag=${1:-q1}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_e2e.py -q -k "loftr or coarse or scene" > $out/pytest.log 2>&1; tail -3 $out/pytest.log
for m in 0 auto; do
  if [ $m = auto ]; then unset DFSFM_LN160; else export DFSFM_LN160=$m; fi
  timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-rooflines > $out/bench_$m.json 2> $out/bench_$m.err
  echo "LN160=$m $(grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"breakdown": {[^}]*}' $out/bench_$m.json | head -3 | tr '\n' ' ')"
done
