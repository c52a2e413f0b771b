#!/bin/bash
# r05 session b: packed-fp32 reproducer, full GPU suite on the refactored tree, seeded verify_checkpoint run, default bench line
exec < /dev/null
out=gpurun_out/r5b; mkdir -p $out
export TMPDIR=/tmp
timeout 300 tools/ubench/pk_vs_mfma 1000000 > $out/pk_vs_mfma.txt 2>&1; tail -14 $out/pk_vs_mfma.txt
timeout 1500 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1; tail -4 $out/pytest_gpu.log
timeout 600 python tools/verify_checkpoint.py --make-seeded /tmp/vc_seeded --resize 640 > $out/verify_checkpoint_seeded.txt 2>&1; tail -20 $out/verify_checkpoint_seeded.txt
timeout 600 python bench.py --no-cpu-baseline > $out/bench.json 2> $out/bench.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/r5b/bench.json'))
print({k:d[k] for k in ('value','ms_per_step','breakdown','pipelined','traffic_build_matches') if k in d}, d['secondary']['value'], d['secondary']['ms_per_step'])
PY
