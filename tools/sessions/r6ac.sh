#!/bin/bash
# r06 session ac: the backbone's features as split planes straight into the refinement transformer (model.direct_features): tests, step timing
exec < /dev/null
tag=${1:-r6ac}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_e2e.py -q -x -k "multiview or refine" > $out/pytest.log 2>&1; tail -5 $out/pytest.log
for r in 1 2 3; do
python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-rooflines 2>/dev/null | grep -o "\"refinement_ms_per_step\": [0-9.]*\|\"ms_per_step\": [0-9.]*" | head -2 | tr "\n" " "; echo
done
