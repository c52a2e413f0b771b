#!/bin/bash
# roi_align: kernel tests + the refinement e2e tests that use the NHWC form + the kernels-only roofline lines
exec < /dev/null
tag=${1:-roi1}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -k "roi" > $out/pytest_k.log 2>&1; tail -2 $out/pytest_k.log
timeout 600 python -m pytest tests/test_gpu_e2e.py -q -k "refine or multiview" > $out/pytest_e.log 2>&1; tail -2 $out/pytest_e.log
timeout 600 python bench.py --kernels-only > $out/kernels_only.json 2> $out/kernels_only.err
python - <<PY
import json
d=json.loads(open("$out/kernels_only.json").read().strip().splitlines()[-1])
for r in d.get("rooflines", d.get("kernels", [])):
    print(r.get("kernel"), r.get("ms"), r.get("achieved"), r.get("frac"))
PY
