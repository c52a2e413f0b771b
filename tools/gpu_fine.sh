#!/bin/bash
out=gpurun_out/${1:-fine1}; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "fine or resample" > $out/pytest.log 2>&1; echo "rc=$?" >> $out/pytest.log
timeout 600 python -m pytest tests/test_gpu_e2e.py -q -x -k "multiview or refine" -s >> $out/pytest.log 2>&1; echo "rc=$?" >> $out/pytest.log
DFSFM_FINE_FASTEXP=1 timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "fine" >> $out/pytest.log 2>&1; echo "fast rc=$?" >> $out/pytest.log
timeout 300 python bench.py --kernels-only > $out/kern.json 2> $out/kern.err
DFSFM_FINE_FASTEXP=1 timeout 300 python bench.py --kernels-only > $out/kern_fast.json 2> $out/kern.err
grep -n "passed\|failed\|rc=" $out/pytest.log
python -c "
import json
for f in ('kern','kern_fast'):
    d=json.load(open('$out/'+f+'.json'))
    for r in d['rooflines']:
        if 'fine' in r['kernel']: print(f, r['kernel'], round(r['ms'],3), 'ms', round(r['achieved']), r['unit'])
"
