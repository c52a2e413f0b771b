#!/bin/bash
# ASpanFormer scene path: GPU tests of the matcher, then the 40-image scene (780 pairs) at several pairs-per-pass settings.
exec < /dev/null
tag=${1:-as1}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_aspan.py -q -x > $out/pytest.log 2>&1; tail -3 $out/pytest.log
for ppp in ${PPP:-1 4 8 16}; do
  DFSFM_ASPAN_PAIRS_PER_PASS=$ppp timeout 300 python bench.py --workload scene300 --scene-matcher aspanformer --scene-images 40 --batch 16 > $out/scene40_ppp$ppp.json 2> $out/scene40_ppp$ppp.err
  echo "ppp=$ppp $(grep -o '"value": [0-9.]*\|"match_rows": [0-9]*\|"keypoints": [0-9]*' $out/scene40_ppp$ppp.json | tr '\n' ' ')"
done
