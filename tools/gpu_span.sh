#!/bin/bash
# span_attention A/B: narrow vs wide gathers, one pair per call and the 8-pairs-per-pass scene path; kernel tests
exec < /dev/null
tag=${1:-sp1}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_aspan.py -q -k "span_attention" > $out/pytest.log 2>&1; tail -2 $out/pytest.log
for wide in 0 1; do
  DFSFM_SPAN_WIDE=$wide timeout 300 python tools/profile_step.py aspan 9 2>&1 | tail -1
  DFSFM_SPAN_WIDE=$wide timeout 300 python tools/profile_step.py aspan_scene 9 2>&1 | tail -1
done
