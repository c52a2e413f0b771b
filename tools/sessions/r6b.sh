#!/bin/bash
# r06 session b: fine_match focus experiments around the strong culprit of session a (packed op #25), then the weak ones.
exec < /dev/null
tag=${1:-r6b}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp FINE_BISECT_DIR=/tmp/fine_bisect
timeout 600 python tools/studies/fine_bisect.py focus > $out/focus25.log 2>&1; echo "rc=$?" >> $out/focus25.log; cat $out/focus25.log | grep -v amdgpu.ids
for k in 23 30 31; do
  FINE_FOCUS=$k FINE_RUNS=100 timeout 400 python tools/studies/fine_bisect.py focus > $out/focus$k.log 2>&1; echo "rc=$?" >> $out/focus$k.log; grep "^focus\|^#" $out/focus$k.log
done
