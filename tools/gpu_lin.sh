#!/bin/bash
out=gpurun_out/${1:-lin1}; mkdir -p $out
for m in 0 1 2 3; do DFSFM_LIN2_STAGGER=$m timeout 300 python tools/bench_linear.py 2>&1 | grep -v amdgpu > $out/stag_$m.log; done
paste -d'\n' $out/stag_0.log $out/stag_1.log $out/stag_2.log $out/stag_3.log | cut -c1-130
