#!/bin/bash
# r05 session f: fine_match spread, guard-band hardening of the final build
exec < /dev/null
out=gpurun_out/r5f; mkdir -p $out
export TMPDIR=/tmp
timeout 300 python tools/fine_spread.py 2>&1 | grep -v amdgpu.ids | tee $out/fine_spread.txt
DFSFM_GUARD=1 timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_encoder_fused.py tests/test_gpu_encoder256.py -q -x 2>&1 | tail -2 | tee $out/harden_kernels.txt
DFSFM_GUARD=1 AMD_SERIALIZE_KERNEL=3 timeout 600 python -m pytest tests/test_gpu_aspan.py -q -x 2>&1 | tail -2 | tee $out/harden_aspan.txt
