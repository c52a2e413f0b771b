#!/bin/bash
# r06 hardening on the final kernels: DFSFM_GUARD=1 (64-KB canary bands around every workspace, checked after every test) over the kernel, encoder and
# JPEG suites; the ASpanFormer suite with guard bands + serialized kernels
exec < /dev/null
out=gpurun_out/${1:-r6af}; mkdir -p $out
export TMPDIR=/tmp
{
echo "# r06 hardening on the final kernels (gpurun session r6af): DFSFM_GUARD=1 (64-KB canary bands around every workspace, checked after every test)"
echo "## tests/test_gpu_kernels.py tests/test_gpu_encoder_fused.py tests/test_gpu_encoder256.py tests/test_gpu_jpeg.py"
DFSFM_GUARD=1 timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_encoder_fused.py tests/test_gpu_encoder256.py tests/test_gpu_jpeg.py -q 2>&1 | tail -2
echo "## tests/test_gpu_aspan.py with AMD_SERIALIZE_KERNEL=3"
DFSFM_GUARD=1 AMD_SERIALIZE_KERNEL=3 timeout 600 python -m pytest tests/test_gpu_aspan.py -q 2>&1 | tail -2
echo "## refinement e2e (S2DNet front end, resident-weight enc_kv, direct split-plane features) with guard bands"
DFSFM_GUARD=1 timeout 1500 python -m pytest tests/test_gpu_e2e.py -q -k "multiview or refine" 2>&1 | tail -2
} | tee $out/hardening.txt
