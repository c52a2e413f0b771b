#!/bin/bash
# PMC passes over the JPEG decoder's kernels (20 decodes of a 1600x1200 file): HBM traffic and where the waves' cycles go.
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 500 python tools/pmc_collect.py gpurun_out/jpeg_pmc.json --target "python tools/profile_jpeg.py 20 rgb" --scratch gpurun_out/jpeg_pmc_scratch \
  --extra SQ_WAVES,SQ_INSTS_VALU,SQ_INSTS_SALU,SQ_INSTS_LDS,SQ_INSTS_BRANCH SQ_WAVE_CYCLES,SQ_WAIT_ANY,SQ_WAIT_INST_ANY,SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU,SQ_ACTIVE_INST_SCA,SQ_ACTIVE_INST_LDS,SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE,SQ_BUSY_CYCLES,SQ_INSTS_VMEM_RD,SQ_INSTS_SMEM 2>&1 | tail -5
rm -rf gpurun_out/jpeg_pmc_scratch
python - <<'P'
import json
d = json.load(open("gpurun_out/jpeg_pmc.json"))
for r in d["kernels"]:
    print({k: (round(v, 1) if isinstance(v, float) else v) for k, v in r.items()})
P
