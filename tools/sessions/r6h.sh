#!/bin/bash
# r06 session h: SQ / LDS counters of the fused front kernel (what do its two workgroups per CU wait for?)
exec < /dev/null
tag=${1:-r6h}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python tools/pmc_collect.py $out/pmc_s2d_front.json --target "python tools/bench_s2d_front.py" --scratch $out/pmc_scratch \
  --extra SQ_WAVE_CYCLES,SQ_BUSY_CYCLES,SQ_WAIT_ANY,SQ_WAIT_INST_ANY,SQ_ACTIVE_INST_ANY,SQ_VALU_MFMA_BUSY_CYCLES \
          SQ_ACTIVE_INST_VALU,SQ_ACTIVE_INST_LDS,SQ_ACTIVE_INST_SCA,SQ_ACTIVE_INST_MISC,SQ_WAIT_INST_LDS,SQ_ACTIVE_INST_VMEM \
          SQ_INSTS_VALU,SQ_INSTS_MFMA,SQ_INSTS_LDS,SQ_INSTS_SALU,SQ_INSTS_SMEM,SQ_LDS_BANK_CONFLICT,SQ_LDS_IDX_ACTIVE \
          GRBM_GUI_ACTIVE,SQ_WAVES,SQ_INSTS_VMEM_RD,SQ_INSTS_VMEM_WR 2>&1 | tail -3
rm -rf $out/pmc_scratch
python - <<PY
import json
d=json.load(open("$out/pmc_s2d_front.json"))
for k in d["kernels"]:
    if "s2d_front" in k["kernel"] or "conv_gemm_sf_same_kernel<64" in k["kernel"]:
        print(json.dumps(k, indent=0))
PY
