#!/bin/bash
out=gpurun_out/${1:-pmclin}; mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
EX="SQ_WAVE_CYCLES,SQ_WAIT_ANY,SQ_WAIT_INST_ANY,SQ_ACTIVE_INST_ANY,SQ_VALU_MFMA_BUSY_CYCLES,SQ_BUSY_CYCLES,SQ_WAIT_INST_LDS,SQ_INSTS_LDS"
EX2="TCC_HIT_sum,TCC_MISS_sum,TCC_EA0_RDREQ_sum,TCC_EA0_WRREQ_sum"
EX3="SQ_LDS_BANK_CONFLICT,SQ_LDS_IDX_ACTIVE,GRBM_GUI_ACTIVE,SQ_INSTS_VALU,SQ_INSTS_VMEM_RD,SQ_INSTS_VMEM_WR"
python tools/pmc_collect.py $out/mlp0.json --target "python tools/pmc_linear.py 1800000 256 256 relu" --extra $EX $EX2 $EX3 --scratch $out/s1 > $out/log1.txt 2>&1
python tools/pmc_collect.py $out/qkv.json --target "python tools/pmc_linear.py 76800 256 768 f32" --extra $EX $EX2 $EX3 --scratch $out/s2 > $out/log2.txt 2>&1
rm -rf $out/s1 $out/s2
tail -3 $out/log1.txt; cat $out/mlp0.json | head -60
