#!/bin/bash
# round-2 trip F: PMC passes over the attention kernels (SQ busy / wait / instruction mix), separate rocprofv3 --pmc runs
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash scripts/pmc_run.sh at1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" -- python $GRAFT_REPO_ROOT/scripts/attn_bench.py | grep -i "attn"
bash scripts/pmc_run.sh at2 "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" -- python $GRAFT_REPO_ROOT/scripts/attn_bench.py | grep -i "attn"
bash scripts/pmc_run.sh at3 "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" -- python $GRAFT_REPO_ROOT/scripts/attn_bench.py | grep -i "attn"
