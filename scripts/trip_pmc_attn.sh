#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash scripts/pmc_run.sh a1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" -- python $GRAFT_REPO_ROOT/scripts/attn_bench.py 257 | grep -i "attn_"
bash scripts/pmc_run.sh a2 "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM" -- python $GRAFT_REPO_ROOT/scripts/attn_bench.py 257 | grep -i "attn_"
bash scripts/pmc_run.sh a3 "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_MISC" -- python $GRAFT_REPO_ROOT/scripts/attn_bench.py 257 | grep -i "attn_"
