#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash scripts/pmc_run.sh g1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" -- python $GRAFT_REPO_ROOT/scripts/gemm_one.py 8192 8192 8192 2 | grep -i "256p"
bash scripts/pmc_run.sh g2 "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM" -- python $GRAFT_REPO_ROOT/scripts/gemm_one.py 8192 8192 8192 2 | grep -i "256p"
bash scripts/pmc_run.sh g3 "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TCP_TCC_READ_REQ_sum TCP_TA_TCP_STATE_READ_sum" -- python $GRAFT_REPO_ROOT/scripts/gemm_one.py 8192 8192 8192 2 | grep -i "256p"
tail -3 gpurun_out/pmc_g3/log.txt
