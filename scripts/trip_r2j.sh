#!/bin/bash
mkdir -p gpurun_out
timeout 300 python scripts/gemm_waits.py 128 144 130 > gpurun_out/gemm_waits.log 2>&1
tail -14 gpurun_out/gemm_waits.log
timeout 300 python scripts/sched_probe.py 0 16 > gpurun_out/sched_probe2.log 2>&1
tail -8 gpurun_out/sched_probe2.log
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_BUSY_avr TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $set --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/pmc_gemm_$tag -o pmc -- python $GRAFT_REPO_ROOT/scripts/gemm_bench.py 2 > $GRAFT_REPO_ROOT/gpurun_out/pmc_gemm_$tag.log 2>&1
  tail -3 $GRAFT_REPO_ROOT/gpurun_out/pmc_gemm_$tag.log
done
ls -R $GRAFT_REPO_ROOT/gpurun_out | grep pmc | head
