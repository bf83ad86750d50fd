#!/bin/bash
# In-pipeline utilisation evidence: rocprofv3 --pmc passes over the bench command itself
# (bench.py --steps 1 --warmup 0: the e0 forward + one 10-step pgd() call), --kernel-trace only, one counter group per pass.
# Summarised per kernel by scripts/pmc_pipeline_summary.py into gpurun_out/pmc_pipeline.json (copy to profiles/rNN_pmc_pipeline.json).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
i=0
for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" \
            "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"; do
  i=$((i+1)); out=$GRAFT_REPO_ROOT/gpurun_out/pmc_pipe$i
  rm -rf $out; mkdir -p $out
  ( cd /tmp && timeout 600 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $out -o pmc -- \
      python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline ${BENCH_ARGS:-} ) > $out/log.txt 2>&1
  echo "pass $i rc=$?"
  [ "${PMC_PASSES:-3}" = "$i" ] && break
done
python scripts/pmc_pipeline_summary.py gpurun_out/pmc_pipe* > gpurun_out/pmc_pipeline.json
tail -c 600 gpurun_out/pmc_pipeline.json
find gpurun_out/pmc_pipe* -name "*.csv" -size +8M -delete 2>/dev/null
