#!/bin/bash
# Round 6 probes: two-workgroups-per-CU GEMM form vs the shipped kernel (same process), LayerNorm forward limiter, and how many
# fill / copy launches belong to ONE pgd() call (kernel-trace stats of --steps 1 vs --steps 3).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 600 scripts/probes/gemm_two_wg_probe.bin robustvlm_amd/librvlm.so ) > gpurun_out/gemm_two_wg_probe.log 2>&1; tail -30 gpurun_out/gemm_two_wg_probe.log
( timeout 300 scripts/probes/ln_fwd_limiter_probe.bin 10 ) > gpurun_out/ln_fwd_limiter_probe.log 2>&1; tail -16 gpurun_out/ln_fwd_limiter_probe.log
for st in 1 3; do
  rm -rf gpurun_out/prof_s$st
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_s$st -o trace -- \
      python $GRAFT_REPO_ROOT/bench.py --steps $st --warmup 1 --no-cpu-baseline --no-roofline ) > gpurun_out/rocprof_s$st.log 2>&1
  find gpurun_out/prof_s$st -name "*kernel_trace.csv" -delete 2>/dev/null
  echo "steps $st:"; grep -h -E "fillBuffer|copyBuffer" $(find gpurun_out/prof_s$st -name "*kernel_stats.csv") | cut -d, -f1-4
done 2>&1 | tee gpurun_out/fill_copy_per_call.log
