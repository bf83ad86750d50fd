#!/bin/bash
# Round 6: the K/V/O-prefetching fused attention backward (RVLM_ATTN_BWD_PF) - parity under both settings, micro-benchmark A/B with the
# phase split, in-pipeline A/B, then the full-size suite (CLIP-like fixture included).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; export TMPDIR=/tmp
for pf in 1 0; do
  ( export RVLM_ATTN_BWD_PF=$pf; timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_engine.py -m gpu -q -x -p no:cacheprovider -k "attention or attn or engine_vs_oracle" ) 2>&1 | tail -3 | tee -a gpurun_out/attn_pf_tests.log
done
SKIP_TESTS=1 AB_REPS=3 AB_LIBS="robustvlm_amd/librvlm_pf1.so robustvlm_amd/librvlm.so" AB_ENVS="RVLM_ATTN_BWD_PF=0|RVLM_ATTN_BWD_PF=1" bash scripts/trip_attn_ab.sh > /dev/null 2>&1
cp gpurun_out/attn_ab.log gpurun_out/attn_ab_pf.log
for pf in 0 1; do
  echo "== phase split, RVLM_ATTN_BWD_PF=$pf" >> gpurun_out/attn_ab_pf.log
  ( export RVLM_ATTN_BWD_PF=$pf RVLM_ATTN_TRACE=1; timeout 300 python scripts/attn_bench.py 2>&1 | grep -v amdgpu.ids | tail -12 ) >> gpurun_out/attn_ab_pf.log
done
grep -E "bwd|==" gpurun_out/attn_ab_pf.log | cut -c1-150
for rep in 1 2; do for pf in 0 1; do
  ( export RVLM_ATTN_BWD_PF=$pf; timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pmc ) > gpurun_out/bench_pf${pf}_$rep.log 2>&1
  python - <<PY
import json
for l in open("gpurun_out/bench_pf${pf}_$rep.log"):
    if l.startswith("{"):
        d = json.loads(l); pc = d["roofline"]["per_class"]
        print("PF=$pf rep $rep: %.1f img/s, attn_bwd %.2f ms, attn_fwd %.2f ms, clock %.0f" % (d["value"], pc["attn_bwd"]["ms"], pc["attn_fwd"]["ms"], d["roofline"]["clock_in_kernel"]["sclk_mhz_effective"]))
PY
done; done 2>&1 | tee gpurun_out/bench_pf_ab.log
rm -f gpurun_out/parity_metrics.jsonl
( timeout 1500 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -p no:cacheprovider --durations=5 ) > gpurun_out/pytest_fullsize.log 2>&1; tail -15 gpurun_out/pytest_fullsize.log
