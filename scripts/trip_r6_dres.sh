#!/bin/bash
# Round 6: bf16-only residual-gradient stream in the attack backward (RVLM_DRES_FP32=1 restores the fp32 stream): parity metrics of the
# full-size tests under both settings, then alternating bench lines on one box.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; export TMPDIR=/tmp
for v in 1 0; do
  rm -f gpurun_out/parity_metrics.jsonl
  ( RVLM_DRES_FP32=$v timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_engine.py -m gpu -q -x -p no:cacheprovider ) 2>&1 | tail -4
  cp gpurun_out/parity_metrics.jsonl gpurun_out/parity_metrics_dres_fp32_$v.jsonl
done
for rep in 1 2 3; do for v in 1 0; do
  ( RVLM_DRES_FP32=$v timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-pmc ) > gpurun_out/bench_dres_$v.log 2>&1
  python - <<PY
import json
for l in open("gpurun_out/bench_dres_$v.log"):
    if l.startswith("{"):
        d = json.loads(l); pc = d["roofline"]["per_class"]
        print("RVLM_DRES_FP32=$v rep $rep: %.1f img/s, %.1f ms per call; layernorm_bwd %.1f ms, layernorm_fwd %.1f, fc2_bwd %.1f, out_bwd %.1f" % (d["value"], d["ms_per_step"], pc["layernorm_bwd"]["ms"], pc["layernorm_fwd"]["ms"], pc["gemm_fc2_bwd"]["ms"], pc["gemm_out_bwd"]["ms"]))
PY
done; done 2>&1 | tee gpurun_out/bench_dres_ab.log
