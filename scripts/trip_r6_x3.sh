#!/bin/bash
# Round 6: the split-bf16 ("x3") precision - parity tests, then the bench lines of fp32 / x3 / the two mixed modes on one box.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_fullsize.py -m gpu -q -x -p no:cacheprovider -k "x3 or mixed" ) 2>&1 | tail -15 | tee gpurun_out/x3_tests.log
for p in x3 fp32 bf16+x3-first bf16+fp32-first bf16; do
  ( timeout 600 python bench.py --precision $p --steps 2 --warmup 1 --no-cpu-baseline --no-pmc ) > gpurun_out/bench_prec_$p.log 2>&1
  python - <<PY
import json
for l in open("gpurun_out/bench_prec_$p.log"):
    if l.startswith("{"):
        d = json.loads(l); r = d["roofline"]; pc = r["per_class"]
        top = sorted(pc.items(), key=lambda kv: -kv[1]["ms"])[:8]
        print("$p: %.1f img/s, %.1f ms per call, gemm achieved %.0f TFLOP/s; " % (d["value"], d["ms_per_step"], r["achieved"]) + ", ".join("%s=%.1f" % (k, v["ms"]) for k, v in top))
PY
done 2>&1 | tee gpurun_out/bench_precisions.log
tail -3 gpurun_out/bench_prec_x3.log | cut -c1-600
