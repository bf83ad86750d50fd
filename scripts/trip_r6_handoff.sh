#!/bin/bash
# Round 6: the handoff (x3 forward -> bf16 backward, precision bf16+x3fwd-first): parity tests, then the bench lines of the mixed modes on one box.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_fullsize.py tests/test_gpu_kernels.py -m gpu -q -x -p no:cacheprovider -k "handoff or x3 or mixed or softmax" ) 2>&1 | tail -15 | tee gpurun_out/handoff_tests.log
for p in bf16+x3fwd-first bf16+x3-first bf16 bf16+x3fwd-first bf16+x3-first bf16; do
  ( timeout 600 python bench.py --precision $p --steps 3 --warmup 1 --no-cpu-baseline --no-pmc ) > gpurun_out/bench_prec_$p.log 2>&1
  python - <<PY
import json
for l in open("gpurun_out/bench_prec_$p.log"):
    if l.startswith("{"):
        d = json.loads(l); r = d["roofline"]; pc = r["per_class"]
        top = sorted(pc.items(), key=lambda kv: -kv[1]["ms"])[:6]
        print("$p: %.1f img/s, %.1f ms per call; " % (d["value"], d["ms_per_step"]) + ", ".join("%s=%.1f" % (k, v["ms"]) for k, v in top) + "; handoff_export=%s" % (pc.get("handoff_export", {}).get("ms")))
PY
done 2>&1 | tee gpurun_out/bench_handoff.log
grep -h "x3fwd\|x3_first" gpurun_out/parity_metrics.jsonl 2>/dev/null | tail -3
