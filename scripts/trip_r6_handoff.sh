#!/bin/bash
# Round 6: the handoff (x3 forward -> bf16 backward; precisions bf16+x3fwd-first / bf16+x3fwd): bench lines of the precision ladder on one box.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; export TMPDIR=/tmp
for p in ${PRECS:-bf16+x3fwd-first bf16+x3-first bf16 bf16+x3fwd x3 bf16+x3fwd-first bf16+x3-first bf16 bf16+x3fwd x3}; do
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
