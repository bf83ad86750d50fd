#!/bin/bash
# A/B on ONE box (boxes of the pool differ by up to 5 %): the headline bench alternating between two settings.
#   env switch:       AB_VAR=RVLM_GEMM_PINGPONG AB_VALS="0 1" bash scripts/trip_ab.sh
#   two library builds (AB_VAR=RVLM_LIB_PATH): AB_VALS="robustvlm_amd/librvlm_base.so robustvlm_amd/librvlm.so" bash scripts/trip_ab.sh
# SKIP_TESTS=1 skips the kernel/engine parity tests in front; AB_REPS (default 2) alternations; BENCH_ARGS are passed on.
# Output: gpurun_out/ab.log (one line per run: value, ms per step, GEMM roofline TFLOP/s, the twelve largest classes).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; rm -f gpurun_out/ab.log
if [ "${SKIP_TESTS:-0}" != "1" ]; then
( timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_engine.py -m gpu -q -x -p no:cacheprovider ${PYTEST_ARGS:-} ) > gpurun_out/pytest_gpu.log 2>&1
tail -3 gpurun_out/pytest_gpu.log
fi
for rep in $(seq 1 ${AB_REPS:-2}); do
for v in ${AB_VALS:-0 1}; do
tag=$(basename $v)
( export ${AB_VAR:-RVLM_GEMM_TAIL}=$v; timeout 600 python bench.py --steps ${AB_STEPS:-3} --warmup 1 --no-cpu-baseline ${BENCH_ARGS:-} ) > gpurun_out/bench_ab_$tag.log 2>&1
python - <<PY | tee -a gpurun_out/ab.log
import json
ls=[x for x in open('gpurun_out/bench_ab_$tag.log') if x.startswith('{')]
if not ls:
    print("${AB_VAR:-RVLM_GEMM_TAIL}=$v rep $rep: FAILED", open('gpurun_out/bench_ab_$tag.log').read()[-400:])
else:
    d=json.loads(ls[-1])
    pc=d['roofline']['per_class']
    print("${AB_VAR:-RVLM_GEMM_TAIL}=$v rep $rep:", round(d['value'],2), round(d['ms_per_step'],2), round(d['roofline']['achieved'],1),
          ' '.join(f"{k}={v['ms']:.2f}" for k,v in sorted(pc.items(), key=lambda kv:-kv[1]['ms'])[:12]))
PY
done
done
