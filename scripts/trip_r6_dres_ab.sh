cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
rocminfo 2>/dev/null | grep -E "Marketing Name" | head -2
for rep in 1 2 3; do for v in 1 0; do
  ( RVLM_DRES_FP32=$v timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-pmc ) 2>&1 | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('RVLM_DRES_FP32=$v rep $rep: %.1f img/s, %.1f ms per call, clock %s MHz' % (d['value'], d['ms_per_step'], round(d['roofline'].get('clock_in_kernel',{}).get('sclk_mhz_effective',0))))"
done; done | tee gpurun_out/bench_dres_ab2.log
