cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
( RVLM_LIB_PATH=$GRAFT_REPO_ROOT/robustvlm_amd/librvlm_bk32.so timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -p no:cacheprovider -k "gemm_f32" ) 2>&1 | tail -3
for rep in 1 2; do for lib in librvlm.so librvlm_bk32.so; do
  ( RVLM_LIB_PATH=$GRAFT_REPO_ROOT/robustvlm_amd/$lib timeout 600 python bench.py --precision fp32 --steps 2 --warmup 1 --no-cpu-baseline --no-pmc ) 2>&1 | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', round(d['value'],2), round(d['ms_per_step'],1), round(d['roofline']['achieved'],1))"
done; done
