#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_trainer.py -q -x -p no:cacheprovider -k "wgrad or gemm or trainer or weight or train" ) > gpurun_out/pytest_wgrad.log 2>&1
tail -5 gpurun_out/pytest_wgrad.log
( timeout 900 python bench.py --steps 3 --warmup 1 --mode train --no-cpu-baseline --no-roofline ) > gpurun_out/bench_train.log 2>&1
tail -1 gpurun_out/bench_train.log | cut -c1-300
( timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline ) > gpurun_out/bench_quick.log 2>&1
tail -1 gpurun_out/bench_quick.log | cut -c1-200
