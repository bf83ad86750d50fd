#!/bin/bash
# bench lines of the other BASELINE configurations and the training step (builder-run; the driver measures config 2)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() { tag=$1; shift; ( timeout 900 python ${BENCH_SCRIPT:-bench.py} --no-cpu-baseline "$@" ) > gpurun_out/bench_$tag.log 2>&1; echo "$tag rc=$?"; tail -c 420 gpurun_out/bench_$tag.log | head -c 300; echo; }
run c3 --attack apgd --steps 3 --warmup 1
run c5 --attack autopgd --iterations 100 --batch 256 --steps 1 --warmup 1 --no-roofline
run train --mode train --steps 3 --warmup 1
run b32 --model ViT-B-32 --steps 5 --warmup 2
BENCH_SCRIPT=scripts/bench_extra.py run l14_336 --model ViT-L-14-336 --batch 64 --steps 2 --warmup 1 --no-roofline
run fp32 --precision fp32 --steps 2 --warmup 1 --no-pmc
run mixed --precision bf16+fp32-first --steps 3 --warmup 1 --no-pmc
run x3 --precision x3 --steps 2 --warmup 1 --no-pmc
run mixed_x3 --precision bf16+x3-first --steps 3 --warmup 1 --no-pmc
run mixed_x3fwd --precision bf16+x3fwd-first --steps 3 --warmup 1 --no-pmc
