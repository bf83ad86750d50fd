#!/bin/bash
# sanity sweep over batch sizes / models: remainders of every size, the class-token tail on both GEMM paths
cd $GRAFT_REPO_ROOT
for args in "--model ViT-B-32 --batch 128" "--batch 64" "--batch 200" "--batch 600 --steps 1" "--batch 37"; do   # (--batch 1 trips the reference's own `assert out.shape[0] > 1` in l2())
  echo "== $args"
  ( timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline $args 2>&1 | tail -1 | cut -c1-230 )
done
