#!/bin/bash
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/hbl -o t -- python $GRAFT_REPO_ROOT/scripts/hipblaslt_names.py > /tmp/hbl.log 2>&1
f=$(find /tmp/hbl -name "*kernel_stats.csv" | head -1)
cut -d, -f1-4 $f | head -12
