#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/tests_odd.log 2>&1
tail -12 gpurun_out/tests_odd.log
grep -h "shard" gpurun_out/parity_metrics.jsonl | tail -3
