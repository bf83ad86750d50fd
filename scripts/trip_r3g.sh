#!/bin/bash
timeout 600 python -m pytest tests/test_gpu_engine.py -x -q -m gpu -k test_full_size_properties_vit_l14_bf16 2>&1 | tail -40
