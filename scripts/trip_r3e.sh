#!/bin/bash
timeout 200 python scripts/odd_debug.py 2>&1 | grep -v amdgpu.ids
