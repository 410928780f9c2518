#!/bin/bash
mkdir -p gpurun_out/r04
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 200 python -m pytest tests/test_gpu_analytic.py tests/test_gpu_levels.py -m gpu -q -x --timeout 100 2>&1 | tail -5
GST_PLAN_TIMING=1 timeout 100 python tools/level_timing.py 2>&1 | grep -v "^\[plan\] [a-z+]* [0-9.]* s" | tail -6
GST_TEST_FORCE=wide=0 timeout 100 python tools/level_timing.py 2>&1 | tail -3
