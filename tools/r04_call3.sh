#!/bin/bash
mkdir -p gpurun_out/r04
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_gpu_levels.py tests/test_gpu_lindblad.py tests/test_gpu_analytic.py -m gpu -q --timeout 300 2>&1 | tail -30 > gpurun_out/r04/pytest3.txt
timeout 300 python tools/level_timing.py > gpurun_out/r04/level_timing.txt 2>&1
timeout 300 python bench.py --no-cpu-baseline --no-host-fill --no-other-configs --no-lm-step > gpurun_out/r04/bench3.json 2> gpurun_out/r04/bench3.err
tail -8 gpurun_out/r04/pytest3.txt
cat gpurun_out/r04/level_timing.txt
tail -c 600 gpurun_out/r04/bench3.err
