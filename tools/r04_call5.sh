#!/bin/bash
# whole GPU suite after the switch consolidation + level timing
mkdir -p gpurun_out/r04
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests -m gpu -q --timeout 200 2>&1 | tail -40 > gpurun_out/r04/pytest5.txt
timeout 120 python tools/level_timing.py > gpurun_out/r04/level_timing5.txt 2>&1
tail -12 gpurun_out/r04/pytest5.txt
cat gpurun_out/r04/level_timing5.txt
