#!/bin/bash
# round 4, call 2: whole GPU suite (level passes, parity, depth profiles, fit replay) + default bench + quick level timing
mkdir -p gpurun_out/r04
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1200 python -m pytest tests -m gpu -q --timeout 600 2>&1 | tail -60 > gpurun_out/r04/pytest2.txt
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r04/bench2.json 2> gpurun_out/r04/bench2.err
timeout 300 python tools/level_timing.py > gpurun_out/r04/level_timing.txt 2>&1
tail -25 gpurun_out/r04/pytest2.txt
tail -c 1500 gpurun_out/r04/bench2.err
cat gpurun_out/r04/level_timing.txt
