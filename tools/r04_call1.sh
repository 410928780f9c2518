#!/bin/bash
# round 4, call 1: the whole GPU suite (new parity / depth-profile / fit-replay tests) + the default bench line
mkdir -p gpurun_out/r04
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests -m gpu -q --timeout 600 2>&1 | tail -40 > gpurun_out/r04/pytest1.txt
timeout 600 python bench.py > gpurun_out/r04/bench1.json 2> gpurun_out/r04/bench1.err
tail -5 gpurun_out/r04/pytest1.txt
tail -c 600 gpurun_out/r04/bench1.err
