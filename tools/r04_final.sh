#!/bin/bash
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests -m gpu -q --timeout 300 2>&1 | tail -25 > gpurun_out/r04/pytest_final.txt
tail -8 gpurun_out/r04/pytest_final.txt
bash tools/collect_r04.sh 2>&1 | tail -5
