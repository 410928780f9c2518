#!/bin/bash
# The round's closing run on the GPU box: the whole GPU suite, then the evidence collection (tools/collect_r04.sh).
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests -m gpu -q --timeout 300 2>&1 | grep -v "version\|Hostname\|Librccl" | tail -25 > /tmp/pytest_final.txt
tail -4 /tmp/pytest_final.txt
bash tools/collect_r04.sh 2>&1 | tail -5
cp /tmp/pytest_final.txt gpurun_out/r04c/pytest_final.txt
