#!/bin/bash
# The GPU suite with every grown device buffer starting from 0xFF bytes (GST_TEST_FORCE poison=1): a test that passes
# only because fresh memory is zero fails here.  Tests that set GST_TEST_FORCE themselves override the variable (they run
# unpoisoned); everything else -- the bulk of the suite -- runs poisoned.
export HSA_ENABLE_IPC_MODE_LEGACY=0 GST_TEST_FORCE=poison=1
mkdir -p gpurun_out/poison
timeout 1200 python -m pytest tests -m gpu -q --timeout 300 2>&1 | grep -E "passed|failed|Error|assert|FAILED|^tests" | head -40 | tee gpurun_out/poison/pytest.txt
