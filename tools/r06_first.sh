#!/bin/bash
# Round 6, first GPU call: the whole GPU suite once (with the new host-array stress test), then the poisoned soak of the
# suite (every run with poisoned device buffers), then the default bench line.  Results under gpurun_out/r06a/.
R=$PWD
OUT=$R/gpurun_out/r06a
rm -rf $OUT; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -X faulthandler -m pytest tests -m gpu -q -s --timeout 600 -p no:cacheprovider > $OUT/pytest_full.txt 2>&1
grep -E "passed|failed|error" $OUT/pytest_full.txt | tail -3 | tee $OUT/pytest.txt
POISON_ALL=1 N=${N:-8} bash tools/soak_suite.sh 2>&1 | tee $OUT/soak.txt
cp gpurun_out/soak/summary.txt $OUT/soak_summary.txt
timeout 400 python bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -c 600 $OUT/bench.json
