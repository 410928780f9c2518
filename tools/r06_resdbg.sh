#!/bin/bash
# ablations of chain64_resident_kernel (wrong results by construction): 1 = no MFMA, 2 = no state-cache stores, 4 = no barrier,
# 8 = the backward pass AFTER the forward pass instead of beside it (standalone kernel durations)
R=$PWD; O=$R/gpurun_out/r06resdbg; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for cfg in "1 8" "1 24" "1 0" "0 0"; do
  set -- $cfg
  GST_TEST_FORCE=chain_resident=$1,tile_dbg=$2 timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $O/p$1_$2 -o s -- python $R/tools/t3q_quick.py > $O/out$1_$2.json 2>/dev/null
  find $O/p$1_$2 -name "*kernel_trace.csv" -delete
  echo "resident=$1 dbg=$2 step $(python -c "import json;print(json.load(open('$O/out$1_$2.json'))['step_ms'])")"; grep 'chain64' $O/p$1_$2/*kernel_stats.csv | sed 's/.*kernel/kernel/' | cut -c1-120
done
