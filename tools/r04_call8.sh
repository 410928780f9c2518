#!/bin/bash
mkdir -p gpurun_out/r04
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd /tmp; export TMPDIR=/tmp
for w in 1 0; do
  GST_TEST_FORCE=wide=$w timeout 150 rocprofv3 --kernel-trace --stats -f csv -d $GRAFT_REPO_ROOT/gpurun_out/r04/wide${w}_stats -o s -- python $GRAFT_REPO_ROOT/tools/level_timing.py > $GRAFT_REPO_ROOT/gpurun_out/r04/wide${w}_stats.log 2>&1
  echo "== wide=$w"; find $GRAFT_REPO_ROOT/gpurun_out/r04/wide${w}_stats -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -7 {}' | cut -c1-150
done
cd $GRAFT_REPO_ROOT; find gpurun_out/r04 -name "*kernel_trace.csv" -size +2M -delete
