#!/bin/bash
# development aid: the 1Q configuration (BASELINE configs[1]); with PROFILE=1 also its kernel times (rocprofv3)
mkdir -p gpurun_out
python -c "
import json, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tools')
import bench_configs as B
print(json.dumps(B.one_q()))"
if [ -n "$PROFILE" ]; then
  R=$PWD; cd /tmp; export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats -f csv -d $R/gpurun_out/prof_1q -o oneq -- python -c "
import sys
sys.path.insert(0, '$R'); sys.path.insert(0, '$R/tools')
import bench_configs as B
B.one_q()" > /dev/null 2>&1
  cd $R
  find gpurun_out/prof_1q -name "*kernel_stats.csv" | head -1 | xargs cat | head -8
fi
