#!/bin/bash
mkdir -p gpurun_out/r04
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python -m pytest tests/test_gpu_analytic.py tests/test_gpu_levels.py tests/test_general_params.py tests/test_gpu_lindblad.py -m gpu -q -x --timeout 120 2>&1 | tail -30 > gpurun_out/r04/pytest6.txt
tail -12 gpurun_out/r04/pytest6.txt
timeout 120 python tools/level_timing.py > gpurun_out/r04/level_timing6a.txt 2>&1
GST_TEST_FORCE=wide=0 timeout 120 python tools/level_timing.py > gpurun_out/r04/level_timing6b.txt 2>&1
cat gpurun_out/r04/level_timing6a.txt gpurun_out/r04/level_timing6b.txt
cd /tmp; export TMPDIR=/tmp
timeout 180 rocprofv3 --kernel-trace --stats -f csv -d $GRAFT_REPO_ROOT/gpurun_out/r04/wide_stats -o s -- python $GRAFT_REPO_ROOT/tools/level_timing.py > $GRAFT_REPO_ROOT/gpurun_out/r04/wide_stats.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/r04/wide_stats -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -12 {}'
find gpurun_out/r04 -name "*kernel_trace.csv" -size +2M -delete
