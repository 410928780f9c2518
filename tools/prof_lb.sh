#!/bin/bash
# exact CPTPLND Jacobian of the bench design: correctness tests of the chain rule, then rocprofv3 kernel statistics
R=$PWD; O=$R/gpurun_out/lbprof; rm -rf $O; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_general_params.py tests/test_gpu_lindblad.py tests/test_gpu_adapter_modes.py -m gpu -q --timeout 300 2>&1 | grep -E "passed|failed|Error|assert|FAILED" | head -12
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $O/p -o s -- python $R/tools/lb_analytic_profile.py > $O/out.txt 2>&1
cd $R
grep "analytic CPTPLND" $O/out.txt
find $O -name "*kernel_stats.csv" | head -1 | xargs cat | cut -c1-150 | head -12
find $O -name "*kernel_trace.csv" -delete
