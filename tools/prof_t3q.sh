#!/bin/bash
# rocprofv3 kernel statistics of tools/t3q_quick.py (the 3-qubit exact Jacobian alone)
R=$PWD; O=$R/gpurun_out/t3qprof; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $O/p -o s -- python $R/tools/t3q_quick.py > $O/out.json 2>/dev/null
cd $R
find $O -name "*kernel_stats.csv" | head -1 | xargs cat | cut -c1-150
find $O -name "*kernel_trace.csv" -delete
tail -1 $O/out.json
