#!/bin/bash
# the three-qubit exact Jacobian (tools/t3q_quick.py): D = 64 parity tests, kernel statistics, step time
R=$PWD; O=$R/gpurun_out/t3q; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "64 or 3q or three or chain" 2>&1 | tail -1
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $O/p -o s -- python $R/tools/t3q_quick.py x > $O/out.json 2>/dev/null
find $O/p -name "*kernel_trace.csv" -delete
cut -c1-150 $O/p/s_kernel_stats.csv | head -4
python $R/tools/t3q_quick.py x | tail -1 | cut -c1-200
