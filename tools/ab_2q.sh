#!/bin/bash
# A/B of two builds of the library on the 2Q exact Jacobian (bench.py --deriv analytic), interleaved in one call:
# A = tools/bin/libgstfwd_A.so (the previous build), B = the tree's
R=$PWD; O=$R/gpurun_out/ab2q; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-host-fill --no-other-configs --no-cptplnd --no-lm-step --no-fit-replay --steps 8 --warmup 2 --deriv analytic"
for rep in 1 2 3; do for v in A B; do
  L=$R/pygsti_amd/libgstfwd.so; [ $v = A ] && L=$R/tools/bin/libgstfwd_A.so
  GST_LIBGSTFWD=$L timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $O/p$v$rep -o s -- $B > $O/out$v$rep.json 2>/dev/null
  find $O/p$v$rep -name "*kernel_trace.csv" -delete
  echo "$v$rep: $(grep -h 'analytic_mfma_kernel' $O/p$v$rep/s_kernel_stats.csv | cut -d, -f2-4,6,7)  step $(python -c "import json;print(json.loads(open('$O/out$v$rep.json').read().strip().splitlines()[-1])['ms_per_step'])")"
done; done
