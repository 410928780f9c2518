#!/bin/bash
# development: item dispatch order of the exact contraction (locality order vs longest bucket first)
R=$PWD; OUT=$R/gpurun_out/r06e; rm -rf $OUT; mkdir -p $OUT
Q="--no-cpu-baseline --no-host-fill --no-other-configs --no-cptplnd --no-fit-replay --no-lm-step"
cd /tmp; export TMPDIR=/tmp
for v in 0 1; do
  GST_TEST_FORCE=lpt=$v timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $OUT/s$v -o s -- python $R/bench.py $Q --deriv analytic --steps 7 --warmup 2 > $OUT/s$v.log 2>&1
  echo "lpt=$v: $(grep analytic_mfma_kernel $OUT/s$v/s_kernel_stats.csv | cut -d, -f2-7)"
  find $OUT -name "*kernel_trace.csv" -delete
done
