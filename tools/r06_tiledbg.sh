#!/bin/bash
# development: where the tile kernel's time goes (stores / remnants / segment stream switched off in turn)
R=$PWD; OUT=$R/gpurun_out/r06d; rm -rf $OUT; mkdir -p $OUT
Q="--no-cpu-baseline --no-host-fill --no-other-configs --no-cptplnd --no-fit-replay --no-lm-step"
cd /tmp; export TMPDIR=/tmp
for dbg in 0 1 2 3 4 7; do
  GST_TEST_FORCE=tile_dbg=$dbg timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $OUT/s$dbg -o s -- python $R/bench.py $Q --deriv analytic --steps 5 --warmup 2 > $OUT/s$dbg.log 2>&1
  echo "dbg=$dbg: $(grep analytic_tile_kernel $OUT/s$dbg/s_kernel_stats.csv | cut -d, -f2-4)"
  find $OUT -name "*kernel_trace.csv" -delete
done
