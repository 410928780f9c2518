#!/bin/bash
# Collect the rocprofv3 evidence kept under profiles/ (run on the GPU box through gpurun; results land in
# gpurun_out/prof/, tools/summarize_profiles.py turns them into the committed summaries).
#   kernel-trace/stats runs and PMC runs are separate passes (PMC is never combined with other trace domains).
R=$PWD
OUT=$R/gpurun_out/prof
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-analytic"
rocprofv3 --kernel-trace --stats -f csv -d $OUT/fd_stats -o s -- $B --steps 5 --warmup 2 > $OUT/fd_stats.log 2>&1
rocprofv3 --kernel-trace --stats -f csv -d $OUT/an_stats -o s -- $B --steps 5 --warmup 2 --deriv analytic > $OUT/an_stats.log 2>&1
for mode in fd analytic; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c -f csv -d $OUT/pmc_${mode}_$c -o s -- $B --steps 1 --warmup 0 --deriv $mode > $OUT/pmc_${mode}_$c.log 2>&1
  done
done
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -f csv -d $OUT/pmc_fd_sq -o s -- $B --steps 1 --warmup 0 > $OUT/pmc_fd_sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -f csv -d $OUT/pmc_an_sq -o s -- $B --steps 1 --warmup 0 --deriv analytic > $OUT/pmc_an_sq.log 2>&1
ls -R $OUT | head -60
