#!/bin/bash
# Extra PMC evidence: matrix-pipe utilisation of the analytic MFMA kernel (2Q workload) and VALU utilisation of the
# D=64 kernels (tools/bench_configs.py).  Counter-only passes (PMC is never combined with other trace domains).
R=$PWD
OUT=$R/gpurun_out/prof_extra
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU -f csv -d $OUT/an_mfma -o s -- python $R/bench.py --no-cpu-baseline --steps 1 --warmup 0 --deriv analytic > $OUT/an_mfma.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES -f csv -d $OUT/cfg_valu -o s -- python $R/tools/bench_configs.py > $OUT/cfg_valu.log 2>&1
rocprofv3 --kernel-trace --stats -f csv -d $OUT/cfg_stats -o s -- python $R/tools/bench_configs.py > $OUT/cfg_stats.log 2>&1
ls $OUT/*
