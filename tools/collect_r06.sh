#!/bin/bash
# Round-6 evidence, collected on the GPU box in one call (results under gpurun_out/r06f/, summarised into profiles/ by
# tools/summarize_r06.py).  PMC passes are counter-only (never combined with other trace domains).
R=$PWD
OUT=$R/gpurun_out/r06f
rm -rf $OUT; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1200 python -X faulthandler -m pytest tests -m gpu -q -s --timeout 600 -p no:cacheprovider > $OUT/pytest_full.txt 2>&1
grep -E "passed|failed|error" $OUT/pytest_full.txt | tail -3 | tee $OUT/pytest.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -1 $OUT/smoke.txt | cut -c1-160
POISON_ALL=1 N=${N:-12} bash tools/soak_suite.sh 2>&1 | tee $OUT/soak.txt
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err
Q="--no-cpu-baseline --no-host-fill --no-other-configs --no-cptplnd --no-fit-replay"
timeout 200 python bench.py $Q --deriv analytic --steps 10 --warmup 3 > $OUT/bench_analytic.json 2>/dev/null
GST_TEST_FORCE=tiles=1 timeout 200 python bench.py $Q --no-lm-step --deriv analytic --steps 10 --warmup 3 > $OUT/bench_analytic_tiles.json 2>/dev/null
N=8 bash tools/emulate_all_ranks.sh > $OUT/emulate_all_ranks.txt 2>&1
timeout 400 python bench.py --gpus 2 --steps 5 --warmup 2 --no-fit-replay > $OUT/two_ranks_one_gpu.json 2> $OUT/two_ranks_one_gpu.err
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-host-fill --no-other-configs --no-cptplnd --no-lm-step --no-fit-replay"
timeout 200 rocprofv3 --kernel-trace --stats -f csv -d $OUT/fd_stats -o s -- $B --no-analytic --steps 5 --warmup 2 > $OUT/fd_stats.log 2>&1
timeout 200 rocprofv3 --kernel-trace --stats -f csv -d $OUT/an_stats -o s -- $B --steps 5 --warmup 2 --deriv analytic > $OUT/an_stats.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $OUT/cfg_stats -o s -- python $R/tools/bench_configs.py > $OUT/cfg_stats.log 2>&1
for mode in fd analytic; do
  X="--no-analytic"; [ $mode = analytic ] && X=""
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 200 rocprofv3 --kernel-trace --pmc $c -f csv -d $OUT/pmc_${mode}_$c -o s -- $B $X --steps 2 --warmup 1 --deriv $mode > $OUT/pmc_${mode}_$c.log 2>&1
  done
done
timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -f csv -d $OUT/pmc_fd_sq -o s -- $B --no-analytic --steps 1 --warmup 0 > $OUT/pmc_fd_sq.log 2>&1
cd $R
find $OUT -name "*kernel_trace.csv" -size +2M -delete
find $OUT -name "*counter_collection.csv" -size +8M -delete
du -sh $OUT; ls $OUT | head -60
