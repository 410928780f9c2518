#!/bin/bash
# Round-3 evidence, collected on the GPU box in one call (results under gpurun_out/r03/, summarised into profiles/ by
# tools/summarize_r03.py).  PMC passes are counter-only (never combined with other trace domains).
R=$PWD
OUT=$R/gpurun_out/r03
rm -rf $OUT; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
python bench.py > $OUT/bench.json 2> $OUT/bench.err
python bench.py --jtj --no-cpu-baseline --no-host-fill --no-other-configs --no-cptplnd > $OUT/bench_jtj.json 2> $OUT/bench_jtj.err
GST_JTJ_SPARSE=0 python bench.py --jtj --no-cpu-baseline --no-host-fill --no-other-configs --no-cptplnd --no-analytic > $OUT/bench_jtj_dense.json 2>/dev/null
python bench.py --deriv analytic --keep-zeros --steps 10 --warmup 3 --no-cpu-baseline --no-host-fill --no-other-configs --no-cptplnd > $OUT/bench_analytic_keepzeros.json 2>/dev/null
python bench.py --deriv analytic --steps 10 --warmup 3 --no-cpu-baseline --no-host-fill --no-other-configs --no-cptplnd > $OUT/bench_analytic.json 2>/dev/null
Q="--no-cpu-baseline --no-host-fill --no-other-configs --no-cptplnd --no-analytic"
for rep in 1 2 3; do
  for E in 2 4 8; do
    python bench.py $Q --emulate-ranks $E --steps 10 --warmup 3 > $OUT/emu${E}_rep$rep.json 2>/dev/null
  done
  GST_FD_OVERLAP=0 python bench.py $Q --emulate-ranks 8 --steps 10 --warmup 3 > $OUT/emu8_nooverlap_rep$rep.json 2>/dev/null
done
python bench.py --gpus 2 --steps 5 --warmup 2 --jtj > $OUT/two_ranks_one_gpu.json 2> $OUT/two_ranks_one_gpu.err
( echo "## base pass inside the persistent launch (default)"; GST_FD_OVERLAP=1 RANKS=8 bash tools/trace_one.sh; echo; echo "## separate base pass (GST_FD_OVERLAP=0)"; GST_FD_OVERLAP=0 RANKS=8 bash tools/trace_one.sh ) > $OUT/fd_small_atom_trace.txt 2>&1
python tools/lb_timing.py full > $OUT/lindblad_timing.txt 2>&1
GST_LB_SHARE=0 python tools/lb_timing.py lite > $OUT/lindblad_timing_noshare_lite.txt 2>&1
python tools/lb_timing.py lite > $OUT/lindblad_timing_lite.txt 2>&1
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-analytic --no-host-fill --no-other-configs --no-cptplnd"
rocprofv3 --kernel-trace --stats -f csv -d $OUT/fd_stats -o s -- $B --steps 5 --warmup 2 > $OUT/fd_stats.log 2>&1
rocprofv3 --kernel-trace --stats -f csv -d $OUT/an_stats -o s -- $B --steps 5 --warmup 2 --deriv analytic --keep-zeros > $OUT/an_stats.log 2>&1
rocprofv3 --kernel-trace --stats -f csv -d $OUT/jtj_stats -o s -- $B --steps 2 --warmup 1 --jtj > $OUT/jtj_stats.log 2>&1
rocprofv3 --kernel-trace --stats -f csv -d $OUT/lb_stats -o s -- python $R/tools/lb_timing.py full > $OUT/lb_stats.log 2>&1
rocprofv3 --kernel-trace --stats -f csv -d $OUT/emu8_stats -o s -- $B --steps 5 --warmup 2 --emulate-ranks 8 > $OUT/emu8_stats.log 2>&1
for mode in fd analytic; do
  X=""; [ $mode = analytic ] && X="--keep-zeros"
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c -f csv -d $OUT/pmc_${mode}_$c -o s -- $B --steps 2 --warmup 1 --deriv $mode $X > $OUT/pmc_${mode}_$c.log 2>&1
  done
done
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -f csv -d $OUT/pmc_fd_sq -o s -- $B --steps 1 --warmup 0 > $OUT/pmc_fd_sq.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU -f csv -d $OUT/pmc_jtj -o s -- $B --steps 1 --warmup 0 --jtj > $OUT/pmc_jtj.log 2>&1
GST_JTJ_SPARSE=0 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU -f csv -d $OUT/pmc_jtj_dense -o s -- $B --steps 1 --warmup 0 --jtj > $OUT/pmc_jtj_dense.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -f csv -d $OUT/pmc_lb_sq -o s -- python $R/tools/lb_timing.py lite > $OUT/pmc_lb_sq.log 2>&1
cd $R
find $OUT -name "*kernel_trace.csv" -size +2M -delete
find $OUT -name "*counter_collection.csv" -size +8M -delete
du -sh $OUT; ls $OUT | head -60
