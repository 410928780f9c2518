#!/bin/bash
# Round-2 evidence, collected on the GPU box in one call (results under gpurun_out/r02/, summarised into profiles/ by
# tools/summarize_r02.py).  PMC passes are counter-only (never combined with other trace domains).
R=$PWD
OUT=$R/gpurun_out/r02
rm -rf $OUT; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
python bench.py > $OUT/bench.json 2> $OUT/bench.err
python bench.py --jtj --no-cpu-baseline --no-host-fill > $OUT/bench_jtj.json 2> $OUT/bench_jtj.err
for E in 2 4 8; do
  python bench.py --no-cpu-baseline --no-host-fill --emulate-ranks $E --steps 10 --warmup 3 > $OUT/emu$E.json 2>/dev/null
done
GST_FD_HANDOVER=0 python bench.py --no-cpu-baseline --no-host-fill --emulate-ranks 8 --steps 10 --warmup 3 > $OUT/emu8_nohandover.json 2>/dev/null
python tools/bench_configs.py > $OUT/configs.json 2> $OUT/configs.err
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 5 --warmup 2 --jtj > $OUT/two_ranks_one_gpu.json 2> $OUT/two_ranks_one_gpu.err
( echo "## persistent per-SIMD queues with walk hand-over (default)"; GST_FD_HANDOVER=1 RANKS=8 bash tools/trace_one.sh; echo; echo "## the same without hand-over (GST_FD_HANDOVER=0)"; GST_FD_HANDOVER=0 RANKS=8 bash tools/trace_one.sh ) > $OUT/fd_small_atom_trace.txt 2>&1
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-analytic --no-host-fill"
rocprofv3 --kernel-trace --stats -f csv -d $OUT/fd_stats -o s -- $B --steps 5 --warmup 2 > $OUT/fd_stats.log 2>&1
rocprofv3 --kernel-trace --stats -f csv -d $OUT/an_stats -o s -- $B --steps 5 --warmup 2 --deriv analytic > $OUT/an_stats.log 2>&1
for mode in fd analytic; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c -f csv -d $OUT/pmc_${mode}_$c -o s -- $B --steps 1 --warmup 0 --deriv $mode > $OUT/pmc_${mode}_$c.log 2>&1
  done
done
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -f csv -d $OUT/pmc_fd_sq -o s -- $B --steps 1 --warmup 0 > $OUT/pmc_fd_sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -f csv -d $OUT/pmc_an_sq -o s -- $B --steps 1 --warmup 0 --deriv analytic > $OUT/pmc_an_sq.log 2>&1
rocprofv3 --kernel-trace --stats -f csv -d $OUT/cfg_stats -o s -- python $R/tools/bench_configs.py > $OUT/cfg_stats.log 2>&1
cd $R
# keep what travels back small: the per-dispatch traces are not needed
find $OUT -name "*kernel_trace.csv" -size +2M -delete
du -sh $OUT; ls $OUT | head -40
