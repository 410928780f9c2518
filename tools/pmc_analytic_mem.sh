#!/bin/bash
# development aid: L1 / TLB counters of the analytic contraction (counter-only passes, two counters each, every pass
# under its own timeout: a counter set the hardware cannot collect makes rocprofv3 abort and then hang in its finaliser)
R=$PWD; OUT=$R/gpurun_out/pmc_ana_mem; rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-analytic --no-host-fill --steps 1 --warmup 0 --deriv analytic"
i=0
for SET in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum"; do
  i=$((i+1))
  timeout -s KILL 100 rocprofv3 --kernel-trace --pmc $SET -f csv -d $OUT/p$i -o s -- $B > $OUT/p$i.log 2>&1 || echo "pass $i ($SET) failed or timed out"
done
cd $R
python - <<'PY'
import csv, glob, collections
out = collections.defaultdict(float)
for f in glob.glob("gpurun_out/pmc_ana_mem/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "analytic_mfma_kernel" in row["Kernel_Name"]:
            out[row["Counter_Name"]] += float(row["Counter_Value"])
print("PMC", dict(out))
PY
find $OUT -name "*.csv" -size +1M -delete
