R=$PWD; OUT=$R/gpurun_out/pmc_grp; rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-analytic --no-host-fill --steps 1 --warmup 0 --deriv analytic"
for G in 0 1; do
  GST_ANALYTIC_GROUP=$G timeout -s KILL 100 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum -f csv -d $OUT/g$G -o s -- $B > $OUT/g$G.log 2>&1 || echo "pass $G failed"
  GST_ANALYTIC_GROUP=$G timeout -s KILL 100 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum -f csv -d $OUT/h$G -o s -- $B > $OUT/h$G.log 2>&1 || echo "pass $G failed"
done
cd $R
python - <<'PY'
import csv, glob, collections
for G in "01":
    out = collections.defaultdict(float)
    for f in glob.glob("gpurun_out/pmc_grp/[gh]%s/**/*counter_collection.csv" % G, recursive=True):
        for row in csv.DictReader(open(f)):
            if "analytic_mfma_kernel" in row["Kernel_Name"]:
                out[row["Counter_Name"]] += float(row["Counter_Value"])
    print("PMC group=%s" % G, dict(out))
PY
find $OUT -name "*.csv" -size +1M -delete
