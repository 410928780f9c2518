R=$PWD; OUT=$R/gpurun_out/fin; rm -rf $OUT; mkdir -p $OUT
timeout 300 python bench.py > $OUT/bench.json 2> $OUT/bench.err
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-analytic --no-host-fill"
timeout -s KILL 150 rocprofv3 --kernel-trace --stats -f csv -d $OUT/an_stats -o s -- $B --steps 5 --warmup 2 --deriv analytic > $OUT/an_stats.log 2>&1 || echo "stats failed"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 150 rocprofv3 --kernel-trace --pmc $c -f csv -d $OUT/pmc_an_$c -o s -- $B --steps 1 --warmup 0 --deriv analytic > $OUT/pmc_an_$c.log 2>&1 || echo "pmc $c failed"
done
cd $R
find $OUT -name "*kernel_trace.csv" -size +2M -delete
python - <<'PY'
import csv, glob, json, collections
d = json.loads(open("gpurun_out/fin/bench.json").read().strip().splitlines()[-1])
print("BENCH", d["ms_per_step"], d["value"], d["roofline"]["frac"], d["analytic_dprobs"], d["host_fill"]["GBps"])
for f in glob.glob("gpurun_out/fin/an_stats/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "analytic_mfma" in row["Name"] or "walk_base" in row["Name"]: print("STATS", row["Name"][:50], row["Calls"], row["AverageNs"])
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    tot = collections.defaultdict(float)
    for f in glob.glob("gpurun_out/fin/pmc_an_%s/**/*counter_collection.csv" % c, recursive=True):
        for row in csv.DictReader(open(f)):
            if "analytic_mfma" in row["Kernel_Name"]: tot[row["Counter_Name"]] += float(row["Counter_Value"])
    print("PMC", c, dict(tot))
PY
