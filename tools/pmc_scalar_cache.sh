#!/bin/bash
# development aid: scalar data cache counters of the FD walk (full design and a 1/8 atom); counter-only rocprofv3 passes
R=$PWD; O=$R/gpurun_out/pmc_sqc; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-host-fill --no-other-configs --no-cptplnd --no-lm-step --no-analytic --steps 1 --warmup 0"
timeout 200 rocprofv3 --kernel-trace --pmc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES -f csv -d $O/full -o s -- $B > $O/full.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES -f csv -d $O/r8 -o s -- $B --emulate-ranks 8 --emulate-rank 4 > $O/r8.log 2>&1
cd $R
python - <<'P'
import csv, glob, collections
for d in ("full", "r8"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in glob.glob("gpurun_out/pmc_sqc/%s/**/*counter_collection.csv" % d, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:50]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in acc.items():
        if "walk_kernel" in k:
            print(d, k, {c: "%.3g" % x for c, x in v.items()})
P
tail -3 $O/r8.log | cut -c1-200
find $O -name "*.csv" -size +1M -delete
