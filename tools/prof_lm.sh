#!/bin/bash
# development aid: kernel statistics of the LM-step leg (fill + maps + normal equations) under rocprofv3
R=$PWD; OUT=$R/gpurun_out/lmprof; rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $OUT/stats -o s -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs --no-cptplnd --no-host-fill > $OUT/bench.json 2> $OUT/log.txt
cd $R
python - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/lmprof/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Name"]
        if any(k in n for k in ("jtj", "jtf", "objective_rows", "analytic_mfma", "scale_rows")):
            print("%-60s calls %4s avg %9.1f us  min %9.1f" % (n[:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
python -c "
import json; d=json.loads(open('gpurun_out/lmprof/bench.json').read().strip().splitlines()[-1]); print({k:v for k,v in d.get('lm_step').items() if k!='note'})"
