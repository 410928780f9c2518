#!/bin/bash
# instruction counts of the 1/8-atom FD launch with the base pass inside (GST_FD_OVERLAP=1), in front (=0), and the
# overlap kernel without its chains (=2): counter-only passes
R=$PWD; OUT=$R/gpurun_out/pmc_ovl; rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-analytic --no-host-fill --no-other-configs --no-cptplnd --emulate-ranks 8 --steps 2 --warmup 1"
for m in ${MODES:-1 0}; do
  GST_FD_OVERLAP=$m timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES -f csv -d $OUT/m$m -o s -- $B > $OUT/m$m.log 2>&1
done
cd $R
python - <<'PY'
import csv, glob, os
from collections import defaultdict
for d in sorted(glob.glob("gpurun_out/pmc_ovl/*/")):
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0]
            c = acc[k][row["Counter_Name"]]; c[0] += float(row["Counter_Value"]); c[1] += 1
    for k, cs in acc.items():
        if "walk_kernel<16, 1, 3" in k:
            print(os.path.basename(d.rstrip("/")), k[:60], {c: "%.4g" % (v[0] / v[1]) for c, v in cs.items()})
PY
find $OUT -name "*.csv" -size +1M -delete
