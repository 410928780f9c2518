#!/bin/bash
# development aid: HBM traffic of the J^T J kernels (FETCH_SIZE / WRITE_SIZE in separate counter-only passes)
R=$PWD; OUT=$R/gpurun_out/jtjhbm; rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --kernel-trace --pmc $c -f csv -d $OUT/$c -o s -- env PYTHONPATH=$R python $R/tools/jtj_sweep.py > $OUT/$c.log 2>&1
  python - "$OUT/$c" <<'PY'
import sys, glob, csv, collections
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        acc[(r["Kernel_Name"][:40], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        if "jtj" in k[0] or "jtf" in k[0]:
            print(k[0], k[1], "per launch: %.4g KB (n=%d)" % (sum(v) / len(v), len(v)))
PY
done
