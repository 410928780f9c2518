R=$PWD; OUT=$R/gpurun_out/tr; rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
timeout -s KILL 150 rocprofv3 --kernel-trace -f csv -d $OUT/t -o s -- python $R/bench.py --no-cpu-baseline --no-analytic --no-host-fill --steps 2 --warmup 1 --deriv analytic > $OUT/t.log 2>&1 || echo failed
cd $R
python - <<'PY'
import csv, glob
rows=[]
for f in glob.glob("gpurun_out/tr/t/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40], r.get("Stream_Id","")))
rows.sort()
t0=rows[0][0]
for s,e,n,st in rows[-14:]:
    print("%9.3f ms  +%7.3f ms  %s  stream %s" % ((s-t0)/1e6, (e-s)/1e6, n, st))
PY
rm -rf $OUT/t
