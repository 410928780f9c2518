#!/bin/bash
# development aid: instruction mix / wait counters of the analytic contraction (counter-only passes)
R=$PWD; OUT=$R/gpurun_out/pmc_ana_mix; rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-analytic --no-host-fill --steps 1 --warmup 0 --deriv analytic"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_INSTS_LDS -f csv -d $OUT/a -o s -- $B > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -f csv -d $OUT/b -o s -- $B > $OUT/b.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS -f csv -d $OUT/c -o s -- $B > $OUT/c.log 2>&1
cd $R
python - <<'PY'
import csv, glob, collections
for d in "abc":
    out = collections.defaultdict(float)
    for f in glob.glob("gpurun_out/pmc_ana_mix/%s/**/*counter_collection.csv" % d, recursive=True):
        for row in csv.DictReader(open(f)):
            if "analytic_mfma_kernel" in row["Kernel_Name"]:
                out[row["Counter_Name"]] += float(row["Counter_Value"])
    print(d, dict(out))
    if not out:
        print(open("gpurun_out/pmc_ana_mix/%s.log" % d).read()[-600:])
PY
