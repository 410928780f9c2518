#!/bin/bash
# development aid: hardware counters of the J^T J kernel on a synthetic resident matrix (tools/jtj_sweep.py)
R=$PWD; OUT=$R/gpurun_out/jtjpmc; rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -oE "\b(SQ|TCP|TCC|TA|TD)_[A-Z0-9_]+" | sort -u > $OUT/avail.txt
wc -l $OUT/avail.txt
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS" "SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F64" "TCP_PENDING_STALL_CYCLES TCC_HIT_sum TCC_MISS_sum SQ_LDS_DATA_FIFO_FULL"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 200 rocprofv3 --kernel-trace --pmc $set -f csv -d $OUT/$tag -o s -- env PYTHONPATH=$R python $R/tools/jtj_sweep.py > $OUT/$tag.log 2>&1
  python - "$OUT/$tag" <<'PY'
import sys, glob, csv, collections
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "jtj_mfma" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(k, "per launch: %.4g (n=%d)" % (sum(v) / len(v), len(v)))
PY
done
