#!/bin/bash
# development aid: FD launch forms -- persistent per-SIMD queues vs one workgroup per pair -- on the whole design and 1/N atoms
mkdir -p gpurun_out
for P in 1 0; do
  for R in ${RANKS:-0 2 4 8}; do
    GST_FD_PERSIST=$P timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-analytic --emulate-ranks $R > gpurun_out/ab_${P}_${R}.json 2> gpurun_out/ab_${P}_${R}.err
    python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/ab_${P}_${R}.json").read().strip().splitlines()[-1])
    print("persist=$P ranks=$R step_ms %.3f kernel_ms %.3f probs_ms %.3f" % (d["ms_per_step"], d["roofline"]["kernel_ms"], d["probs_ms"]))
except Exception as e:
    print("persist=$P ranks=$R ERR", e, open("gpurun_out/ab_${P}_${R}.err").read()[-800:])
PY
  done
done
