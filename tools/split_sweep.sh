#!/bin/bash
# development aid: parity + timing of the FD row-split variants (GST_FD_SPLIT = 1, 2, 4)
mkdir -p gpurun_out
for sp in 4 2; do
  echo "== parity with GST_FD_SPLIT=$sp"; GST_FD_SPLIT=$sp python -m pytest tests/test_gpu_parity.py tests/test_gpu_forwardsim.py tests/test_gpu_ragged.py -m gpu -x -q 2>&1 | tail -2
done
for er in 8 4 2 1; do
  for sp in 1 2 4; do
    GST_FD_SPLIT=$sp python bench.py --steps 5 --warmup 2 --no-cpu-baseline --emulate-ranks $er > gpurun_out/s.json 2>gpurun_out/s.err
    python - "$er" "$sp" <<'PY'
import json, sys
try:
    d = json.loads(open("gpurun_out/s.json").read().strip().splitlines()[-1])
    print("atoms %s split %s: step_ms %.3f kernel_ms %.3f value %.4g" % (sys.argv[1], sys.argv[2], d["ms_per_step"], d["roofline"]["kernel_ms"], d["value"]))
except Exception as e:
    print("ERR", e, open("gpurun_out/s.err").read()[-800:])
PY
  done
done
