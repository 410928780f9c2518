#!/bin/bash
# development aid: one traced FD run (GST_FD_TRACE) on a 1/RANKS atom; env: RANKS, GST_TEST_FORCE
mkdir -p gpurun_out
GST_FD_TRACE=gpurun_out/trace_one.bin timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-analytic --no-host-fill --emulate-ranks ${RANKS:-8} > gpurun_out/tr_one.json 2> gpurun_out/tr_one.err
grep gstfwd gpurun_out/tr_one.err | tail -2
python tools/trace_stats.py gpurun_out/trace_one.bin
python - <<'PY'
import json
d = json.loads(open("gpurun_out/tr_one.json").read().strip().splitlines()[-1])
print("step_ms %.3f kernel_ms %.3f tasks %d" % (d["ms_per_step"], d["roofline"]["kernel_ms"], d["plan"]["n_tasks"]))
PY
