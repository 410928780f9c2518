#!/bin/bash
# development aid: parity subset + 1-GPU bench + emulated 1/8-atom bench; prints the figures that matter
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_forwardsim.py -m gpu -x -q 2>&1 | tail -3
python bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" > gpurun_out/b1.json 2>gpurun_out/b1.err
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --emulate-ranks 8 "$@" > gpurun_out/b8.json 2>gpurun_out/b8.err
python - <<'PY'
import json
for f in ("b1", "b8"):
    try:
        d = json.loads(open("gpurun_out/%s.json" % f).read().strip().splitlines()[-1])
        print(f, "value %.4g step_ms %.3f probs_ms %.3f kernel_ms %.3f tasks %d" % (
            d["value"], d["ms_per_step"], d["probs_ms"], d["roofline"]["kernel_ms"], d["plan"]["n_tasks"]))
    except Exception as e:
        print(f, "ERR", e, open("gpurun_out/%s.err" % f).read()[-1500:])
PY
