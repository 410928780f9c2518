#!/bin/bash
# Round 6, second GPU call: the tests of what changed since the first (three-qubit Lindblad members, the Lindblad FD gate, the
# communicator's direct fan-in and self-send, the 2Q L<=64 fit replay), then the default bench line.  Results: gpurun_out/r06b/.
R=$PWD
OUT=$R/gpurun_out/r06b
rm -rf $OUT; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -X faulthandler -m pytest tests/test_gpu_lindblad64.py tests/test_gpu_lindblad.py tests/test_gpu_comm.py tests/test_fit_replay2q.py \
    tests/test_gpu_adapter_modes.py tests/test_gpu_models.py tests/test_gpu_composite.py tests/test_gpu_grid.py -m gpu -q -x --timeout 900 -p no:cacheprovider > $OUT/pytest.txt 2>&1
tail -25 $OUT/pytest.txt
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -5 $OUT/bench.err
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r06b/bench.json").read().strip().splitlines()[-1])
    print(json.dumps({k: d[k] for k in ("value", "ms_per_step", "config")}, indent=0)[:1500])
    print(json.dumps(d["roofline"], indent=0)[:1800])
    print(json.dumps(d["gst_fit_2Q_L64"], indent=0)[:1500])
    print(json.dumps(d["cptplnd_dprobs"], indent=0)[:800])
    print(json.dumps({k: v for k, v in d["cpu_baseline"].items() if k != "sample"}, indent=0)[:900])
except Exception as e:
    print("bench FAILED", e)
PY
