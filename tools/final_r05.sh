#!/bin/bash
# Round-5 closing check on the GPU box: the whole GPU suite (summary line kept), smoke(), the driver's bench command and the
# two-ranks-on-one-GPU line.  Results under gpurun_out/r05f/.
R=$PWD
OUT=$R/gpurun_out/r05f
rm -rf $OUT; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -X faulthandler -m pytest tests -m gpu -q > $OUT/pytest_full.txt 2> $OUT/pytest_err.txt
grep -E "passed|failed|error" $OUT/pytest_full.txt | tail -3 | tee $OUT/pytest.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -1 $OUT/smoke.txt | cut -c1-120
timeout 400 python bench.py > $OUT/bench.json 2> $OUT/bench.err
timeout 300 python bench.py --gpus 2 --steps 5 --warmup 2 > $OUT/two_ranks_one_gpu.json 2> $OUT/two_ranks_one_gpu.err
python - <<'PY'
import json
for f in ("bench", "two_ranks_one_gpu"):
    try:
        d = json.loads(open("gpurun_out/r05f/%s.json" % f).read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], json.dumps(d.get("lm_step", {}))[:400])
    except Exception as e:
        print(f, "FAILED", e)
PY
