#!/bin/bash
# Round 6: the tile kernel of the D = 16 exact contraction -- parity against the item kernel, the analytic GPU tests, then the
# analytic bench leg with and without tiles.  Results: gpurun_out/r06c/.
R=$PWD
OUT=$R/gpurun_out/r06c
rm -rf $OUT; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -X faulthandler -m pytest tests/test_tiles.py tests/test_gpu_analytic.py tests/test_gpu_levels.py tests/test_gpu_jtj.py tests/test_gpu_general_dimension.py -m gpu -q -x --timeout 600 -p no:cacheprovider > $OUT/pytest.txt 2>&1
tail -15 $OUT/pytest.txt
Q="--no-cpu-baseline --no-host-fill --no-other-configs --no-cptplnd --no-fit-replay --no-lm-step"
GST_TEST_FORCE=tiles=1 timeout 300 python bench.py $Q --deriv analytic --steps 10 --warmup 3 > $OUT/bench_analytic_tiles.json 2> $OUT/bench_analytic_tiles.err
timeout 300 python bench.py $Q --deriv analytic --steps 10 --warmup 3 > $OUT/bench_analytic_items.json 2> $OUT/bench_analytic_items.err
python - <<'PY'
import json
for f in ("tiles", "items"):
    try:
        d = json.loads(open("gpurun_out/r06c/bench_analytic_%s.json" % f).read().strip().splitlines()[-1])
        print(f, "ms/step %.3f" % d["ms_per_step"], "kernel_ms %.3f" % d["roofline"]["kernel_ms"], "frac %.3f" % d["roofline"]["frac"])
    except Exception as e:
        print(f, "FAILED", e)
PY
tail -3 $OUT/bench_analytic_tiles.err
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $OUT/an_stats -o s -- python $R/bench.py $Q --deriv analytic --steps 5 --warmup 2 > $OUT/an_stats.log 2>&1
cd $R
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -8 {} | cut -c1-200'
find $OUT -name "*kernel_trace.csv" -size +2M -delete
