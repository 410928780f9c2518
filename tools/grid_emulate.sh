#!/bin/bash
# What one rank of an 8-GPU job does per step under different processor grids (one GPU, --emulate-ranks), and the
# self-launched 2-rank grid run end to end (ranks share the GPU: plumbing, not speed).  Results: gpurun_out/grid/
R=$PWD; OUT=$R/gpurun_out/grid; rm -rf $OUT; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
Q="--no-cpu-baseline --no-host-fill --no-other-configs --no-cptplnd --no-analytic --steps 10 --warmup 3"
python bench.py $Q --emulate-ranks 8 > $OUT/atoms8_r0.json 2>/dev/null
for r in 0 1; do python bench.py $Q --emulate-ranks 8 --grid 4x2 --emulate-rank $r > $OUT/grid4x2_r$r.json 2>/dev/null; done
for r in 0 1 2 3; do python bench.py $Q --emulate-ranks 8 --grid 2x4 --emulate-rank $r > $OUT/grid2x4_r$r.json 2>/dev/null; done
for r in 0 1; do python bench.py $Q --emulate-ranks 4 --grid 2x2 --emulate-rank $r > $OUT/grid2x2_r$r.json 2>/dev/null; done
python bench.py --gpus 2 --grid 1x2 --jtj --steps 3 --warmup 1 --no-cpu-baseline > $OUT/two_ranks_grid1x2.json 2> $OUT/two_ranks_grid1x2.err
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob(os.path.join(os.environ.get("OUT", "gpurun_out/grid"), "*.json"))):
    try:
        b = [json.loads(l) for l in open(f) if l.startswith("{")][-1]
        print(os.path.basename(f), "ms/step %.3f" % b["ms_per_step"], "kernel", b["roofline"].get("kernel_ms"), b["config"]["parallelism"], json.dumps(b.get("normal_equations"))[:400] if "two_ranks" in f else "")
    except Exception as e:
        print(os.path.basename(f), "FAILED", e)
PY
tail -5 $OUT/two_ranks_grid1x2.err
