#!/bin/bash
# development aid: A/B environment settings on the bench workload, interleaved.   VARS="A=1 B=0|A=0" EMU="0 8" tools/ab_env.sh
mkdir -p gpurun_out
IFS='|' read -ra SETS <<< "$VARS"
for rep in 1 2 3; do
  for S in "${SETS[@]}"; do
    for E in $EMU; do
      env $S timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-fill --no-analytic --no-cptplnd --no-other-configs --emulate-ranks $E 2>/dev/null \
        | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$S] emu=$E rep=$rep ms_per_step=%.3f kernel_ms=%.3f frac=%.3f' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac']))"
    done
  done
done
