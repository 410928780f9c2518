#!/bin/bash
# development aid: A/B builds / settings of the analytic contraction.  SETS="GST_LIBGSTFWD=... X=1|..." tools/ab_analytic.sh
IFS='|' read -ra S <<< "$SETS"
for rep in 1 2 3; do
  for V in "${S[@]}"; do
    env $V timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-fill --no-analytic --deriv analytic 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$V] rep=$rep ms_per_step=%.3f kernel_ms=%.3f' % (d['ms_per_step'], d['roofline']['kernel_ms']))"
  done
done
