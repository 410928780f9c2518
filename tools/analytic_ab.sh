#!/bin/bash
# development aid: the D = 16 analytic contraction -- two-circuit work items (GST_ANALYTIC_PAIRS) and germ-major order of
# the items (GST_ANALYTIC_GERM_ORDER) -- interleaved repeats on the 2Q design
for rep in 1 2; do for V in "1 1" "1 0" "0 0"; do set -- $V
  GST_ANALYTIC_PAIRS=$1 GST_ANALYTIC_GERM_ORDER=$2 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --deriv analytic 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pairs=$1 germ_order=$2 step_ms %.3f kernel_ms %.3f' % (d['ms_per_step'], d['roofline']['kernel_ms']))"
done; done
