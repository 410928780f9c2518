#!/bin/bash
# development aid: the D = 16 analytic contraction with / without two-circuit work items
python -m pytest tests/test_gpu_analytic.py tests/test_general_params.py tests/test_objective.py -m gpu -x -q 2>&1 | tail -2
for V in 1 0; do GST_ANALYTIC_PAIRS=$V python bench.py --steps 6 --warmup 2 --no-cpu-baseline --deriv analytic 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pairs=$V', d['ms_per_step'], d['roofline']['kernel_ms'], d['value'])"; done
