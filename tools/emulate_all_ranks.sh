#!/bin/bash
# every rank's share of an N-atom job, one after the other on one GPU: the N-GPU step is the slowest of them
N=${N:-8}
Q="--no-cpu-baseline --no-host-fill --no-other-configs --no-cptplnd --no-analytic --no-fit-replay --no-lm-step --steps 10 --warmup 3"
for r in $(seq 0 $((N-1))); do
  timeout 120 python bench.py $Q --emulate-ranks $N --emulate-rank $r 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rank $r of $N: step %.3f ms kernel %.3f  tasks %d  applies/pass %d  nE %d' % (b['ms_per_step'], b['roofline']['kernel_ms'], b['plan']['n_tasks'], b['plan']['applies_per_pass'], b['plan']['n_elements']))"
done
