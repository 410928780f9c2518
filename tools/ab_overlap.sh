#!/bin/bash
# 1/8 atom: base pass inside the persistent launch (GST_FD_OVERLAP=1) against in front of it (=0), interleaved repeats
Q="--no-cpu-baseline --no-host-fill --no-other-configs --no-cptplnd --no-analytic --emulate-ranks ${RANKS:-8} --steps 10 --warmup 3"
for rep in 1 2 3; do
  for m in 1 0; do
    GST_FD_OVERLAP=$m timeout 120 python bench.py $Q 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('overlap=$m rep=$rep step %.3f kernel %.3f' % (b['ms_per_step'], b['roofline']['kernel_ms']))"
  done
done
