#!/bin/bash
# A/B: memory type of the state cache / p_base under the in-launch base pass (1/8 atom), interleaved repeats
Q="--no-cpu-baseline --no-host-fill --no-other-configs --no-cptplnd --no-analytic --emulate-ranks 8 --steps 10 --warmup 3"
for rep in 1 2 3; do
  for m in 0 1 2; do
    GST_FD_OVL_MEM=$m python bench.py $Q 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('mem=$m rep=$rep step %.3f kernel %.3f' % (b['ms_per_step'], b['roofline']['kernel_ms']))"
  done
done
GST_FD_OVERLAP=0 python bench.py $Q 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('no overlap: step %.3f kernel %.3f' % (b['ms_per_step'], b['roofline']['kernel_ms']))"
GST_FD_OVERLAP=0 GST_FD_OVL_MEM=1 python bench.py $Q 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('no overlap, uncached: step %.3f kernel %.3f' % (b['ms_per_step'], b['roofline']['kernel_ms']))"
