#!/bin/bash
# development aid: FD launch forms -- persistent per-SIMD queues (GST_FD_PERSIST=2) vs one workgroup per pair (=0) -- on the
# whole design and 1/N atoms, interleaved repeats (run-to-run noise on one box is about +-3 %)
mkdir -p gpurun_out
for R in ${RANKS:-0 2 4 8}; do
  for rep in 1 2 3; do
    for P in 2 0; do
      GST_FD_PERSIST=$P timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-analytic --emulate-ranks $R 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ranks=$R persist=$P step_ms %.3f kernel_ms %.3f' % (d['ms_per_step'], d['roofline']['kernel_ms']))"
    done
  done
done
