#!/bin/bash
mkdir -p gpurun_out
for m in 1 0; do
  GST_FD_OVERLAP=$m GST_FD_TRACE=gpurun_out/trace_ovl$m.bin timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-analytic --no-host-fill --no-other-configs --no-cptplnd --emulate-ranks 8 > /dev/null 2>&1
done
python tools/trace_compare.py gpurun_out/trace_ovl1.bin gpurun_out/trace_ovl0.bin
