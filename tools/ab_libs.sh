#!/bin/bash
# development aid: A/B builds of libgstfwd on the bench workload, interleaved (box-to-box variation is +-6 %).
#   LIBS="a.so b.so ..." EMU="0 8" tools/ab_libs.sh
mkdir -p gpurun_out
for rep in 1 2 3; do
  for L in $LIBS; do
    for E in $EMU; do
      GST_LIBGSTFWD=$PWD/$L timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-fill --no-analytic --emulate-ranks $E 2>/dev/null \
        | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L emu=$E rep=$rep ms_per_step=%.3f kernel_ms=%.3f frac=%.3f' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac']))"
    done
  done
done
