#!/bin/bash
# A/B of two builds on emulated 1/4 and 1/8 atoms, interleaved:  LIBS="a.so b.so" tools/ab_libs2.sh
Q="--no-cpu-baseline --no-host-fill --no-other-configs --no-cptplnd --no-analytic --steps 10 --warmup 3"
for rep in 1 2 3; do
  for L in $LIBS; do
    for E in ${EMU:-8 4}; do
      GST_LIBGSTFWD=$PWD/$L timeout 120 python bench.py $Q --emulate-ranks $E 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L emu=$E rep=$rep step %.3f kernel %.3f' % (b['ms_per_step'], b['roofline']['kernel_ms']))"
    done
  done
done
