#!/bin/bash
# FD-of-FD Hessian pass compiled for 3 / 2 wavefronts per SIMD (2Q lite L<=256, 16 x 1616 block contracted on the device)
for rep in 1 2 3; do
  for L in $LIBS; do
    GST_LIBGSTFWD=$PWD/$L timeout 200 python tools/hess_timing.py 2>&1 | grep "objective Hessian block" | sed "s|^|$L: |"
  done
done
