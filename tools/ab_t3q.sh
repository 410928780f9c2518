#!/bin/bash
# A/B of two builds of the library on the three-qubit exact Jacobian, interleaved in one call (box-to-box differences are
# larger than most kernel changes): A = tools/bin/libgstfwd_A.so (the previous build), B = the tree's
R=$PWD; O=$R/gpurun_out/ab; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for rep in 1 2; do for v in A B; do
  L=$R/pygsti_amd/libgstfwd.so; [ $v = A ] && L=$R/tools/bin/libgstfwd_A.so
  GST_LIBGSTFWD=$L timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $O/p$v$rep -o s -- python $R/tools/t3q_quick.py x > $O/out$v$rep.json 2>/dev/null
  find $O/p$v$rep -name "*kernel_trace.csv" -delete
  echo "$v$rep: $(grep -h 'chain64\|mfma64' $O/p$v$rep/s_kernel_stats.csv | sed 's/.*kernel</k</; s/(gst[^,]*,[^,]*)//; s/(gst[^)]*)//' | cut -d, -f1,3,5,6 | tr '\n' ' ')"
  echo "$v$rep unprofiled: $(GST_LIBGSTFWD=$L python $R/tools/t3q_quick.py x | tail -1 | cut -c1-40)"
done; done
