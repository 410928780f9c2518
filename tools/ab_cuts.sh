#!/bin/bash
# hand-over cut policies on a 1/8 atom (env switches of pack_fd_queues), one run each + a repeat of the default
Q="--no-cpu-baseline --no-host-fill --no-other-configs --no-cptplnd --no-analytic --emulate-ranks 8 --steps 10 --warmup 3"
run() {
  env "$@" GST_FD_DEBUG=1 timeout 120 python bench.py $Q 2> /tmp/err.txt | python -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-60s step %.3f kernel %.3f' % ('$*', b['ms_per_step'], b['roofline']['kernel_ms']))"
  grep "per-SIMD queues" /tmp/err.txt | tail -1
}
run A=0
run GST_FD_CUT_RICH=1
run GST_FD_CUT_RICH=1 GST_FD_CUT_FRAC=0.6
run GST_FD_CUT_RICH=1 GST_FD_CUT_FRAC=0.4
run GST_FD_CUT_FRAC=0.6
run GST_FD_CUT_FRAC=0.4
run GST_FD_CUT_RICH=1 GST_FD_CUT=1
run GST_FD_CUT_RICH=1 GST_FD_CUT=2
run GST_FD_CUT_GAIN=100
run A=0
