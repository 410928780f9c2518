#!/bin/bash
# randomised parity on the final binary: tools/fuzz_gpu.py (every kind, new seeds), D = 64 once more under the opt-in
# register-resident backward walk, tools/fuzz_jtj.py (the normal-equation kernels incl. the round-6 reductions)
O=gpurun_out/r06fuzz; rm -rf $O; mkdir -p $O
timeout 900 python tools/fuzz_gpu.py 60 6000 2>&1 | tail -4 | tee $O/default.txt
timeout 600 python tools/fuzz_gpu.py 24 6100 d64 2>&1 | tail -4 | tee $O/d64.txt
GST_FUZZ_FORCE=chain_resident=1 timeout 600 python tools/fuzz_gpu.py 24 6200 d64 2>&1 | tail -4 | tee $O/d64_resident.txt
timeout 600 python tools/fuzz_gpu.py 30 6300 anyd 2>&1 | tail -4 | tee $O/anyd.txt
timeout 600 python tools/fuzz_gpu.py 6 6400 big 2>&1 | tail -4 | tee $O/big.txt
timeout 600 python tools/fuzz_jtj.py 80 66 2>&1 | tail -3 | tee $O/jtj.txt
