#!/bin/bash
# the three-qubit exact Jacobian with the default backward walk (chain64_mfma_kernel beside the forward pass) and with the
# opt-in register-resident one (chain64_resident_kernel after the forward pass): parity tests, step time, kernel statistics;
# then what a store-only kernel reaches on this part (the ceiling of the contraction's 10.6 GB of output)
R=$PWD; O=$R/gpurun_out/r06res; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "64 or 3q or three or chain" > $O/pytest.txt 2>&1; tail -1 $O/pytest.txt
GST_TEST_FORCE=chain_resident=1 timeout 900 python -m pytest tests -m gpu -x -q -k "64 or 3q or three or chain" > $O/pytest_resident.txt 2>&1; tail -1 $O/pytest_resident.txt
cd /tmp; export TMPDIR=/tmp
for v in 0 1; do
  GST_TEST_FORCE=chain_resident=$v timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $O/p$v -o s -- python $R/tools/t3q_quick.py x > $O/out$v.json 2>/dev/null
  find $O/p$v -name "*kernel_stats.csv" | head -1 | xargs cat | cut -c1-150 > $O/stats$v.csv
  find $O/p$v -name "*kernel_trace.csv" -delete
  echo "resident=$v"; head -4 $O/stats$v.csv; tail -1 $O/out$v.json
  GST_TEST_FORCE=chain_resident=$v timeout 300 python $R/tools/t3q_quick.py x | tail -1
done
python - <<'PY'
import torch, time
x = torch.empty(10_600_000_000 // 8, dtype=torch.float64, device="cuda")
for _ in range(3): x.fill_(1.0)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): x.fill_(2.0)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
print("store-only fill of 10.6 GB: %.3f ms = %.0f GB/s" % (1e3 * dt, x.numel() * 8 / dt / 1e9))
PY
