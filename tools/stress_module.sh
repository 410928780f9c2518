#!/bin/bash
# Repeat GPU test modules N times in fresh processes, keeping the full output (stderr included) of every run that fails.
M=${M:-"tests/test_gpu_adapter_modes.py tests/test_gpu_analytic.py"}
N=${N:-12}
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out/stress
rm -f gpurun_out/stress/*
bad=0
for i in $(seq 1 $N); do
  timeout 300 python -X faulthandler -m pytest $M -m gpu -q -x --timeout 200 -p no:cacheprovider > gpurun_out/stress/run_$i.txt 2>&1
  rc=$?
  if [ $rc != 0 ]; then bad=$((bad+1)); echo "run $i: rc=$rc"; tail -40 gpurun_out/stress/run_$i.txt; else rm -f gpurun_out/stress/run_$i.txt; fi
done
echo "stress: $N runs, $bad bad" | tee gpurun_out/stress/summary.txt
