#!/bin/bash
# 1Q L<=128 fills, this tree against the round-2 tree staged under gpurun_tmp_r02/ (same box, interleaved)
for rep in 1 2 3; do
  for T in . gpurun_tmp_r02; do
    ( cd $T && timeout 120 python - <<'PY'
import json, sys, os
sys.path.insert(0, os.getcwd())
import importlib.util
spec = importlib.util.spec_from_file_location("bc", "tools/bench_configs.py"); bc = importlib.util.module_from_spec(spec); spec.loader.exec_module(bc)
r = bc.one_q()
print(os.path.basename(os.getcwd()) or ".", {k: round(v, 1) for k, v in r["blocking_host_fill_us"].items()}, "async", round(r["probs_us"], 1), round(r["dprobs_fd_us"], 1), round(r["dprobs_analytic_us"], 1))
PY
    )
  done
done
