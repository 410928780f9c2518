#!/bin/bash
python -c "
import json, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tools')
import bench_configs as B
d = B.three_q()
print({k: round(v, 3) for k, v in d.items() if k in ('probs_ms', 'dprobs_fd_ms', 'dprobs_analytic_same_block_ms', 'dprobs_analytic_full_ms')})"
