#!/bin/bash
R=$PWD; OUT=$R/gpurun_out/r06h; rm -rf $OUT; mkdir -p $OUT
cat > /tmp/lm1q.py <<'PY'
import sys, json, time
import os; sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.')); sys.path.insert(0, os.path.join(os.environ.get('GRAFT_REPO_ROOT', '.'), 'tools'))
import numpy as np
from pygsti_amd import _lib, modelpacks
from pygsti_amd.layout import HipCOPALayout
pack = modelpacks.smq1Q_XYI
model = pack.target_model().depolarize(op_noise=0.01, spam_noise=0.01)
circuits = pack.create_gst_circuits(128)
layout = HipCOPALayout(circuits, model, num_atoms=1, devices=[0])
plan = layout.atoms[0].plan()
G_, R_, E_ = layout.model_arrays(model)
plan.set_model(G_, R_, E_); plan.set_param_map(*layout.param_map(model))
nE, nP = layout.num_elements, model.num_params
bufs = [plan.device_malloc(n * 8) for n in (nE * nP, nE, nE, nE, nE, nE, nP * nP, nP)]
d_J, d_p, d_c, d_N, d_ls, d_w, d_jtj, d_jtf = bufs
plan.memcpy_h2d(d_c, np.full(nE, 500.0)); plan.memcpy_h2d(d_N, np.full(nE, 1000.0))
pidx = np.arange(nP, dtype=np.int64)
def one():
    plan.set_model(G_, R_, E_)
    plan.lm_step_dev(nP, d_c, d_N, d_J, nP, d_p, d_ls, d_w, d_jtj, d_jtf, "logl")
def four():
    plan.set_model(G_, R_, E_)
    plan.fill_dprobs_dev(d_J, nP, pidx, None, 1e-7, d_p, _lib.DERIV_FD)
    plan.objective_rows_dev("logl", d_p, d_c, d_N, nE, d_ls, d_w)
    plan.fill_normal_eqs_dev(d_J, nE, nP, nP, d_w, d_ls, d_jtj, d_jtf)
    plan.sync()
def fd_only():
    plan.set_model(G_, R_, E_); plan.fill_dprobs_dev(d_J, nP, pidx, None, 1e-7, d_p, _lib.DERIV_FD); plan.sync()
def lat(fn, reps=300):
    fn(); fn(); fn()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    return 1e6 * (time.perf_counter() - t0) / reps
print(json.dumps({"one_call_us": lat(one), "replays": plan.stats()["lm_graph_replays"], "four_calls_us": lat(four), "fd_only_blocking_us": lat(fd_only)}))
PY
timeout 600 python -X faulthandler -m pytest tests/test_gpu_lm_graph.py tests/test_gpu_jtj.py tests/test_fit_replay2q.py tests/test_fit_replay.py -m gpu -q -x --timeout 300 -p no:cacheprovider 2>&1 | tail -3
timeout 120 python /tmp/lm1q.py 2>&1 | tail -2
GST_TEST_FORCE=lm_graph=0 timeout 120 python /tmp/lm1q.py 2>&1 | tail -2
cd /tmp; export TMPDIR=/tmp
PYTHONPATH=$R:$R/tools timeout 200 rocprofv3 --kernel-trace --stats -f csv -d $OUT/st -o s -- python /tmp/lm1q.py > $OUT/st.log 2>&1
cut -d, -f1-4 $OUT/st/s_kernel_stats.csv | head -12
