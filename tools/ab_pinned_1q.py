"""development aid: blocking 1Q fills into pageable against page-locked host arrays (tools/bench_configs.one_q's loop)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygsti_amd import _lib, modelpacks
from pygsti_amd.layout import HipCOPALayout
pack = modelpacks.smq1Q_XYI
model = pack.target_model().depolarize(op_noise=0.01, spam_noise=0.01)
layout = HipCOPALayout(pack.create_gst_circuits(128), model, num_atoms=1, devices=[0])
plan = layout.atoms[0].plan()
plan.set_model(*layout.model_arrays(model)); plan.set_param_map(*layout.param_map(model))
nE, nP = layout.num_elements, model.num_params
pidx = np.arange(nP, dtype=np.int64)


def lat(fn, reps=300):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return 1e6 * (time.perf_counter() - t0) / reps


for pinned in (False, True, False, True):
    J = np.empty((nE, nP)); pr = np.empty(nE)
    if pinned:
        assert _lib.pin_host_array(J) and _lib.pin_host_array(pr)
    print("pinned" if pinned else "pageable", "probs %.1f us  FD %.1f us  analytic %.1f us" % (
        lat(lambda: plan.fill_probs(pr)), lat(lambda: plan.fill_dprobs(J, pidx, None, 1e-7, pr, _lib.DERIV_FD)),
        lat(lambda: plan.fill_dprobs(J, pidx, None, 1e-7, pr, _lib.DERIV_ANALYTIC))))
    if pinned:
        _lib.unpin_host_array(J); _lib.unpin_host_array(pr)
