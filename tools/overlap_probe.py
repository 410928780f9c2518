"""development aid: does running a rank's share as 2 (or 4) atoms on their own streams hide the base pass and the
FD tail?  Emulates rank 0 of 8.   python tools/overlap_test.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygsti_amd import modelpacks, _lib
from pygsti_amd.layout import HipCOPALayout

pack = modelpacks.smq2Q_XYICNOT
model = pack.target_model().depolarize(op_noise=0.01, spam_noise=0.01)
circuits = pack.create_gst_circuits(1024, lite=False)
nP = model.num_params
pidx = np.arange(nP, dtype=np.int64)
for per_rank in (1, 2, 4):
    layout = HipCOPALayout(circuits, model, num_atoms=8 * per_rank, devices=[0], rank=0, size=8)
    gates, rhos, effects = layout.model_arrays(model)
    atoms = layout.atoms
    plans = [a.plan() for a in atoms]
    bufs = []
    for a, p in zip(atoms, plans):
        p.set_model(gates, rhos, effects); p.set_param_map(*layout.param_map(model))
        bufs.append((p.device_malloc(a.num_elements * nP * 8), p.device_malloc(a.num_elements * 8)))
    def step():
        for p, (dJ, dp) in zip(plans, bufs):
            p.set_model(gates, rhos, effects)
            p.fill_dprobs_dev(dJ, nP, pidx, None, 1e-7, dp, _lib.DERIV_FD)
    for _ in range(3):
        step()
    for p in plans: p.sync()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        step()
        for p in plans: p.sync()
    t = (time.perf_counter() - t0) / n
    nE = sum(a.num_elements for a in atoms)
    print("atoms per rank %d: %d elements, %.3f ms per step, %.4g el/s" % (per_rank, nE, 1e3 * t, nE * nP / t))
    for p, (dJ, dp) in zip(plans, bufs):
        p.device_free(dJ); p.device_free(dp)
