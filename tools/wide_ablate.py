import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygsti_amd import modelpacks as MP, _lib
from pygsti_amd.layout import HipCOPALayout
pack = MP.smq2Q_XYICNOT
model = pack.target_model().depolarize(0.01, 0.01)
layout = HipCOPALayout(pack.create_gst_circuits(1024, lite=False), model, num_atoms=1, devices=[0], rank=0, size=1)
plan = layout.atoms[0].plan()
G, R, E = layout.model_arrays(model)
plan.set_model(G, R, E); plan.set_param_map(*layout.param_map(model))
nE, nP = layout.num_elements, model.num_params
d_p = plan.device_malloc(nE * 8); d_J = plan.device_malloc(nE * nP * 8)
pidx = np.arange(nP, dtype=np.int64)
for dbg in (0, 4, 8, 16, 12, 20, 24, 28):
    os.environ["GST_WIDE_DEBUG"] = str(dbg)
    for _ in range(2):
        plan.fill_dprobs_dev(d_J, nP, pidx, None, 1e-7, d_p, _lib.DERIV_ANALYTIC)
    plan.sync()
    t0 = time.perf_counter()
    for _ in range(5):
        plan.fill_dprobs_dev(d_J, nP, pidx, None, 1e-7, d_p, _lib.DERIV_ANALYTIC)
    plan.sync()
    print("debug=%2d (4: no stores, 8: no MFMAs, 16: no global loads in the loop): %.3f ms per fill, kernel_ms %.3f" % (dbg, 1e3 * (time.perf_counter() - t0) / 5, plan.stats()["last_kernel_ms"]))
