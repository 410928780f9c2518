"""Development aid: where the CPTPLND finite-difference Jacobian's time goes -- columns of one gate, of the preparation,
of the POVM, all."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pygsti_amd import modelpacks, lindblad as LBM, _lib
from pygsti_amd.layout import HipCOPALayout
design = sys.argv[1] if len(sys.argv) > 1 else "full"
pack = modelpacks.smq2Q_XYICNOT
circuits = pack.create_gst_circuits(1024, lite=(design == "lite"))
model = pack.target_model().depolarize(op_noise=0.01, spam_noise=0.01)
layout = HipCOPALayout(circuits, model, num_atoms=1, devices=[0])
plan = layout.atoms[0].plan()
lm = LBM.LindbladModel.from_target(pack.target_model(), layout.model_gate_labels, layout.effect_labels, "CPTPLND")
theta = 0.003 * np.random.default_rng(9).standard_normal(lm.num_params)
plan.set_lindblad(lm); plan.set_lindblad_params(theta)
nE = layout.atoms[0].num_elements
subsets = {"rho": np.arange(0, 240), "povm": np.arange(240, 480)}
for m in lm.members:
    if m.kind == 0:
        subsets["gate%d(%s)" % (m.obj, layout.model_gate_labels[m.obj])] = np.arange(m.param0, m.param0 + 240)
subsets["all"] = np.arange(lm.num_params)
d = plan.device_malloc(nE * lm.num_params * 8)
dp = plan.device_malloc(nE * 8)
for name, cols in subsets.items():
    n = len(cols)
    plan.fill_dprobs_dev(d, n, cols, None, 1e-7, dp, _lib.DERIV_FD); plan.sync()
    t0 = time.perf_counter()
    for _ in range(2):
        plan.fill_dprobs_dev(d, n, cols, None, 1e-7, dp, _lib.DERIV_FD)
    plan.sync()
    dt = (time.perf_counter() - t0) / 2
    print("%-22s %4d cols  %8.2f ms  kernel %.2f ms  %.3g el/s" % (name, n, 1e3 * dt, plan.stats()["last_kernel_ms"], nE * n / dt))
