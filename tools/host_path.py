"""Development aid: the host-array variant of the Jacobian fill (what the pyGSTi adapter calls) -- PCIe-inclusive rate."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from pygsti_amd import modelpacks, _lib
from pygsti_amd.layout import HipCOPALayout

design = sys.argv[1] if len(sys.argv) > 1 else "lite"
pack = modelpacks.smq2Q_XYICNOT
circuits = pack.create_gst_circuits(1024, lite=(design == "lite"))
model = pack.target_model().depolarize(op_noise=0.01, spam_noise=0.01)
lay = HipCOPALayout(circuits, model, num_atoms=1, devices=[0], rank=0, size=1)
atom = lay.atoms[0]
plan = atom.plan()
plan.set_model(*lay.model_arrays(model))
plan.set_param_map(*lay.param_map(model))
nE, nP = atom.num_elements, model.num_params
out = np.empty((nE, nP))
out[:] = 0.0                    # touch the pages
for mode, nm in ((_lib.DERIV_FD, "fd"), (_lib.DERIV_ANALYTIC, "analytic")):
    ts = []
    for _ in range(4):
        t0 = time.perf_counter()
        plan.fill_dprobs(out, np.arange(nP), None, 1e-7, None, mode)
        ts.append(time.perf_counter() - t0)
    st = plan.stats()
    print("%s %s: nE=%d nP=%d (%.2f GB): host call %.1f ms (best of %s), device kernel %.2f ms -> %.3g el/s, %.1f GB/s over PCIe" % (
        design, nm, nE, nP, nE * nP * 8 / 1e9, 1e3 * min(ts), ["%.0f" % (1e3 * t) for t in ts], st["last_kernel_ms"], nE * nP / min(ts), nE * nP * 8 / 1e9 / min(ts)))
