"""development aid: the streamed two-circuit contraction (default) against the gate-by-gate one (GST_ANALYTIC_STREAM=0) and
against the FD Jacobian, on the whole bench workload (every row, every column).   python tools/check_stream.py [lite|full]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygsti_amd import modelpacks, _lib
from pygsti_amd.layout import HipCOPALayout
design = sys.argv[1] if len(sys.argv) > 1 else "full"
pack = modelpacks.smq2Q_XYICNOT
model = pack.target_model().depolarize(op_noise=0.01, spam_noise=0.01)
circuits = pack.create_gst_circuits(1024, lite=(design == "lite"))
outs = {}
for tag, env in (("stream", "1"), ("gatewise", "0")):
    os.environ["GST_ANALYTIC_STREAM"] = env
    lay = HipCOPALayout(circuits, model, devices=[0])
    plan = lay.atoms[0].plan()
    plan.set_model(*lay.model_arrays(model)); plan.set_param_map(*lay.param_map(model))
    nE, nP = lay.num_elements, model.num_params
    d_J = plan.device_malloc(nE * nP * 8)
    plan.fill_dprobs_dev(d_J, nP, np.arange(nP), None, 1e-7, None, _lib.DERIV_ANALYTIC); plan.sync()
    J = np.empty((nE, nP)); plan.memcpy_d2h(J, d_J)
    outs[tag] = J
    if tag == "gatewise":
        plan.fill_dprobs_dev(d_J, nP, np.arange(nP), None, 1e-7, None, _lib.DERIV_FD); plan.sync()
        Jfd = np.empty((nE, nP)); plan.memcpy_d2h(Jfd, d_J)
    plan.device_free(d_J); del plan, lay
a, b = outs["stream"], outs["gatewise"]
print("max |stream - gatewise| = %.3e (max |J| = %.3e); nonfinite: %d" % (np.abs(a - b).max(), np.abs(b).max(), int((~np.isfinite(a)).sum())))
print("max |stream - FD| = %.3e, max |gatewise - FD| = %.3e" % (np.abs(a - Jfd).max(), np.abs(b - Jfd).max()))
print("rows whose stream result differs from gatewise by > 1e-9: %d of %d" % (int((np.abs(a - b).max(axis=1) > 1e-9).sum()), a.shape[0]))
