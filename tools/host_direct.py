"""development aid: end-to-end host fill (gst_fill_dprobs into the layout's page-locked 'ep' array) with the kernel
writing straight into host memory (default) vs HBM + copy (GST_HOST_DIRECT=0).   python tools/host_direct.py [lite|full]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygsti_amd import modelpacks
from pygsti_amd.layout import HipCOPALayout
design = sys.argv[1] if len(sys.argv) > 1 else "full"
pack = modelpacks.smq2Q_XYICNOT
model = pack.target_model().depolarize(op_noise=0.01, spam_noise=0.01)
lay = HipCOPALayout(pack.create_gst_circuits(1024, lite=(design == "lite")), model, devices=[0])
plan = lay.atoms[0].plan()
plan.set_model(*lay.model_arrays(model)); plan.set_param_map(*lay.param_map(model))
nE, nP = lay.num_elements, model.num_params
J = lay.allocate_local_array("ep", "d"); pr = np.empty(nE)
pidx = np.arange(nP, dtype=np.int64)
plan.fill_dprobs(J, pidx, None, 1e-7, pr)
ts = []
for _ in range(3):
    t0 = time.perf_counter(); plan.fill_dprobs(J, pidx, None, 1e-7, pr); ts.append(time.perf_counter() - t0)
t = min(ts)
print("GST_HOST_DIRECT=%s pinned=%s: %.1f ms per fill, %.1f GB/s, checksum %.6e" % (
    os.environ.get("GST_HOST_DIRECT", "1"), lay.last_array_pinned, 1e3 * t, nE * nP * 8 / t / 1e9, float(np.abs(J).sum())))
