"""Development aid: how fast does ONE wavefront walk when it has a SIMD to itself?  FD columns of 64 parameters of one
gate on a 1/8 atom of the 2Q design: 173 (task, wavefront) pairs on 1024 SIMDs."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from pygsti_amd import modelpacks, _lib
from pygsti_amd.layout import HipCOPALayout

pack = modelpacks.smq2Q_XYICNOT
circuits = pack.create_gst_circuits(1024, lite=False)
model = pack.target_model().depolarize(op_noise=0.01, spam_noise=0.01)
lay = HipCOPALayout(circuits, model, num_atoms=8, devices=[0], rank=0, size=8)
atom = lay.atoms[0]
plan = atom.plan()
plan.set_model(*lay.model_arrays(model))
plan.set_param_map(*lay.param_map(model))
nE = atom.num_elements
st = plan.stats()
for ncol in (64, 128, 256, 1024):
    pidx = np.arange(80, 80 + ncol, dtype=np.int64)       # gate parameters start at 80
    d_out = plan.device_malloc(nE * ncol * 8)
    d_p = plan.device_malloc(nE * 8)
    for _ in range(3):
        plan.fill_dprobs_dev(d_out, ncol, pidx, None, 1e-7, d_p, _lib.DERIV_FD)
        plan.sync()
    ks = []
    for _ in range(5):
        plan.fill_dprobs_dev(d_out, ncol, pidx, None, 1e-7, d_p, _lib.DERIV_FD)
        plan.sync()
        ks.append(plan.stats()["last_kernel_ms"])
    print("columns %4d: kernel %.3f ms (tasks %d, applies per pass %d)" % (ncol, float(np.median(ks)), st["n_tasks"], st["applies_per_pass"]))
