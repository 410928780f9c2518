"""development aid: the exact (analytic) Jacobian of the CPTPLND model, a few fills (for rocprofv3 --kernel-trace --stats)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pygsti_amd import _lib, lindblad as LBM
pack, model, circuits, layout = bench.build_workload("full", 1024, 1, 0, 0, 0, 0, "strong")
plan = layout.atoms[0].plan()
lmodel = LBM.LindbladModel.from_target(pack.target_model(), layout.model_gate_labels, layout.effect_labels, "CPTPLND")
theta = 0.003 * np.random.default_rng(9).standard_normal(lmodel.num_params)
nE, nP = layout.num_elements, lmodel.num_params
plan.set_lindblad(lmodel); plan.set_lindblad_params(theta)
d = plan.device_malloc(nE * nP * 8); dp = plan.device_malloc(nE * 8)
pidx = np.arange(nP, dtype=np.int64)
for _ in range(4):
    t0 = time.perf_counter()
    plan.set_lindblad_params(theta)
    plan.fill_dprobs_dev(d, nP, pidx, None, 1e-7, dp, _lib.DERIV_ANALYTIC); plan.sync()
    print("analytic CPTPLND Jacobian: %.2f ms" % (1e3 * (time.perf_counter() - t0)))
