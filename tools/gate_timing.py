import sys, numpy as np
sys.path.insert(0, '.')
from pygsti_amd import modelpacks as MP
from pygsti_amd.layout import HipCOPALayout
pack = MP.smq2Q_XYICNOT; model = pack.target_model().depolarize(0.01, 0.01)
circuits = pack.create_gst_circuits(1024, lite=False)
lay = HipCOPALayout(circuits, model, 1); atom = lay.atoms[0]; plan = atom.plan()
plan.set_model(*lay.model_arrays(model)); plan.set_param_map(*lay.param_map(model))
nE = atom.num_elements
d_out = plan.device_malloc(nE * 256 * 8); d_pr = plan.device_malloc(nE * 8)
for name, lo, hi in [('rho', 0, 16), ('eff', 16, 80)] + [('gate%d' % g, 80 + 256 * g, 80 + 256 * (g + 1)) for g in range(6)]:
    idx = np.arange(lo, hi)
    for rep in range(2):
        plan.fill_dprobs_dev(d_out, 256, idx, None, 1e-7, d_pr); plan.sync()
    print(name, 'kernel_ms %.3f' % plan.stats()['last_kernel_ms'], 'waves', (hi - lo + 63) // 64)
# all-clean control: pretend gate 0's parameters belong to no object of this atom
k, o, e = lay.param_map(model)
k = k.copy(); k[80:336] = -1
plan.set_param_map(k, o, e)
idx = np.arange(80, 336)
for rep in range(2):
    plan.fill_dprobs_dev(d_out, 256, idx, None, 1e-7, d_pr); plan.sync()
print('none', 'kernel_ms %.3f' % plan.stats()['last_kernel_ms'])
