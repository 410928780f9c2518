"""gst_lm_step_dev: one Levenberg-Marquardt evaluation per call, replayed as ONE HIP graph launch on launch-bound plans
(BASELINE configs[1], smq1Q_XYI L <= 128) -- against the four-call composition `Plan.lsq_step` (same kernels: same bits) over a
sequence of changing models, with the graph dropped and rebuilt when the destination changes."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _case(L=128):
    from pygsti_amd import modelpacks
    from pygsti_amd.layout import HipCOPALayout
    pack = modelpacks.smq1Q_XYI
    model = pack.target_model().depolarize(op_noise=0.01, spam_noise=0.01)
    circuits = pack.create_gst_circuits(L)
    lay = HipCOPALayout(circuits, model, num_atoms=1)
    plan = lay.atoms[0].plan()
    plan.set_model(*lay.model_arrays(model)); plan.set_param_map(*lay.param_map(model))
    return pack, model, lay, plan


@pytest.mark.parametrize("objective", ["chi2", "logl"])
def test_graph_replay_equals_the_four_call_composition(objective):
    pack, model, lay, plan = _case()
    nE, nP = lay.global_num_elements, model.num_params
    rng = np.random.default_rng(5)
    p0 = plan.fill_probs()
    counts = rng.binomial(1000, np.clip(p0, 0, 1)).astype(np.float64); totals = np.full(nE, 1000.0)
    bufs = {k: plan.device_malloc(n * 8) for k, n in (("J", nE * nP), ("p", nE), ("ls", nE), ("w", nE), ("jtj", nP * nP), ("jtf", nP), ("c", nE), ("N", nE))}
    plan.memcpy_h2d(bufs["c"], counts); plan.memcpy_h2d(bufs["N"], totals)
    v0 = model.to_vector()
    for it in range(6):
        model.from_vector(v0 + 1e-3 * it * np.sin(np.arange(nP) + it))
        G, R, E = lay.model_arrays(model)
        plan.set_model(G, R, E)
        tot = plan.lm_step_dev(nP, bufs["c"], bufs["N"], bufs["J"], nP, bufs["p"], bufs["ls"], bufs["w"], bufs["jtj"], bufs["jtf"], objective)
        jtj = np.empty((nP, nP)); plan.memcpy_d2h(jtj, bufs["jtj"])
        jtf = np.empty(nP); plan.memcpy_d2h(jtf, bufs["jtf"])
        J = np.empty((nE, nP)); plan.memcpy_d2h(J, bufs["J"])
        plan.set_model(G, R, E)
        tot2, jtj2, jtf2 = plan.lsq_step(nP, counts, totals, objective)
        J2 = plan.fill_dprobs(eps=1e-7)
        assert np.array_equal(J, J2), "iteration %d: Jacobian" % it
        assert np.array_equal(jtj, jtj2) and np.array_equal(jtf, jtf2), "iteration %d: normal equations" % it
        assert tot == tot2, (it, tot, tot2)
    # another destination: the graph is dropped, the ordinary path serves the call, a new graph forms
    d2 = plan.device_malloc(nP * nP * 8)
    for _ in range(3):
        tot3 = plan.lm_step_dev(nP, bufs["c"], bufs["N"], bufs["J"], nP, bufs["p"], bufs["ls"], bufs["w"], d2, bufs["jtf"], objective)
        jtj3 = np.empty((nP, nP)); plan.memcpy_d2h(jtj3, d2)
        assert np.array_equal(jtj3, jtj) and tot3 == tot
    plan.device_free(d2)
    for d in bufs.values():
        plan.device_free(d)


def test_lm_step_is_refused_where_it_does_not_apply():
    from pygsti_amd import _lib
    pack, model, lay, plan = _case(4)
    nE, nP = lay.global_num_elements, model.num_params
    d = plan.device_malloc(max(nE * nP, nP * nP) * 8)
    with pytest.raises(ValueError):
        plan.lm_step_dev(nP + 1, d, d, d, nP + 1, d, d, d, d, d)           # not the parameter map's size
    plan.set_derivs(nP, [])
    plan.device_free(d)
