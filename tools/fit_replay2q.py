"""Replay of a recorded end-to-end GST fit of the reference on the device (tests/golden/make_golden_fit2q.py wrote the record in
the build container: the reference's only published workload, test/performance/mpi_2D_scaling/run_me_with_mpirun.py:12-59 --
smq2Q_XYICNOT, `create_gst_experiment_design(64)`, chi^2 stages L = 1 ... 64 then the Poisson-picture logL stage).

For every `objective.dlsvec` call of the recorded run -- one per Levenberg-Marquardt iteration, simplerlm.py:663-678 -- the
recorded parameter vector goes through what the drop-in does for it: set_model -> finite-difference Jacobian (bit-identical to
the Map path) -> objective rows -> J_s^T J_s and J_s^T lsvec, all on the device (`Plan.lsq_step`), and the results are
compared with the checksums numpy formed from the reference's own dlsvec / lsvec arrays: diag(J^T J), J^T f, (J^T J) u for a
seeded unit vector u, trace, Frobenius norm, sum(lsvec^2).  Layouts here order their rows their own way; every checked
quantity is a sum over rows.

Used by tests/test_fit_replay2q.py (-m gpu) and by bench.py's `gst_fit_2Q_L64` leg."""
import os
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(name):
    return dict(np.load(os.path.join(ROOT, "tests", "golden", name + ".npz")))


def replay(name="fit_smq2Q_XYICNOT_L64_full", device=-1, mode="fd", max_calls=None, log=None):
    from pygsti_amd import _lib, modelpacks
    from pygsti_amd.layout import HipCOPALayout
    fx = load(name)
    pack = modelpacks.smq2Q_XYICNOT
    model = pack.target_model()
    nP = int(fx["nP"])
    assert model.num_params == nP and np.allclose(model.to_vector(), fx["start_vec"], rtol=0, atol=1e-15), "parameter order differs from the reference's"
    ops = [str(l) for l in fx["op_labels"]]
    assert ops == [str(l) for l in pack.gate_labels], (ops, pack.gate_labels)
    eff = [str(l) for l in fx["effect_labels"]]
    n_obj = int(fx["n_obj"]) if max_calls is None else min(int(fx["n_obj"]), int(max_calls))
    dmode = _lib.DERIV_ANALYTIC if mode == "analytic" else _lib.DERIV_FD
    stages = {}

    def stage(s):           # noqa: C901
        if s not in stages:
            t0 = time.perf_counter()
            ptr, g = fx["s%d_circ_ptr" % s], fx["s%d_circ_gates" % s]
            circuits = [tuple(ops[k] for k in g[ptr[i]:ptr[i + 1]]) for i in range(len(ptr) - 1)]
            design = pack.create_gst_circuits(int(fx["s%d_L" % s]), lite=True)
            assert circuits == [tuple(c) for c in design], "stage %d is not this build's design for L = %d" % (s, int(fx["s%d_L" % s]))
            lay = HipCOPALayout(circuits, model, num_atoms=1, devices=[device])
            atom = lay.atoms[0]
            plan = atom.plan()
            plan.set_param_map(*lay.param_map(model))
            # counts / totals in THIS layout's element order: element (circuit i, outcome j) at circuit_offset[i] + j, outcomes
            # in model.effect_labels order
            order = [eff.index(str(l).split("_", 1)[-1]) for l in model.effect_labels]
            nE = lay.global_num_elements
            counts, totals = np.empty(nE), np.empty(nE)
            c_in, t_in = fx["s%d_counts" % s].astype(np.float64), fx["s%d_totals" % s].astype(np.float64)
            for i in range(len(circuits)):
                sl = lay.indices_for_index(i)
                counts[sl] = c_in[i, order]; totals[sl] = t_in[i]
            assert nE == int(fx["s%d_nE" % s])
            stages[s] = (lay, plan, counts, totals, time.perf_counter() - t0)
        return stages[s]

    u = fx["u"]
    worst = {"diag": 0.0, "jtf": 0.0, "jtju": 0.0, "trace": 0.0, "fro": 0.0, "fsum": 0.0}
    per_call, setup_s = [], 0.0
    ref_seconds = 0.0
    for k in range(n_obj):
        s, kind = int(fx["ob%d_stage" % k]), int(fx["ob%d_kind" % k])
        new = s not in stages
        lay, plan, counts, totals, t_setup = stage(s)
        if new:
            setup_s += t_setup
        model.from_vector(fx["ob%d_vec" % k])
        mpc = float(fx["ob%d_mpcw" % k]) if kind == 0 else float(fx["ob%d_min_p" % k])
        rad = 1e-4 if kind == 0 else float(fx["ob%d_radius" % k])
        clip = (float(fx["ob%d_clip_lo" % k]), float(fx["ob%d_clip_hi" % k]))
        plan.sync()
        t0 = time.perf_counter()
        plan.set_model(*lay.model_arrays(model))
        total, jtj, jtf = plan.lsq_step(nP, counts, totals, "chi2" if kind == 0 else "logl", float(fx["derivative_eps"]), dmode,
                                        mpc, rad, clip)
        dt = time.perf_counter() - t0
        per_call.append(dt)
        ref_seconds += float(fx["ob%d_seconds" % k])
        scale = float(np.abs(fx["ob%d_diag" % k]).max())
        fs = float(fx["ob%d_fsum" % k])
        dev = {"diag": np.abs(np.diag(jtj) - fx["ob%d_diag" % k]).max() / scale,
               "jtju": np.abs(jtj @ u - fx["ob%d_jtju" % k]).max() / scale,
               # J_s^T lsvec -> 0 at an optimum: its error scale is that of its cancelling summands, |J_s| |lsvec|
               "jtf": np.abs(jtf - fx["ob%d_jtf" % k]).max() / np.sqrt(scale * fs),
               "trace": abs(np.trace(jtj) - float(fx["ob%d_trace" % k])) / float(fx["ob%d_trace" % k]),
               "fro": abs(np.linalg.norm(jtj) - float(fx["ob%d_fro" % k])) / float(fx["ob%d_fro" % k]),
               "fsum": abs(total - fs) / fs}
        for q, v in dev.items():
            worst[q] = max(worst[q], float(v))
        if log:
            log("  dlsvec %3d stage %d kind %d: %.2f ms (reference %.1f s)  worst dev %.2e" % (k, s, kind, 1e3 * dt, float(fx["ob%d_seconds" % k]), max(dev.values())))
    out = {"record": name, "iterations": n_obj, "stages": int(fx["n_stages"]), "n_params": nP,
           "nE_last_stage": int(fx["s%d_nE" % (int(fx["n_stages"]) - 1)]), "derivative": mode,
           "device_ms_sum": 1e3 * float(np.sum(per_call)), "device_ms_max": 1e3 * float(np.max(per_call)),
           "layout_and_plan_s": setup_s,
           "reference_dlsvec_seconds_same_calls": ref_seconds,
           "reference_run_seconds_build_container": float(fx["reference_run_seconds"]),
           "reference_published_seconds_1_rank": float(fx["published_seconds_1_rank"]),
           "worst_relative_deviation": worst, "worst": max(worst.values()),
           "what": "every LM iteration of the recorded reference fit (vector -> FD Jacobian -> objective rows -> JtJ, Jtf on the "
                   "device, incl. the nP^2 download) vs checksums of the reference's own dlsvec arrays"}
    for lay, plan, *_ in stages.values():
        plan.close()
    return out


if __name__ == "__main__":
    import json
    import sys
    sys.path.insert(0, ROOT)
    print(json.dumps(replay(sys.argv[1] if len(sys.argv) > 1 else "fit_smq2Q_XYICNOT_L64_full", log=print), indent=1))
