#!/usr/bin/env python3
"""Record the reference's only published end-to-end workload for replay on the GPU box (build container only; the `.npz` is
data): test/performance/mpi_2D_scaling/run_me_with_mpirun.py:12-59 -- smq2Q_XYICNOT, `create_gst_experiment_design(64)`
(13,958 circuits, nE = 55,832, 1,616 parameters of the `full` parameterisation), data simulated from the target depolarized
by 0.01 / 0.01 (1,000 samples, seed 1234), chi^2 iterations over L = 1 ... 64 then the Poisson-picture logL stage, MINCLIP
1e-4, MapForwardSimulator(num_atoms=1).  The published figure for it is 3,113 s on one rank
(reference-8955f5d/timings.json:7, "1_1").

    PYTHONPATH=/tmp/pgref OMP_NUM_THREADS=4 python3 tests/golden/make_golden_fit2q.py [max_L]

No MPI here (mpi4py is absent): one process, comm=None.  `max_cache_size=0` of the original script is NOT kept -- prefix
caching changes the schedule, not one bit of any probability (a state is a pure function of its gate string), and the run
finishes sooner.  Written down, per `objective.dlsvec` call of the optimizer (simplerlm.py:663-678 turns each into J^T J and
J^T f), i.e. per LM iteration:
    the parameter vector; which stage; objective kind; sum(lsvec^2); and CHECKSUMS of the normal equations formed by numpy
    from the reference's own dlsvec / lsvec arrays -- diag(J^T J), J^T f, (J^T J) u for a fixed seeded unit vector u,
    trace and Frobenius norm -- (the 1,616 x 1,616 matrices themselves would be 21 MB per iterate);
    the wall-clock seconds the reference spent inside that dlsvec call.
Per stage: the circuits as integerised gate strings (checked equal to `create_gst_experiment_design(L)`), counts per
(circuit, outcome) in the design's circuit order with outcomes in the POVM's key order.
The whole run's wall-clock time in THIS container is recorded as context next to the published 3,113 s.
"""
import os
import sys
import time

import numpy as np

import pygsti
from pygsti.forwardsims import MapForwardSimulator
from pygsti.modelpacks import smq2Q_XYICNOT as std
from pygsti.objectivefns import objectivefns as O

HERE = os.path.dirname(os.path.abspath(__file__))
MAXL = int(sys.argv[1]) if len(sys.argv) > 1 else 64
OBJ = []
T_RUN0 = None

_orig_dlsvec = O.TimeIndependentMDCObjectiveFunction.dlsvec
RNG_U = np.random.default_rng(20260)


def _dlsvec(self, paramvec=None):
    t0 = time.perf_counter()
    jac = _orig_dlsvec(self, paramvec)
    t_ref = time.perf_counter() - t0
    n = self.nelements
    assert jac.shape[0] == n, "penalty rows are not part of this workload (cptp_penalty_factor 0)"
    assert self.firsts is None
    J = np.asarray(jac[:n])
    ls = np.array(self.lsvec(paramvec)[:n])
    nP = J.shape[1]
    if not hasattr(_dlsvec, "u"):
        u = RNG_U.standard_normal(nP); _dlsvec.u = u / np.linalg.norm(u)
    u = _dlsvec.u
    Ju = J @ u
    raw = self.raw_objfn
    kind = {"RawChi2Function": 0, "RawPoissonPicDeltaLogLFunction": 1}[type(raw).__name__]
    JTJ = J.T @ J                                    # 146 GFLOP at the last stage: seconds with threaded BLAS
    rec = dict(kind=kind, vec=self.model.to_vector().copy(), diag=np.diag(JTJ).copy(), jtf=J.T @ ls, jtju=J.T @ Ju,
               trace=float(np.trace(JTJ)), fro=float(np.linalg.norm(JTJ)), fsum=float(np.sum(ls * ls)),
               seconds=t_ref, t_since_start=time.perf_counter() - T_RUN0,
               layout=self.layout, circuits=self.circuits, counts=np.array(self.counts), totals=np.array(self.total_counts),
               clip=self.prob_clip_interval,
               reg=dict(mpcw=getattr(raw, "min_prob_clip_for_weighting", np.nan), min_p=getattr(raw, "min_p", np.nan),
                        radius=getattr(raw, "radius", np.nan), regtype=getattr(raw, "regtype", "")))
    del JTJ
    OBJ.append(rec)
    print("  dlsvec %3d: kind %d nE %6d  f=%.8g  reference %.1f s  (run %.0f s)" % (len(OBJ) - 1, kind, n, rec["fsum"], t_ref, rec["t_since_start"]), flush=True)
    return jac


O.TimeIndependentMDCObjectiveFunction.dlsvec = _dlsvec


def main():
    global T_RUN0
    mdl = std.target_model()
    exp_design = std.create_gst_experiment_design(MAXL)
    mdl_datagen = mdl.depolarize(op_noise=0.01, spam_noise=0.01)
    ds = pygsti.data.simulate_data(mdl_datagen, exp_design, 1000, seed=1234)
    MINCLIP = 1e-4
    chi2_builder = pygsti.objectivefns.ObjectiveFunctionBuilder(
        pygsti.objectivefns.Chi2Function, 'chi2', regularization={'min_prob_clip_for_weighting': MINCLIP},
        penalties={'cptp_penalty_factor': 0.0})
    mle_builder = pygsti.objectivefns.ObjectiveFunctionBuilder(
        pygsti.objectivefns.PoissonPicDeltaLogLFunction, 'logl', regularization={'min_prob_clip': MINCLIP, 'radius': MINCLIP})
    builders = pygsti.protocols.GSTObjFnBuilders([chi2_builder], [mle_builder])
    data = pygsti.protocols.ProtocolData(exp_design, ds)
    mdl.sim = MapForwardSimulator(num_atoms=1)
    gst = pygsti.protocols.GateSetTomography(mdl, objfn_builders=builders, optimizer=None, verbosity=2)
    T_RUN0 = time.perf_counter()
    results = gst.run(data, disable_checkpointing=True)
    t_run = time.perf_counter() - T_RUN0
    print("gst.run: %.1f s in this container (1 process)" % t_run)
    n_fit = len(OBJ)

    ops = list(mdl.operations.keys())
    lookup = {l: i for i, l in enumerate(ops)}
    povm_keys = list(mdl.povms['Mdefault'].keys())
    out = dict(max_L=np.int32(MAXL), nP=np.int32(mdl.num_params), D=np.int32(mdl.dim), u=_dlsvec.u,
               op_labels=np.array([str(l) for l in ops]), effect_labels=np.array([str(k) for k in povm_keys]),
               reference_run_seconds=np.float64(t_run), published_seconds_1_rank=np.float64(3113.0),
               derivative_eps=np.float64(mdl.sim.derivative_eps), n_obj=np.int32(n_fit),
               start_vec=std.target_model().to_vector())
    # stages: distinct circuit lists in order of first use
    stages = []
    for o in OBJ[:n_fit]:
        hit = [s for s, st in enumerate(stages) if st["layout"] is o["layout"]]
        if hit:
            o["stage"] = hit[0]
        else:
            o["stage"] = len(stages)
            stages.append(o)
    out["n_stages"] = np.int32(len(stages))
    Ls = exp_design.maxlengths
    for s, st in enumerate(stages):
        clist = list(st["circuits"])
        ptr = np.zeros(len(clist) + 1, np.int64); g = []
        for i, c in enumerate(clist):
            g.extend(lookup[l] for l in c.layertup); ptr[i + 1] = len(g)
        design = list(exp_design.circuit_lists[s])
        assert [str(c) for c in design] == [str(c) for c in clist], "stage %d is not the design's list for L=%d" % (s, Ls[s])
        lay = st["layout"]
        cnt = np.zeros((len(clist), len(povm_keys))); tot = np.zeros(len(clist))
        for i, c in enumerate(clist):
            idx = lay.indices(c)
            outs = lay.outcomes(c)
            for e, oc in zip(np.arange(idx.start, idx.stop) if isinstance(idx, slice) else idx, outs):
                cnt[i, povm_keys.index(oc[0])] = st["counts"][e]
                tot[i] = st["totals"][e]
        assert (cnt.sum(axis=1) == tot).all()
        pre = "s%d_" % s
        out.update({pre + "L": np.int32(Ls[s]), pre + "circ_ptr": ptr, pre + "circ_gates": np.array(g, np.int32),
                    pre + "counts": cnt.astype(np.int32), pre + "totals": tot.astype(np.int32), pre + "nE": np.int32(len(st["counts"]))})
    for k, o in enumerate(OBJ[:n_fit]):
        pre = "ob%d_" % k
        out.update({pre + "kind": np.int32(o["kind"]), pre + "stage": np.int32(o["stage"]), pre + "vec": o["vec"], pre + "diag": o["diag"],
                    pre + "jtf": o["jtf"], pre + "jtju": o["jtju"],
                    pre + "trace": np.float64(o["trace"]), pre + "fro": np.float64(o["fro"]), pre + "fsum": np.float64(o["fsum"]),
                    pre + "seconds": np.float64(o["seconds"]), pre + "t_since_start": np.float64(o["t_since_start"]),
                    pre + "mpcw": np.float64(o["reg"]["mpcw"]), pre + "min_p": np.float64(o["reg"]["min_p"]),
                    pre + "radius": np.float64(o["reg"]["radius"]),
                    pre + "clip_lo": np.float64(o["clip"][0] if o["clip"] is not None else -np.inf),
                    pre + "clip_hi": np.float64(o["clip"][1] if o["clip"] is not None else np.inf)})
    final = results.estimates[list(results.estimates.keys())[0]].models['final iteration estimate']
    out["final_vec"] = final.to_vector()
    name = "fit_smq2Q_XYICNOT_L%d_full.npz" % MAXL
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **out)
    print("wrote %s (%.1f KB): %d stages, %d dlsvec calls, reference dlsvec seconds total %.1f" % (
        path, os.path.getsize(path) / 1e3, len(stages), n_fit, sum(o["seconds"] for o in OBJ[:n_fit])))


if __name__ == "__main__":
    main()
