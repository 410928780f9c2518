#!/usr/bin/env python3
"""Record one END-TO-END GST fit of the reference for replay on the GPU box (build container only; the `.npz` is data).

    PYTHONPATH=/tmp/pgref OMP_NUM_THREADS=1 python3 tests/golden/make_golden_fit.py

The reference pins this hot path with whole GST runs (test/unit/protocols/test_gst.py:243-292,
test/unit/objects/test_forwardsim.py:351-378: `GateSetTomography(target_model("full TP"), 'stdgaugeopt').run(data,
simulator=...)`, then 2*delta-logL of the estimate).  pyGSTi cannot travel to the GPU box, so the run happens HERE --
smq1Q_XYI, L <= 8 (448 circuits), 1,000 binomially sampled shots per circuit, full-TP model (43 parameters) -- through a
RECORDING MapForwardSimulator: the reference's own Cython path computes every number, and every call the optimizer makes
through the `_bulk_fill_*_atom` seams (mapforwardsim.py:372-391) is written down:

  per stage (L = 1, 2, 4, 8; one layout atom each): the atom's prefix table and effect CSR, the data set's counts and
      totals in layout element order, the objective's regularisation constants;
  per Jacobian request (every `_bulk_fill_dprobs_atom` call, in order): the parameter vector, the dense model the
      members had at that moment (atom order), the probabilities and the FULL finite-difference Jacobian the reference
      returned;
  per `dlsvec` call of the objective (what simplish_leastsq / simplerlm.py:677-678 turns into J^T J and J^T f): objective
      kind, J_s^T J_s, J_s^T lsvec and sum(lsvec^2) computed by numpy from the reference's own dlsvec / lsvec arrays;
  the final estimate: its parameter vector and dense model, `two_delta_logl(model, dataset)` and its arguments.

The full-TP parameter map (one parameter per dense element, complement effect) is stored as in `3q_explicit_TP`.
"""
import os

import numpy as np

import pygsti  # noqa: F401
from pygsti.data import simulate_data
from pygsti.forwardsims import MapForwardSimulator
from pygsti.modelpacks import smq1Q_XYI
from pygsti.objectivefns import objectivefns as O
from pygsti.protocols import gst, ProtocolData
from pygsti.tools import two_delta_logl

HERE = os.path.dirname(os.path.abspath(__file__))
CALLS = []          # simulator-level records
OBJ = []            # objective-level records


def dense_of(model, atom):
    D = model.dim
    g = np.array([model._circuit_layer_operator(l, 'op').to_dense('minimal') for l in atom.op_labels]).real
    r = np.array([model._circuit_layer_operator(l, 'prep').to_dense('minimal') for l in atom.rho_labels]).real
    e = np.array([model._circuit_layer_operator(l, 'povm').to_dense('minimal') for l in atom._fit_eff_labels]).real
    return (np.ascontiguousarray(g.reshape(len(atom.op_labels), D, D), dtype=np.float64),
            np.ascontiguousarray(r.reshape(len(atom.rho_labels), D), dtype=np.float64),
            np.ascontiguousarray(e.reshape(len(atom._fit_eff_labels), D), dtype=np.float64))


class RecordingMapForwardSimulator(MapForwardSimulator):
    def _note_atom(self, atom):
        if not hasattr(atom, "_fit_eff_labels"):
            atom._fit_eff_labels = list(atom.full_effect_labels)       # a set: freeze its iteration order once

    def _bulk_fill_probs_atom(self, array_to_fill, layout_atom, resource_alloc):
        super()._bulk_fill_probs_atom(array_to_fill, layout_atom, resource_alloc)
        self._note_atom(layout_atom)
        CALLS.append(dict(what="probs", atom=layout_atom, vec=self.model.to_vector().copy(), out=np.array(array_to_fill)))

    def _bulk_fill_dprobs_atom(self, array_to_fill, dest_param_slice, layout_atom, param_slice, resource_alloc):
        super()._bulk_fill_dprobs_atom(array_to_fill, dest_param_slice, layout_atom, param_slice, resource_alloc)
        self._note_atom(layout_atom)
        nP = self.model.num_params
        assert dest_param_slice is None or (dest_param_slice == slice(0, nP)), dest_param_slice
        assert param_slice is None or param_slice == slice(0, nP), param_slice
        CALLS.append(dict(what="dprobs", atom=layout_atom, vec=self.model.to_vector().copy(), out=np.array(array_to_fill),
                          dense=dense_of(self.model, layout_atom)))


_orig_dlsvec = O.TimeIndependentMDCObjectiveFunction.dlsvec


def _dlsvec(self, paramvec=None):
    jac = _orig_dlsvec(self, paramvec)
    n = self.nelements
    assert jac.shape[0] == n, "penalty rows are not part of this fixture"
    J = np.array(jac[:n])
    ls = np.array(self.lsvec(paramvec)[:n])
    raw = self.raw_objfn
    kind = {"RawChi2Function": 0, "RawPoissonPicDeltaLogLFunction": 1}[type(raw).__name__]
    reg = dict(min_prob_clip_for_weighting=getattr(raw, "min_prob_clip_for_weighting", np.nan),
               min_p=getattr(raw, "min_p", np.nan), radius=getattr(raw, "radius", np.nan),
               regtype=getattr(raw, "regtype", ""))
    assert self.firsts is None
    OBJ.append(dict(kind=kind, n_calls_before=len(CALLS), vec=self.model.to_vector().copy(), jtj=J.T @ J, jtf=J.T @ ls,
                    fsum=float(np.sum(ls * ls)), lsvec=ls, counts=np.array(self.counts), totals=np.array(self.total_counts),
                    clip=self.prob_clip_interval, reg=reg, layout=self.layout))
    return jac


O.TimeIndependentMDCObjectiveFunction.dlsvec = _dlsvec


def table_arrays(atom):
    op_lookup = {l: i for i, l in enumerate(atom.op_labels)}
    rho_lookup = {l: i for i, l in enumerate(atom.rho_labels)}
    contents = atom.table.contents
    R = len(contents)
    t_dest = np.empty(R, np.int32); t_start = np.empty(R, np.int32); t_cache = np.empty(R, np.int32); t_rho = -np.ones(R, np.int32)
    row_ptr = np.zeros(R + 1, np.int64); gidx = []
    for k, (iDest, iStart, remainder, iCache) in enumerate(contents):
        t_dest[k] = iDest; t_start[k] = -1 if iStart is None else iStart; t_cache[k] = -1 if iCache is None else iCache
        rem = list(remainder)
        if iStart is None:
            t_rho[k] = rho_lookup[rem[0]]; rem = rem[1:]
        gidx.extend(op_lookup[g] for g in rem)
        row_ptr[k + 1] = len(gidx)
    eff_ptr = np.zeros(R + 1, np.int64); el, ed = [], []
    for i in range(R):
        el.extend(atom.elbl_indices_by_expcircuit[i]); ed.extend(atom.elindices_by_expcircuit[i]); eff_ptr[i + 1] = len(el)
    return dict(t_dest=t_dest, t_start=t_start, t_cache=t_cache, t_rho=t_rho, row_ptr=row_ptr, gate_idx=np.array(gidx, np.int32),
                eff_ptr=eff_ptr, eff_label=np.array(el, np.int32), eff_dest=np.array(ed, np.int32),
                nE=np.int32(atom.num_elements), cache_size=np.int32(atom.cache_size))


def tp_map(model, atom):
    """One parameter per dense element for TPState / FullTPOp / the TPPOVM's parameterised effects, read off the members'
    deriv_wrt_params (a single 1.0 per column); the complement effect = identity - sum(others)."""
    from pygsti.modelmembers.povms.complementeffect import ComplementPOVMEffect
    nP = model.num_params
    tk = -np.ones(nP, np.int32); to = np.zeros(nP, np.int32); te = np.zeros(nP, np.int32)
    out = {}
    for kind, labels, typ in ((0, atom.op_labels, 'op'), (1, atom.rho_labels, 'prep'), (2, atom._fit_eff_labels, 'povm')):
        for oi, lbl in enumerate(labels):
            member = model._circuit_layer_operator(lbl, typ)
            if isinstance(member, ComplementPOVMEffect):
                others = []
                for oe in member.other_effects:
                    hits = [k for k, l2 in enumerate(atom._fit_eff_labels) if model._circuit_layer_operator(l2, 'povm') is oe]
                    assert len(hits) == 1
                    others.append(hits[0])
                out.update(comp_index=np.int32(oi), comp_others=np.array(others, np.int32),
                           comp_identity=np.ascontiguousarray(np.real(member.identity.to_dense()), dtype=np.float64).ravel())
                continue
            dm = np.real(member.deriv_wrt_params())
            idx = member.gpindices_as_array()
            rows_, cols_ = np.nonzero(dm)
            assert len(cols_) == len(idx) and np.array_equal(np.sort(cols_), np.arange(len(idx))) and (dm[rows_, cols_] == 1.0).all()
            assert (tk[idx] == -1).all()
            tk[idx[cols_]] = kind; to[idx[cols_]] = oi; te[idx[cols_]] = rows_
    out.update(tp_kind=tk, tp_obj=to, tp_elem=te)
    return out


def main():
    design = smq1Q_XYI.create_gst_experiment_design(max_max_length=8)
    target = smq1Q_XYI.target_model()
    datagen = target.depolarize(op_noise=0.05, spam_noise=0.025)
    ds = simulate_data(datagen, design.all_circuits_needing_data, 1000, sample_error='binomial', seed=2026)
    data = ProtocolData(design, ds)
    proto = gst.GateSetTomography(smq1Q_XYI.target_model("full TP"), 'stdgaugeopt', name="fit", verbosity=0)
    # (pyGSTi would otherwise drop gst_checkpoints/*.json into the working directory: protocols/gst.py:1497-1504)
    results = proto.run(data, simulator=RecordingMapForwardSimulator, disable_checkpointing=True)
    final = results.estimates["fit"].models['final iteration estimate']
    n_fit_calls = len(CALLS)
    tdl = two_delta_logl(final, ds)                   # (further simulator calls: not part of the replayed sequence)
    del CALLS[n_fit_calls:]

    atoms = []
    for c in CALLS:
        if not any(c["atom"] is a for a in atoms):
            atoms.append(c["atom"])
    out = dict(n_stages=np.int32(len(atoms)), D=np.int32(final.dim), nP=np.int32(final.num_params),
               derivative_eps=np.float64(final.sim.derivative_eps), two_delta_logl=np.float64(tdl),
               tdl_min_prob_clip=np.float64(1e-6), tdl_radius=np.float64(1e-4), tdl_clip_lo=np.float64(-1e6), tdl_clip_hi=np.float64(1e6),
               final_vec=final.to_vector())
    for s, atom in enumerate(atoms):
        pre = "s%d_" % s
        for k, v in table_arrays(atom).items():
            out[pre + k] = v
        out[pre + "op_labels"] = np.array([str(l) for l in atom.op_labels])
        out[pre + "rho_labels"] = np.array([str(l) for l in atom.rho_labels])
        out[pre + "eff_labels"] = np.array([str(l) for l in atom._fit_eff_labels])
    out.update(tp_map(final, atoms[-1]))
    for s, atom in enumerate(atoms):           # the same labels, in the same order, at every stage (one parameter map serves all)
        assert [str(l) for l in atom.op_labels] == [str(l) for l in atoms[-1].op_labels]
        assert [str(l) for l in atom._fit_eff_labels] == [str(l) for l in atoms[-1]._fit_eff_labels]
    # counts / totals per stage, from the objective that owned the stage's layout
    for o in OBJ:
        s = [i for i, a in enumerate(atoms) if a is o["layout"].atoms[0]]
        assert len(s) == 1 and len(o["layout"].atoms) == 1
        o["stage"] = s[0]
        out["s%d_counts" % s[0]] = o["counts"]; out["s%d_totals" % s[0]] = o["totals"]
    # Jacobian requests
    dcalls = [c for c in CALLS if c["what"] == "dprobs"]
    out["n_iterates"] = np.int32(len(dcalls))
    for k, c in enumerate(dcalls):
        pre = "it%d_" % k
        s = [i for i, a in enumerate(atoms) if a is c["atom"]][0]
        # the probabilities of the same parameter vector: the probs call that preceded this request on the same atom
        idx = [q for q, cc in enumerate(CALLS) if cc is c][0]
        pr = None
        for q in range(idx - 1, -1, -1):
            if CALLS[q]["what"] == "probs" and CALLS[q]["atom"] is c["atom"] and np.array_equal(CALLS[q]["vec"], c["vec"]):
                pr = CALLS[q]["out"]; break
        assert pr is not None
        g, r, e = c["dense"]
        out.update({pre + "stage": np.int32(s), pre + "vec": c["vec"], pre + "gates": g, pre + "rhos": r, pre + "effects": e,
                    pre + "probs": pr, pre + "dprobs": c["out"]})
    # objective-level records
    out["n_obj"] = np.int32(len(OBJ))
    for k, o in enumerate(OBJ):
        pre = "ob%d_" % k
        # the Jacobian request this dlsvec call made: the last dprobs call recorded before it returned
        it = sum(1 for c in CALLS[:o["n_calls_before"]] if c["what"] == "dprobs") - 1
        assert np.array_equal(dcalls[it]["vec"], o["vec"]) and atoms[o["stage"]] is dcalls[it]["atom"]
        out.update({pre + "kind": np.int32(o["kind"]), pre + "stage": np.int32(o["stage"]), pre + "vec": o["vec"], pre + "jtj": o["jtj"],
                    pre + "jtf": o["jtf"], pre + "fsum": np.float64(o["fsum"]), pre + "lsvec": o["lsvec"],
                    pre + "min_prob_clip_for_weighting": np.float64(o["reg"]["min_prob_clip_for_weighting"]),
                    pre + "min_p": np.float64(o["reg"]["min_p"]), pre + "radius": np.float64(o["reg"]["radius"]),
                    pre + "clip_lo": np.float64(o["clip"][0] if o["clip"] is not None else -np.inf),
                    pre + "clip_hi": np.float64(o["clip"][1] if o["clip"] is not None else np.inf)})
        out[pre + "iterate"] = np.int32(it)
    # the final estimate on the last stage's atom
    atom = atoms[-1]
    g, r, e = dense_of(final, atom)
    out.update(final_gates=g, final_rhos=r, final_effects=e)
    # every circuit of the data set is in the last stage's layout (L <= 8 design): 2 delta logL is a sum over its elements
    assert out["s%d_nE" % (len(atoms) - 1)] == 2 * len(ds)
    path = os.path.join(HERE, "fit_smq1Q_XYI_L8_TP.npz")
    np.savez_compressed(path, **out)
    print("wrote %s (%.1f KB): %d stages, %d Jacobian requests, %d dlsvec calls, 2dlogL = %.6f" % (
        path, os.path.getsize(path) / 1e3, len(atoms), len(dcalls), len(OBJ), tdl))
    for k, o in enumerate(OBJ):
        print("  dlsvec %2d: stage %d kind %d iterate %d  f=%.6f" % (k, o["stage"], o["kind"], int(out["ob%d_iterate" % k]), o["fsum"]))


if __name__ == "__main__":
    main()
