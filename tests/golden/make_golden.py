#!/usr/bin/env python3
"""Generate golden vectors for the forward-simulation hot path by IMPORTING the reference.

Runs only in the build container (needs a scratch build of the reference):

    PYTHONPATH=/tmp/pgref OMP_NUM_THREADS=1 python3 tests/golden/make_golden.py

It never travels to the GPU box; only the `.npz` files it writes (inputs + expected outputs,
pure data) are committed.  Each fixture holds

  inputs : model arrays in the order the reference's layout atom uses them (gates [nG,D,D],
           rhos [nR,D], effects [nEl,D]), the parameter -> (kind, object, element) map of the
           `full` parameterisation, the integerised circuit list, the reference's own prefix
           table (`_MapCOPALayoutAtom.table.contents`) as flat int arrays, the effect CSR
           (`elbl_indices_by_expcircuit`, `elindices_by_expcircuit`), the layout element order.
  outputs: probs (Cython map), dprobs_map (Cython FD, eps 1e-7), dprobs_matrix (analytic,
           MatrixForwardSimulator), hprobs_map / hprobs_matrix for small cases.

Reference entry points exercised: ForwardSimulator.bulk_fill_probs / bulk_fill_dprobs /
bulk_fill_hprobs (pygsti/forwardsims/forwardsim.py:584,628,701) on MapForwardSimulator
(mapforwardsim.py:111) and MatrixForwardSimulator (matrixforwardsim.py:578).
"""
import os
import sys
import hashlib
import numpy as np

import pygsti
from pygsti.baseobjs import Label
from pygsti.forwardsims import MapForwardSimulator, MatrixForwardSimulator

HERE = os.path.dirname(os.path.abspath(__file__))


def _label_str(lbl):
    return str(lbl)


def dump_case(name, model, circuits, dprobs_cols=None, want_matrix=True, want_hprobs=False,
              hprobs_blk=None, circuit_subset_for_matrix=None, extra=None, general_params=False,
              matrix_hprobs_blocks=None, matrix_hprobs=True, model_sets=False, dump_derivs=True, tp_map=False):
    """Build a 1-atom Map layout for `circuits`, run the reference, save everything."""
    assert model.sim.calclib.__name__.endswith('calc_densitymx'), "reference Cython path not built!"
    model = model.copy()
    model.sim = MapForwardSimulator(num_atoms=1)
    nP = model.num_params
    D = model.dim

    array_types = ('e', 'ep') + (('epp',) if want_hprobs else ())
    layout = model.sim.create_layout(circuits, array_types=array_types)
    assert len(layout.atoms) == 1
    atom = layout.atoms[0]
    nE = layout.num_elements

    # ---- model arrays in atom order -------------------------------------------------
    op_labels = list(atom.op_labels)
    rho_labels = list(atom.rho_labels)
    eff_labels = list(atom.full_effect_labels)   # a set: capture iteration order
    gates = np.array([model._circuit_layer_operator(l, 'op').to_dense('minimal') for l in op_labels])
    rhos = np.array([model._circuit_layer_operator(l, 'prep').to_dense('minimal') for l in rho_labels])
    effects = np.array([model._circuit_layer_operator(l, 'povm').to_dense('minimal') for l in eff_labels])
    gates = np.ascontiguousarray(gates.real.reshape(len(op_labels), D, D), dtype=np.float64)
    rhos = np.ascontiguousarray(rhos.real.reshape(len(rho_labels), D), dtype=np.float64)
    effects = np.ascontiguousarray(effects.real.reshape(len(eff_labels), D), dtype=np.float64)

    # ---- parameter map (full parameterisation: one param <-> one dense element) ---------
    # kind: 0 = gate, 1 = rho, 2 = effect
    pkind = -np.ones(nP, dtype=np.int32)
    pobj = -np.ones(nP, dtype=np.int32)
    pelem = -np.ones(nP, dtype=np.int32)
    for kind, labels, typ in ((0, op_labels, 'op'), (1, rho_labels, 'prep'), (2, eff_labels, 'povm')):
        for oi, lbl in enumerate(labels):
            member = model._circuit_layer_operator(lbl, typ)
            idx = member.gpindices_as_array()
            assert len(idx) == member.num_params
            n_el = D * D if kind == 0 else D
            if general_params:
                continue      # TP / CPTP ...: parameters are not dense elements; see the `dv_*` arrays below
            assert len(idx) == n_el, "fixture generator assumes `full` parameterisation"
            pkind[idx] = kind
            pobj[idx] = oi
            pelem[idx] = np.arange(n_el)
    paramvec = model.to_vector().copy()
    # consistency: the parameter value IS the dense element
    for p in range(nP):
        if pkind[p] == 0:
            assert gates[pobj[p]].flat[pelem[p]] == paramvec[p]
        elif pkind[p] == 1:
            assert rhos[pobj[p]].flat[pelem[p]] == paramvec[p]
        elif pkind[p] == 2:
            assert effects[pobj[p]].flat[pelem[p]] == paramvec[p]
        else:
            pass  # parameter of a gate/effect this atom never applies: kind -1, derivative exactly 0

    # ---- general parameterisations: d(dense element)/d(parameter) of every object, as the reference's members
    # return it (modelmember.deriv_wrt_params(), consumed by matrixforwardsim.py:_doperation / _dprobs_from_rho_e)
    dv = {}
    if general_params and dump_derivs:
        k_l, o_l, n_l, pi_l, d_l = [], [], [], [], []
        h_l, hz_l = [], []
        for kind, labels, typ in ((0, op_labels, 'op'), (1, rho_labels, 'prep'), (2, eff_labels, 'povm')):
            for oi, lbl in enumerate(labels):
                member = model._circuit_layer_operator(lbl, typ)
                dm = np.ascontiguousarray(np.real(member.deriv_wrt_params()), dtype=np.float64)
                idx = member.gpindices_as_array()
                assert dm.shape == ((D * D if kind == 0 else D), len(idx))
                k_l.append(kind); o_l.append(oi); n_l.append(len(idx)); pi_l.append(idx); d_l.append(dm.ravel())
                # second derivatives of the dense elements (zero for linear parameterisations): what
                # MatrixForwardSimulator._hoperation consumes (matrixforwardsim.py:192-224)
                if member.has_nonzero_hessian():
                    hm = np.ascontiguousarray(np.real(member.hessian_wrt_params()), dtype=np.float64)
                    assert hm.shape == (dm.shape[0], len(idx), len(idx))
                    h_l.append(hm.ravel()); hz_l.append(1)
                else:
                    h_l.append(np.zeros(0)); hz_l.append(0)
        dv = dict(dv_kind=np.array(k_l, np.int32), dv_obj=np.array(o_l, np.int32), dv_ncols=np.array(n_l, np.int32),
                  dv_param_idx=np.concatenate(pi_l).astype(np.int64), dv_deriv=np.concatenate(d_l),
                  dv2_nonzero=np.array(hz_l, np.int32), dv2_hess=np.concatenate(h_l))
    if general_params and tp_map:
        # "full TP" models too large for the dv_* tensors (D = 64): the one-parameter-per-element map itself, read off
        # each member's deriv_wrt_params() (a single 1.0 per column; the complement effect has no parameter of its own)
        from pygsti.modelmembers.povms.complementeffect import ComplementPOVMEffect as _CPE
        tk = -np.ones(nP, np.int32); to = np.zeros(nP, np.int32); te = np.zeros(nP, np.int32)
        for kind, labels, typ in ((0, op_labels, 'op'), (1, rho_labels, 'prep'), (2, eff_labels, 'povm')):
            for oi, lbl in enumerate(labels):
                member = model._circuit_layer_operator(lbl, typ)
                if isinstance(member, _CPE):
                    continue
                dm = np.real(member.deriv_wrt_params())
                idx = member.gpindices_as_array()
                rows_, cols_ = np.nonzero(dm)
                assert len(cols_) == len(idx) and np.array_equal(np.sort(cols_), np.arange(len(idx))) and (dm[rows_, cols_] == 1.0).all()
                assert (tk[idx] == -1).all()
                tk[idx[cols_]] = kind; to[idx[cols_]] = oi; te[idx[cols_]] = rows_
                del dm
        dv.update(tp_kind=tk, tp_obj=to, tp_elem=te)
    if general_params and (dump_derivs or tp_map):
        # TP POVMs: the complement effect = identity - sum(other effects) (modelmembers/povms/complementeffect.py:72-78)
        from pygsti.modelmembers.povms.complementeffect import ComplementPOVMEffect
        for oi, lbl in enumerate(eff_labels):
            member = model._circuit_layer_operator(lbl, 'povm')
            if isinstance(member, ComplementPOVMEffect):
                others = []
                for oe in member.other_effects:
                    hits = [k for k, l2 in enumerate(eff_labels) if model._circuit_layer_operator(l2, 'povm') is oe]
                    assert len(hits) == 1
                    others.append(hits[0])
                dv.update(comp_index=np.int32(oi), comp_others=np.array(others, np.int32),
                          comp_identity=np.ascontiguousarray(np.real(member.identity.to_dense()), dtype=np.float64).ravel())

    # ---- reference prefix table as flat ints -----------------------------------------
    op_lookup = {l: i for i, l in enumerate(op_labels)}
    rho_lookup = {l: i for i, l in enumerate(rho_labels)}
    contents = atom.table.contents
    R = len(contents)
    t_dest = np.empty(R, np.int32); t_start = np.empty(R, np.int32)
    t_cache = np.empty(R, np.int32); t_rho = -np.ones(R, np.int32)
    row_ptr = np.zeros(R + 1, np.int64)
    gidx = []
    for k, (iDest, iStart, remainder, iCache) in enumerate(contents):
        t_dest[k] = iDest
        t_start[k] = -1 if iStart is None else iStart
        t_cache[k] = -1 if iCache is None else iCache
        rem = list(remainder)
        if iStart is None:
            t_rho[k] = rho_lookup[rem[0]]
            rem = rem[1:]
        gidx.extend(op_lookup[g] for g in rem)
        row_ptr[k + 1] = len(gidx)
    gate_idx = np.array(gidx, dtype=np.int32)

    # ---- effect CSR, indexed by expanded-circuit index (== iDest) ---------------------
    nX = len(atom.elbl_indices_by_expcircuit)
    assert nX == R
    eff_ptr = np.zeros(nX + 1, np.int64)
    eff_label, eff_dest = [], []
    for i in range(nX):
        eff_label.extend(atom.elbl_indices_by_expcircuit[i])
        eff_dest.extend(atom.elindices_by_expcircuit[i])
        eff_ptr[i + 1] = len(eff_label)
    eff_label = np.array(eff_label, np.int32)
    eff_dest = np.array(eff_dest, np.int32)
    assert atom.element_slice == slice(0, nE)

    # ---- integerised circuit list in the caller's order + the layout's element order -----
    circ_ptr = np.zeros(len(circuits) + 1, np.int64)
    cg = []
    for ci, c in enumerate(circuits):
        cg.extend(op_lookup[l] for l in c.layertup if l in op_lookup)  # (explicit prep / POVM labels are not gates)
        circ_ptr[ci + 1] = len(cg)
    circ_gates = np.array(cg, np.int32)
    # element k <-> (circuit index, outcome index into eff_labels-less outcome tuple)
    el_circuit = np.empty(nE, np.int32)
    el_outcome = np.empty(nE, np.int32)
    outcome_names = None
    for ci, c in enumerate(circuits):
        inds = layout.indices_for_index(ci)
        outs = layout.outcomes_for_index(ci)
        inds = np.arange(inds.start, inds.stop, inds.step or 1) if isinstance(inds, slice) else np.asarray(inds)
        if outcome_names is None:
            outcome_names = []
        for k, o in zip(inds, outs):
            if ''.join(o) not in outcome_names:
                outcome_names.append(''.join(o))
            el_circuit[k] = ci
            el_outcome[k] = outcome_names.index(''.join(o))

    # ---- reference outputs -----------------------------------------------------------
    probs = np.empty(nE, 'd')
    model.sim.bulk_fill_probs(probs, layout)
    out = dict(probs=probs)

    if dprobs_cols is None:
        dprobs_cols = np.arange(nP)
    dprobs_cols = np.asarray(dprobs_cols, dtype=np.int64)
    out['dprobs_cols'] = dprobs_cols
    ralloc = layout.resource_alloc('param-processing')
    if len(dprobs_cols) == nP:
        J = np.empty((nE, nP), 'd')
        model.sim.bulk_fill_dprobs(J, layout)
    else:
        J = np.empty((nE, len(dprobs_cols)), 'd')
        # column subset through the per-atom seam (distforwardsim.py:148-152)
        model.sim._bulk_fill_dprobs_atom(J, None, atom, dprobs_cols, ralloc)
    out['dprobs_map'] = J
    # the FD loop must restore the model exactly
    assert np.array_equal(model.to_vector(), paramvec)

    if model_sets:
        # the dense model after each finite-difference step of mapfill_dprobs_atom (pyx:362-381: undo the previous
        # parameter, step the current one) -- the input of gst_fill_dprobs_models for parameterisations whose parameters
        # are not dense elements
        def dense_sets():
            g = np.array([model._circuit_layer_operator(l, 'op').to_dense('minimal') for l in op_labels])
            r = np.array([model._circuit_layer_operator(l, 'prep').to_dense('minimal') for l in rho_labels])
            e = np.array([model._circuit_layer_operator(l, 'povm').to_dense('minimal') for l in eff_labels])
            return (np.ascontiguousarray(g.real.reshape(len(op_labels), D, D), dtype=np.float64),
                    np.ascontiguousarray(r.real.reshape(len(rho_labels), D), dtype=np.float64),
                    np.ascontiguousarray(e.real.reshape(len(eff_labels), D), dtype=np.float64))
        eps = model.sim.derivative_eps
        mg, mr, me = [], [], []
        prev = None
        for i in dprobs_cols:
            if prev is None:
                model.set_parameter_value(int(i), paramvec[i] + eps)
            else:
                model.set_parameter_values([int(prev), int(i)], [paramvec[prev], paramvec[i] + eps])
            g, r, e = dense_sets(); mg.append(g); mr.append(r); me.append(e)
            prev = i
        if prev is not None:
            model.set_parameter_value(int(prev), paramvec[prev])
        assert np.array_equal(model.to_vector(), paramvec)
        out.update(mm_gates=np.array(mg), mm_rhos=np.array(mr), mm_effects=np.array(me))

    if model_sets and want_hprobs:
        # two-level stepping of _mapfill_hprobs_atom (mapforwardsim.py:420-436): for a few rows i the model at
        # theta + eps e_i (from_vector(..., close=True)) and, from there, after every FD step over a few columns j
        # (set_parameter_value(s), pyx:362-381); index 0 of the first axis = the unstepped model
        eps_h = model.sim.hessian_eps
        rows2 = np.asarray(hprobs_blk[0], np.int64)[:3]
        cols2 = np.asarray(hprobs_blk[1], np.int64)[:6]

        def dense_sets2():
            g = np.array([model._circuit_layer_operator(l, 'op').to_dense('minimal') for l in op_labels])
            r = np.array([model._circuit_layer_operator(l, 'prep').to_dense('minimal') for l in rho_labels])
            e = np.array([model._circuit_layer_operator(l, 'povm').to_dense('minimal') for l in eff_labels])
            return (np.ascontiguousarray(g.real.reshape(len(op_labels), D, D), dtype=np.float64),
                    np.ascontiguousarray(r.real.reshape(len(rho_labels), D), dtype=np.float64),
                    np.ascontiguousarray(e.real.reshape(len(eff_labels), D), dtype=np.float64))
        G2, R2, E2 = [], [], []
        for i in [None] + [int(q) for q in rows2]:
            vec = paramvec.copy()
            if i is not None:
                vec[i] += eps_h
            model.from_vector(vec, close=True)
            gs, rs, es = [], [], []
            g, r, e = dense_sets2(); gs.append(g); rs.append(r); es.append(e)
            prev = None
            for j in (int(q) for q in cols2):
                if prev is None:
                    model.set_parameter_value(j, vec[j] + eps_h)
                else:
                    model.set_parameter_values([prev, j], [vec[prev], vec[j] + eps_h])
                g, r, e = dense_sets2(); gs.append(g); rs.append(r); es.append(e)
                prev = j
            model.set_parameter_value(prev, vec[prev])
            G2.append(gs); R2.append(rs); E2.append(es)
        model.from_vector(paramvec)
        out.update(mm2_rows=rows2, mm2_cols=cols2, mm2_gates=np.array(G2), mm2_rhos=np.array(R2), mm2_effects=np.array(E2))

    if want_hprobs:
        b1, b2 = hprobs_blk
        b1 = np.asarray(b1, np.int64); b2 = np.asarray(b2, np.int64)
        H = np.empty((nE, len(b1), len(b2)), 'd')
        model.sim._bulk_fill_hprobs_atom(H, None, None, atom, b1, b2, ralloc)
        out['hprobs_map'] = H
        out['hprobs_rows'] = b1
        out['hprobs_cols'] = b2
        assert np.allclose(model.to_vector(), paramvec, atol=0, rtol=0)

    if want_matrix:
        m2 = model.copy()
        m2.sim = MatrixForwardSimulator(num_atoms=1)
        sub = circuits if circuit_subset_for_matrix is None else [circuits[i] for i in circuit_subset_for_matrix]
        lay2 = m2.sim.create_layout(sub, array_types=array_types)
        # the Matrix layout orders elements differently: map by (circuit, outcome)
        J2 = np.empty((lay2.num_elements, nP), 'd')
        p2 = np.empty(lay2.num_elements, 'd')
        m2.sim.bulk_fill_dprobs(J2, lay2, pr_array_to_fill=p2)
        if want_hprobs and matrix_hprobs:
            H2 = np.empty((lay2.num_elements, nP, nP), 'd')
            m2.sim.bulk_fill_hprobs(H2, lay2)
        sub_idx = list(range(len(circuits))) if circuit_subset_for_matrix is None else list(circuit_subset_for_matrix)
        rows_map = []   # element indices (Map layout) in the order we store the matrix results
        rows_mat = []
        for si, ci in enumerate(sub_idx):
            inds_map = layout.indices_for_index(ci)
            inds_mat = lay2.indices_for_index(si)
            inds_map = np.arange(nE)[inds_map]
            inds_mat = np.arange(lay2.num_elements)[inds_mat]
            outs_map = layout.outcomes_for_index(ci)
            outs_mat = lay2.outcomes_for_index(si)
            for k, o in zip(inds_map, outs_map):
                rows_map.append(k)
                rows_mat.append(inds_mat[list(outs_mat).index(o)])
        rows_map = np.array(rows_map, np.int64)
        rows_mat = np.array(rows_mat, np.int64)
        for bi, (s1, s2) in enumerate(matrix_hprobs_blocks or []):
            # exact Hessian blocks straight from the Matrix simulator's per-atom seam (parameter SLICES)
            Hb = np.empty((lay2.num_elements, s1.stop - s1.start, s2.stop - s2.start), 'd')
            m2.sim._bulk_fill_hprobs_atom(Hb, None, None, lay2.atoms[0], s1, s2, lay2.resource_alloc('param-processing'))
            out['mh%d_idx1' % bi] = np.arange(s1.start, s1.stop)
            out['mh%d_idx2' % bi] = np.arange(s2.start, s2.stop)
            out['mh%d_hprobs' % bi] = Hb[rows_mat]
        out['matrix_rows'] = rows_map                      # Map-layout element index of each stored row
        out['probs_matrix'] = p2[rows_mat]
        out['dprobs_matrix'] = J2[rows_mat][:, dprobs_cols]
        if want_hprobs and matrix_hprobs:
            out['hprobs_matrix'] = H2[rows_mat][:, b1][:, :, b2]

    meta = dict(
        D=np.int32(D), nP=np.int32(nP), nE=np.int32(nE), cache_size=np.int32(atom.cache_size),
        gates=gates, rhos=rhos, effects=effects, paramvec=paramvec,
        pkind=pkind, pobj=pobj, pelem=pelem,
        t_dest=t_dest, t_start=t_start, t_cache=t_cache, t_rho=t_rho, row_ptr=row_ptr, gate_idx=gate_idx,
        eff_ptr=eff_ptr, eff_label=eff_label, eff_dest=eff_dest,
        circ_ptr=circ_ptr, circ_gates=circ_gates, el_circuit=el_circuit, el_outcome=el_outcome,
        op_labels=np.array([_label_str(l) for l in op_labels]),
        rho_labels=np.array([_label_str(l) for l in rho_labels]),
        eff_labels=np.array([_label_str(l) for l in eff_labels]),
        outcome_names=np.array(outcome_names),
        derivative_eps=np.float64(model.sim.derivative_eps), hessian_eps=np.float64(model.sim.hessian_eps),
    )
    if extra:
        meta.update(extra)
    meta.update(dv)
    meta.update(out)
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **meta)
    print("%-28s nC=%6d nE=%7d nP=%5d rows=%6d A=%8d  -> %s (%.1f kB)" % (
        name, len(circuits), nE, nP, R, len(gate_idx), os.path.basename(path), os.path.getsize(path) / 1024))
    return meta


def circuit_list_hash(circ_ptr, circ_gates):
    h = hashlib.sha256()
    h.update(np.asarray(circ_ptr, np.int64).tobytes())
    h.update(np.asarray(circ_gates, np.int32).tobytes())
    return h.hexdigest()


def main():
    from pygsti.modelpacks import smq1Q_XYI, smq2Q_XYICNOT
    which = sys.argv[1:] or ['1q4', '1q4k', '1q128', '2q2', '2qdeep', 'designs', 'tp', 'multispam', '3q', 'cptp2q']

    if '1q4' in which:   # BASELINE configs[0] / SURVEY C1: smq1Q_XYI L in {1,2,4}
        m = smq1Q_XYI.target_model().depolarize(op_noise=0.01, spam_noise=0.01)
        circs = list(smq1Q_XYI.create_gst_experiment_design(4).all_circuits_needing_data)
        blk = (np.arange(0, 60, 7), np.arange(3, 60, 5))
        dump_case('smq1Q_XYI_L4_depol', m, circs, want_hprobs=True, hprobs_blk=blk)

    if '1q4k' in which:  # kicked model: generic dense values (FD-vs-analytic stress, run_me_with_mpiexec.py:40)
        m = smq1Q_XYI.target_model().depolarize(op_noise=0.01, spam_noise=0.01).kick(0.1, seed=1234)
        circs = list(smq1Q_XYI.create_gst_experiment_design(4).all_circuits_needing_data)
        dump_case('smq1Q_XYI_L4_kick', m, circs, want_hprobs=False)

    if 'multispam' in which:   # two preparations and two POVMs (one of them with three effects), explicit in the circuits
        import pygsti
        from pygsti.circuits import Circuit
        from pygsti.modelmembers.povms import UnconstrainedPOVM
        from pygsti.modelmembers.states import FullState
        m = smq1Q_XYI.target_model().depolarize(op_noise=0.02, spam_noise=0.01).kick(0.05, seed=7)
        rng = np.random.default_rng(11)
        m.preps['rho1'] = FullState(np.array([1 / np.sqrt(2), 0.1, -0.2, -0.55]) + 0.01 * rng.standard_normal(4), evotype=m.evotype, state_space=m.state_space)
        e0 = np.array([0.5, 0.1, 0.2, 0.1]); e1 = np.array([0.4, -0.1, 0.05, 0.2]); e2 = np.array([np.sqrt(2), 0, 0, 0]) - e0 - e1
        m.povms['Mtri'] = UnconstrainedPOVM([('a', e0), ('b', e1), ('c', e2)], evotype=m.evotype, state_space=m.state_space)
        base = list(smq1Q_XYI.create_gst_experiment_design(2).all_circuits_needing_data)
        circs = []
        for k, c in enumerate(base):
            prep = 'rho0' if k % 3 else 'rho1'
            povm = 'Mdefault' if k % 2 else 'Mtri'
            circs.append(Circuit((prep,) + tuple(c.layertup) + (povm,), line_labels=c.line_labels))
        circs.append(Circuit(('rho1', 'Mtri'), line_labels=base[0].line_labels))          # empty gate string
        blk = (np.arange(0, m.num_params, 9), np.arange(2, m.num_params, 4))
        dump_case('smq1Q_multispam_L2', m, circs, want_hprobs=True, hprobs_blk=blk)

    if '1q128' in which:  # BASELINE configs[1] / SURVEY C2
        m = smq1Q_XYI.target_model().depolarize(op_noise=0.01, spam_noise=0.01)
        circs = list(smq1Q_XYI.create_gst_experiment_design(128).all_circuits_needing_data)
        dump_case('smq1Q_XYI_L128_depol', m, circs, want_matrix=True)

    if 'tp' in which:    # general parameterisations (analytic mode + gst_set_derivs): TP and CPTP-constrained models
        circs = list(smq1Q_XYI.create_gst_experiment_design(4).all_circuits_needing_data)
        m = smq1Q_XYI.target_model("full TP").depolarize(op_noise=0.01, spam_noise=0.01)
        # FD-of-FD Hessian block of the TP model (Map simulator): rho, all four effect parameters (they also move the
        # complement's outcome), gate parameters
        blk = (np.array([0, 3, 4, 5, 6, 8, 20, 42]), np.concatenate([np.arange(0, 8), np.arange(9, 43, 3)]))
        dump_case('smq1Q_XYI_L4_TP', m, circs, general_params=True, want_hprobs=True, hprobs_blk=blk)
        m = smq1Q_XYI.target_model("CPTPLND")
        m.from_vector(m.to_vector() + 0.01 * np.random.default_rng(3).standard_normal(m.num_params))
        blk = (np.array([0, 2, 7, 8, 13, 25, 26, 40, 59]), np.arange(0, 60, 3))
        dump_case('smq1Q_XYI_L4_CPTPLND', m, circs, general_params=True, want_hprobs=True, hprobs_blk=blk, model_sets=True)
        m = smq2Q_XYICNOT.target_model("full TP").depolarize(op_noise=0.01, spam_noise=0.01)
        circs2 = list(smq2Q_XYICNOT.create_gst_experiment_design(1, lite=True).all_circuits_needing_data)
        cols = np.sort(np.random.default_rng(5).choice(m.num_params, 120, replace=False))
        # (parameters 15..62 are the three parameterised effects of the TP POVM; 63.. the gates)
        blk2 = (np.array([2, 17, 40, 63 + 240 + 5]), np.array([16, 17, 33, 40, 49, 62, 1, 63 + 17, 63 + 240 + 5, 63 + 4 * 240 + 100]))
        dump_case('smq2Q_XYICNOT_L1_TP', m, circs2, dprobs_cols=cols, circuit_subset_for_matrix=list(range(0, len(circs2), 7)),
                  general_params=True, want_hprobs=True, hprobs_blk=blk2, matrix_hprobs=False)

    if 'cptp2q' in which:   # 2Q CPTPLND (composed static target x exponentiated Lindblad error generator): Map FD columns
        m = smq2Q_XYICNOT.target_model("CPTPLND")
        m.from_vector(m.to_vector() + 0.003 * np.random.default_rng(9).standard_normal(m.num_params))
        circs2 = list(smq2Q_XYICNOT.create_gst_experiment_design(1, lite=True).all_circuits_needing_data)
        cols = np.sort(np.random.default_rng(6).choice(m.num_params, 40, replace=False))
        cols[:3] = [0, 1, 2]
        cols = np.unique(cols)
        dump_case('smq2Q_XYICNOT_L1_CPTPLND', m, circs2, dprobs_cols=cols, want_matrix=False, general_params=True,
                  model_sets=True, dump_derivs=False)      # (the members' derivative tensors would be 0.5 GB)

    if '2q2' in which:   # 2Q, D=16: every circuit of the L<=2 lite design, a spread of 96 columns
        m = smq2Q_XYICNOT.target_model().depolarize(op_noise=0.01, spam_noise=0.01)
        circs = list(smq2Q_XYICNOT.create_gst_experiment_design(2, lite=True).all_circuits_needing_data)
        rng = np.random.default_rng(7)
        cols = np.sort(np.concatenate([np.arange(0, 16), np.arange(16, 80, 5),
                                       rng.choice(np.arange(80, 1616), 67, replace=False)]))
        sub = list(range(0, len(circs), 9))
        dump_case('smq2Q_XYICNOT_L2_depol', m, circs, dprobs_cols=cols, want_matrix=True,
                  circuit_subset_for_matrix=sub,
                  matrix_hprobs_blocks=[(slice(74, 90), slice(0, 140)), (slice(328, 344), slice(1560, 1616)),
                                        (slice(8, 20), slice(590, 620))])

    if '2qdeep' in which:  # SURVEY C3 slice: whole germ-power families at L<=1024, 64 columns
        m = smq2Q_XYICNOT.target_model().depolarize(op_noise=0.01, spam_noise=0.01)
        design = smq2Q_XYICNOT.create_gst_experiment_design(1024, lite=True)
        allc = list(design.all_circuits_needing_data)
        germs = smq2Q_XYICNOT.germs(lite=True)
        prepf = smq2Q_XYICNOT.prep_fiducials(); measf = smq2Q_XYICNOT.meas_fiducials()
        fam = []
        for g in (germs[1], germs[5], germs[10]):          # Gxpi2:0, Gcnot, a 5-gate germ
            for p in (1, 4, 64, 1024):
                rep = p // len(g)
                if rep == 0: continue
                for pf in (prepf[0], prepf[6], prepf[15]):
                    for mf in (measf[0], measf[3], measf[8]):
                        fam.append(pf + g * rep + mf)
        allset = set(allc)
        fam = [c for c in dict.fromkeys(fam) if c in allset]
        cols = np.arange(m.num_params)       # (round 3: EVERY column at depth ~1030, where rounding is largest; round 2 pinned 64)
        dump_case('smq2Q_XYICNOT_L1024_deep', m, fam, dprobs_cols=cols, want_matrix=False)

    if '3q' in which:    # BASELINE configs[4] / SURVEY C5: 3-qubit explicit densitymx model, D = 64, 10 gates, nP = 41,536
        from pygsti.processors import QubitProcessorSpec
        from pygsti.models import modelconstruction as mc
        from pygsti.circuits import Circuit
        ps = QubitProcessorSpec(3, ['Gxpi2', 'Gypi2', 'Gcnot'], geometry='line')
        m = mc.create_explicit_model(ps, ideal_gate_type='full', ideal_spam_type='full')
        m = m.depolarize(op_noise=0.01, spam_noise=0.01).kick(0.02, seed=64)       # generic dense values
        ops = list(m.operations.keys())
        assert m.num_params == 41536 and m.dim == 64 and len(ops) == 10
        rng = np.random.default_rng(0)                                             # seeded random strings
        lens = np.concatenate([[0, 1, 2, 64, 64, 256, 200, 128], rng.integers(1, 65, 200)])
        strs = [tuple(ops[g] for g in rng.integers(0, len(ops), L)) for L in lens]
        for k in range(12, len(strs), 5):                                          # shared prefixes (prefix-table hits)
            strs[k] = strs[k - 1][:len(strs[k - 1]) // 2] + strs[k][:10]
        circs = [Circuit(s, line_labels=(0, 1, 2)) for s in dict.fromkeys(strs)]
        nP = m.num_params
        # 256 FD columns: rho, effects of several outcomes, rows / columns of several gates incl. both ends of the vector
        cols = np.sort(np.unique(np.concatenate([
            [0, 1, 63, 64, 65, 64 + 63, 64 + 64, 64 + 7 * 64 + 63, 576, 577, 576 + 63, 576 + 64, 576 + 4095, 576 + 4096, nP - 1],
            rng.choice(nP, 256, replace=False)]))[:256])
        blk = (np.array([0, 70, 576, 600, 576 + 64, 9000, 20000, nP - 1]),
               np.sort(np.unique(np.concatenate([[0, 1, 70, 576, 577, 576 + 64 + 1, 30000, nP - 2], rng.choice(nP, 40, replace=False)]))[:32]))
        dump_case('3q_explicit_L64', m, circs, dprobs_cols=cols, want_matrix=False, want_hprobs=True, hprobs_blk=blk)

    if '3qtp' in which:  # the 3-qubit model as "full TP": FD Jacobian columns and an FD-of-FD Hessian block of a D = 64 plan with a complement effect
        from pygsti.processors import QubitProcessorSpec
        from pygsti.models import modelconstruction as mc
        from pygsti.circuits import Circuit
        ps = QubitProcessorSpec(3, ['Gxpi2', 'Gypi2', 'Gcnot'], geometry='line')
        m = mc.create_explicit_model(ps, ideal_gate_type='full TP', ideal_spam_type='full TP')
        m = m.depolarize(op_noise=0.01, spam_noise=0.01)
        m.from_vector(m.to_vector() + 0.02 * np.random.default_rng(65).standard_normal(m.num_params))   # (kick() would turn the TP members into full ones)
        from pygsti.modelmembers.operations import FullTPOp
        assert all(isinstance(o, FullTPOp) for o in m.operations.values())
        ops = list(m.operations.keys())
        assert m.dim == 64 and len(ops) == 10
        nP = m.num_params
        assert nP == 63 + 7 * 64 + 10 * 64 * 63, nP
        rng = np.random.default_rng(11)
        lens = np.concatenate([[0, 1, 2, 40, 33], rng.integers(1, 25, 31)])
        strs = [tuple(ops[g] for g in rng.integers(0, len(ops), L)) for L in lens]
        for k in range(8, len(strs), 4):
            strs[k] = strs[k - 1][:len(strs[k - 1]) // 2] + strs[k][:6]
        circs = [Circuit(s, line_labels=(0, 1, 2)) for s in dict.fromkeys(strs)]
        g0 = 63 + 7 * 64          # first gate parameter
        cols = np.sort(np.unique(np.concatenate([[0, 1, 62, 63, 64, 63 + 63, 63 + 64, 63 + 6 * 64 + 5, g0 - 1, g0, g0 + 1, g0 + 4031, g0 + 4032, nP - 1],
                                                 rng.choice(nP, 40, replace=False)]))[:48])
        blk = (np.array([1, 63 + 2, 63 + 3 * 64 + 9, g0 + 70, g0 + 5 * 4032 + 123, nP - 1]),
               np.sort(np.unique(np.concatenate([[0, 63, 63 + 2, 63 + 64 + 7, g0 - 1, g0, g0 + 70, g0 + 71, nP - 2], rng.choice(nP, 8, replace=False)]))[:16]))
        dump_case('3q_explicit_TP', m, circs, dprobs_cols=cols, want_matrix=False, want_hprobs=True, hprobs_blk=blk,
                  general_params=True, dump_derivs=False, tp_map=True)

    if 'designs' in which:
        # pins for the build's own circuit generator: counts and sha256 of integerised lists, plus
        # the model data of the two packs (target superoperators, SPAM, fiducials, germs)
        rec = {}
        for pack, tag, Ls, lites in ((smq1Q_XYI, '1Q', [1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024], (True, False)),
                                     (smq2Q_XYICNOT, '2Q', [1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024], (True, False))):
            tm = pack.target_model()
            ops = list(tm.operations.keys())
            lookup = {l: i for i, l in enumerate(ops)}
            rec[tag + '_op_labels'] = np.array([str(l) for l in ops])
            rec[tag + '_gates'] = np.array([tm.operations[l].to_dense() for l in ops])
            rec[tag + '_rho'] = tm.preps['rho0'].to_dense()
            rec[tag + '_effects'] = np.array([tm.povms['Mdefault'][k].to_dense() for k in tm.povms['Mdefault'].keys()])
            rec[tag + '_effect_labels'] = np.array(list(tm.povms['Mdefault'].keys()))
            dm = tm.depolarize(op_noise=0.01, spam_noise=0.01)
            rec[tag + '_gates_depol'] = np.array([dm.operations[l].to_dense() for l in ops])
            rec[tag + '_rho_depol'] = dm.preps['rho0'].to_dense()
            rec[tag + '_effects_depol'] = np.array([dm.povms['Mdefault'][k].to_dense() for k in dm.povms['Mdefault'].keys()])
            rec[tag + '_paramvec_depol'] = dm.to_vector()

            def enc(clist):
                ptr = np.zeros(len(clist) + 1, np.int64); g = []
                for i, c in enumerate(clist):
                    g.extend(lookup[l] for l in c.layertup); ptr[i + 1] = len(g)
                return ptr, np.array(g, np.int32)
            for nm, lst in (('prep_fiducials', pack.prep_fiducials()), ('meas_fiducials', pack.meas_fiducials()),
                            ('germs_lite', pack.germs(lite=True)), ('germs_full', pack.germs(lite=False))):
                p, g = enc(lst)
                rec['%s_%s_ptr' % (tag, nm)] = p
                rec['%s_%s_gates' % (tag, nm)] = g
            for lite in lites:
                counts, hashes, depth = [], [], []
                for L in Ls:
                    if tag == '2Q' and not lite and L not in (1, 2, 1024):
                        counts.append(-1); hashes.append(''); depth.append(-1); continue
                    cl = list(pack.create_gst_experiment_design(L, lite=lite).all_circuits_needing_data)
                    p, g = enc(cl)
                    counts.append(len(cl)); hashes.append(circuit_list_hash(p, g)); depth.append(len(g))
                    print(tag, 'lite' if lite else 'full', L, len(cl), len(g), hashes[-1][:12])
                key = '%s_%s' % (tag, 'lite' if lite else 'full')
                rec[key + '_L'] = np.array(Ls)
                rec[key + '_counts'] = np.array(counts)
                rec[key + '_depth'] = np.array(depth)
                rec[key + '_sha256'] = np.array(hashes)
        np.savez_compressed(os.path.join(HERE, 'designs.npz'), **rec)
        print("designs.npz written")


if __name__ == '__main__':
    main()
