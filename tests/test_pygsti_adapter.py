"""Adapter for the real pyGSTi (runs only where the reference is importable, e.g. the build container with
PYTHONPATH=/tmp/pgref; skipped on the GPU box).  Without a GPU we check everything up to the launch: the
subclass plugs into `model.sim`, pyGSTi's own MapCOPALayout is created, the atom's prefix table becomes a
libgstfwd plan whose programs -- interpreted in numpy -- reproduce pyGSTi's bulk_fill_probs bit for bit,
the parameter map matches pyGSTi's gpindices, and a fill raises GstDeviceError (no silent CPU fallback)."""
import numpy as np
import pytest

pygsti = pytest.importorskip("pygsti")

from _interp import run_programs                      # noqa: E402
from conftest import assert_bitwise                    # noqa: E402
from pygsti_amd import _lib                            # noqa: E402
from pygsti_amd import pygsti_adapter as A             # noqa: E402


def test_adapter_plan_matches_pygsti():
    from pygsti.modelpacks import smq1Q_XYI
    from pygsti.forwardsims import MapForwardSimulator
    model = smq1Q_XYI.target_model().depolarize(op_noise=0.01, spam_noise=0.01)
    circuits = list(smq1Q_XYI.create_gst_experiment_design(8).all_circuits_needing_data)
    ref = model.copy(); ref.sim = MapForwardSimulator(num_atoms=2)
    lay_ref = ref.sim.create_layout(circuits, array_types=("e", "ep"))
    p_ref = np.empty(lay_ref.num_elements); ref.sim.bulk_fill_probs(p_ref, lay_ref)

    model.sim = A.HipMapForwardSimulator(num_atoms=2)
    assert isinstance(model.sim, MapForwardSimulator) and model.sim.model is model
    layout = model.sim.create_layout(circuits, array_types=("e", "ep"))      # pyGSTi's own MapCOPALayout
    assert len(layout.atoms) == 2
    out = np.empty(layout.num_elements)
    for atom in layout.atoms:
        plan = A.atom_plan(model, atom)
        G, R, E = A.atom_arrays(model, atom)
        kind, obj, elem = A.atom_param_map(model, atom)
        v = model.to_vector()
        for p in range(model.num_params):
            if kind[p] >= 0:
                assert (G, R, E)[kind[p]][obj[p]].ravel()[elem[p]] == v[p]
        w, off = plan.program()
        n = len(atom.elbl_indices_by_expcircuit)
        eff_ptr = np.zeros(n + 1, np.int64); el, ed = [], []
        for i in range(n):
            el.extend(atom.elbl_indices_by_expcircuit[i]); ed.extend(atom.elindices_by_expcircuit[i]); eff_ptr[i + 1] = len(el)
        o, written, _ = run_programs(w, off, G, R, E, eff_ptr, np.array(el), np.array(ed), atom.num_elements)
        assert (written == 1).all()
        out[atom.element_slice] = o
    assert_bitwise(out, p_ref, "adapter plan vs pyGSTi bulk_fill_probs")
    if _lib.device_count() == 0:
        with pytest.raises(_lib.GstDeviceError):
            model.sim.bulk_fill_probs(np.empty(layout.num_elements), layout)


def test_adapter_rejects_non_full_parameterisations():
    from pygsti.modelpacks import smq1Q_XYI
    model = smq1Q_XYI.target_model("CPTPLND")
    model.sim = A.HipMapForwardSimulator()
    layout = model.sim.create_layout(list(smq1Q_XYI.create_gst_experiment_design(1).all_circuits_needing_data))
    with pytest.raises(NotImplementedError):
        A.atom_plan(model, layout.atoms[0]); A.atom_arrays(model, layout.atoms[0]); A.atom_param_map(model, layout.atoms[0])
