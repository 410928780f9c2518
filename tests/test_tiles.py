"""Tiles of the D = 16 exact contraction (gst_kernels_tiles.hip; build_tiles in csrc/gst_fill_analytic.cpp).

CPU: the host forms the tiles of real designs and CHECKS them (gst_get_tile_stats): for every tiled circuit and gate, the
segment slots of its tile row / column together with its remnant lists are exactly its applications of that gate -- each named
by the STRINGS of its forward and backward state, since a tile reads a row's forward states and a column's backward states
through one member's ids.  GPU: the tile kernel's Jacobian against the item kernel's (the default), which the other
GPU tests pin to the Matrix simulator's golden columns."""
import numpy as np
import pytest

from conftest import load_fixture, plan_from_fixture


def _design_plan(L, lite, **kw):
    from pygsti_amd import modelpacks
    from pygsti_amd.layout import HipCOPALayout
    pack = modelpacks.smq2Q_XYICNOT
    model = pack.target_model().depolarize(0.01, 0.01)
    circuits = pack.create_gst_circuits(L, lite=lite)
    lay = HipCOPALayout(circuits, model, num_atoms=1, **kw)
    return pack, model, circuits, lay, lay.atoms[0].plan()


def test_tiles_of_gst_designs_are_consistent_with_the_application_tables():
    for L, min_share in ((64, 0.3), (256, 0.4)):
        pack, model, circuits, lay, plan = _design_plan(L, True)
        st = plan.tile_stats()
        assert st["inconsistencies"] == 0, st
        assert st["n_tiles"] > 0 and st["tiled_circuits"] == st["circuits_in_one_tile"], st
        assert st["tiled_circuits"] >= min_share * len(circuits), st
        # what the tiles serve from LDS dwarfs what their circuits still gather one by one
        assert st["segment_slots"] * 8 > st["remnant_applications"], st
        assert st["longest_segment"] <= L + 8             # (a germ power plus what every column's ending has in common)


def test_plans_without_product_structure_form_no_tiles():
    fx = load_fixture("smq1Q_XYI_L4_depol")              # D = 4
    assert plan_from_fixture(fx).tile_stats()["n_tiles"] == 0
    fx = load_fixture("smq2Q_XYICNOT_L2_depol")           # D = 16 but every middle string is shorter than a tile's minimum
    st = plan_from_fixture(fx).tile_stats()
    assert st["n_tiles"] == 0 and st["inconsistencies"] == 0


@pytest.mark.gpu
def test_gpu_tile_kernel_equals_the_item_kernel():
    """The same exact Jacobian twice -- tiles + item kernel for the rest, and the default, the item kernel alone -- on a design whose circuits are
    mostly tiled (2Q, L <= 128 lite: 16,567 circuits): they differ by the re-association of the sums only; resident zeros,
    column windows and a shuffled sub-request take the same route."""
    from pygsti_amd import _lib
    pack, model, circuits, lay, plan = _design_plan(128, True)
    G, R, E = lay.model_arrays(model)
    nP, nE = model.num_params, lay.global_num_elements
    plan.set_model(G, R, E); plan.set_param_map(*lay.param_map(model))
    plan.set_option(_lib.OPT_ANALYTIC_TILES, 1)                 # (off by default: include/gstfwd.h says why)
    d = plan.device_malloc(nE * nP * 8, tracked=True)
    pidx = np.arange(nP, dtype=np.int64)
    plan.fill_dprobs_dev(d, nP, pidx, None, 1e-7, None, _lib.DERIV_ANALYTIC)
    plan.sync()
    st = plan.stats()
    assert st["last_tiles"] > 0 and st["last_tiled_circuits"] > 0.3 * len(circuits), st
    Jt = np.empty((nE, nP)); plan.memcpy_d2h(Jt, d)
    plan.fill_dprobs_dev(d, nP, pidx, None, 1e-7, None, _lib.DERIV_ANALYTIC)          # second fill: the zeros stay resident
    plan.sync()
    assert plan.stats()["last_zeros_resident"] == 1
    Jt2 = np.empty((nE, nP)); plan.memcpy_d2h(Jt2, d)
    assert np.array_equal(Jt, Jt2)
    lay2 = type(lay)(circuits, model, num_atoms=1)
    plan2 = lay2.atoms[0].plan()
    plan2.set_model(G, R, E); plan2.set_param_map(*lay2.param_map(model))
    d2 = plan2.device_malloc(nE * nP * 8)
    plan2.fill_dprobs_dev(d2, nP, pidx, None, 1e-7, None, _lib.DERIV_ANALYTIC)
    plan2.sync()
    assert plan2.stats()["last_tiles"] == 0
    Ji = np.empty((nE, nP)); plan2.memcpy_d2h(Ji, d2)
    scale = np.abs(Ji).max()
    assert np.isfinite(Jt).all() and np.abs(Jt - Ji).max() <= 1e-12 * scale, (np.abs(Jt - Ji).max(), scale)
    # a shuffled sub-request into a window of a wider host array (column maps instead of contiguous gate blocks)
    rng = np.random.default_rng(3)
    cols = rng.permutation(nP)[:300]
    dest = rng.permutation(320)[:300]
    out = np.full((nE, 320), np.nan)
    plan.fill_dprobs(out=out, param_idx=cols, dest_idx=dest, mode=_lib.DERIV_ANALYTIC)
    assert plan.stats()["last_tiles"] > 0
    assert np.abs(out[:, dest] - Ji[:, cols]).max() <= 1e-12 * scale
    assert np.isnan(out[:, np.setdiff1d(np.arange(320), dest)]).all()
    plan.device_free(d); plan2.device_free(d2)
