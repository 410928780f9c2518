"""Analytic derivative mode (GST_DERIV_ANALYTIC) on the GPU against the analytic oracles: the golden vectors
of the reference's MatrixForwardSimulator (tolerance 1e-8, the north-star bar; observed ~1e-12) and the numpy
forward/backward Jacobian of oracle/oracle.py on the deep 2Q circuits where Matrix is infeasible."""
import numpy as np
import pytest

from conftest import force, load_fixture, assert_bitwise, plan_from_fixture
from pygsti_amd import _lib

pytestmark = pytest.mark.gpu
TOL = 1e-8


@pytest.mark.parametrize("name", ["smq1Q_XYI_L4_depol", "smq1Q_XYI_L4_kick", "smq1Q_XYI_L128_depol",
                                  "smq2Q_XYICNOT_L2_depol", "smq1Q_multispam_L2"])
def test_analytic_dprobs_vs_matrix_simulator(name):
    fx = load_fixture(name)
    pl = plan_from_fixture(fx)
    pr = np.empty(int(fx["nE"]))
    J = pl.fill_dprobs(param_idx=fx["dprobs_cols"], probs_out=pr, mode=_lib.DERIV_ANALYTIC)
    rows = fx["matrix_rows"]
    assert np.abs(J[rows] - fx["dprobs_matrix"]).max() < TOL
    assert_bitwise(pr, fx["probs"], "probabilities of the analytic call are the base pass's")
    # and it is NOT the finite-difference Jacobian (which carries O(eps * p'') truncation error)
    assert np.abs(J - fx["dprobs_map"]).max() > 1e-9


def test_analytic_deep_circuits_vs_numpy_oracle(oracle_built):
    fx = load_fixture("smq2Q_XYICNOT_L1024_deep")     # depth up to 1030, |J| up to 65
    pl = plan_from_fixture(fx)
    cols = np.concatenate([np.arange(0, 100), np.arange(336, 400), np.arange(1100, 1130), np.arange(1360, 1616)])
    J = pl.fill_dprobs(param_idx=cols, mode=_lib.DERIV_ANALYTIC)
    Jo, _ = oracle_built.analytic_dprobs(fx, cols)
    err = np.abs(J - Jo).max()
    assert err < TOL, err                 # ABSOLUTE (|J| reaches 65 here; observed ~1e-12): the north star's bar, not a relative one
    none = fx["pkind"][cols] == -1
    assert none.any() and (J[:, none] == 0).all()


def test_analytic_column_window_and_permutation():
    fx = load_fixture("smq2Q_XYICNOT_L2_depol")
    pl = plan_from_fixture(fx)
    nE = int(fx["nE"])
    full = pl.fill_dprobs(param_idx=np.arange(1616), mode=_lib.DERIV_ANALYTIC)
    rng = np.random.default_rng(3)
    cols = rng.permutation(1616)[:300]
    out = np.full((nE, 320), -3.0)
    pl.fill_dprobs(out=out, param_idx=cols, dest_idx=np.arange(300) + 7, mode=_lib.DERIV_ANALYTIC)
    assert np.array_equal(out[:, 7:307], full[:, cols])
    assert (out[:, :7] == -3.0).all() and (out[:, 307:] == -3.0).all()


def test_analytic_hprobs_vs_numpy_oracle(oracle_built):
    """gst_fill_hprobs_analytic at D = 16 against the exact numpy Hessian (oracle.analytic_hprobs, itself pinned to the
    MatrixForwardSimulator vectors of the 1Q fixture by tests/test_oracle.py): gate x gate, gate x SPAM, SPAM x SPAM
    pairs, the same parameter in both blocks, a destination window."""
    fx = load_fixture("smq2Q_XYICNOT_L2_depol")
    pl = plan_from_fixture(fx)
    nE = int(fx["nE"])
    i1 = np.array([0, 5, 17, 40, 80, 81, 97, 335, 336, 600, 1615])         # rho, effects, several gates
    i2 = np.concatenate([np.arange(0, 20), np.arange(70, 100), [335, 336, 337, 600, 601, 1200, 1615]])
    H = pl.fill_hprobs(idx1=i1, idx2=i2, mode=_lib.DERIV_ANALYTIC)
    sub = np.arange(0, nE, 23)                                               # (the numpy oracle is slow: a sample of elements)
    fx_small = dict(fx)
    Ho = oracle_built.analytic_hprobs(fx, i1, i2)
    scale = max(1.0, np.abs(Ho).max())
    assert np.abs(H - Ho).max() < 1e-8 * scale
    assert np.abs(H).max() > 0.1
    # FD-of-FD is close, not equal
    Hfd = pl.fill_hprobs(idx1=i1[:4], idx2=i2[:30], eps=1e-5)
    assert np.abs(Hfd - H[:, :4, :30]).max() < 5e-2 * scale
    # destination window / row offsets
    out = np.full((nE, 6, 40), -5.0)
    pl.fill_hprobs(out=out, idx1=i1[2:5], idx2=i2[10:30], dest1=np.array([1, 2, 4]), dest2=np.arange(20) + 7, mode=_lib.DERIV_ANALYTIC)
    assert np.array_equal(out[:, [1, 2, 4], 7:27], H[:, 2:5, 10:30])
    mask = np.ones((6, 40), bool); mask[np.ix_([1, 2, 4], np.arange(7, 27))] = False
    assert (out[:, mask] == -5.0).all()


def test_analytic_hprobs_vs_matrix_simulator_blocks():
    """Exact Hessian blocks against MatrixForwardSimulator._bulk_fill_hprobs_atom vectors (2Q, three parameter
    rectangles covering rho x gate, effect x gate, gate x gate within one gate and across two gates)."""
    fx = load_fixture("smq2Q_XYICNOT_L2_depol")
    pl = plan_from_fixture(fx)
    rows = fx["matrix_rows"]
    for b in range(3):
        i1, i2, ref = fx["mh%d_idx1" % b], fx["mh%d_idx2" % b], fx["mh%d_hprobs" % b]
        H = pl.fill_hprobs(idx1=i1, idx2=i2, mode=_lib.DERIV_ANALYTIC)
        assert np.abs(H[rows] - ref).max() < TOL, b
        assert np.abs(ref).max() > 1e-3


@pytest.mark.parametrize("name", ["smq1Q_XYI_L4_depol", "smq1Q_multispam_L2"])
def test_analytic_hprobs_1q_vs_matrix_simulator(name):
    """D = 4: exact Hessian block against the MatrixForwardSimulator hprobs vectors of the 1Q fixtures (the two-cache
    contraction without MFMA: analytic_small_kernel + dwalk_kernel<4>); the second one has two preparations and two
    POVMs."""
    fx = load_fixture(name)
    pl = plan_from_fixture(fx)
    H = pl.fill_hprobs(idx1=fx["hprobs_rows"], idx2=fx["hprobs_cols"], mode=_lib.DERIV_ANALYTIC)
    ref = fx["hprobs_matrix"]
    assert np.abs(H[fx["matrix_rows"]] - ref).max() < TOL
    full = pl.fill_hprobs(mode=_lib.DERIV_ANALYTIC)                       # the whole nP x nP Hessian of every element
    assert np.abs(full - np.transpose(full, (0, 2, 1))).max() < 1e-11     # symmetric
    assert np.array_equal(full[:, fx["hprobs_rows"]][:, :, fx["hprobs_cols"]], H)


def test_exact_hessian_with_derivative_caches_beyond_4gb():
    """The contraction kernels address the derivative-state caches with 32-bit per-lane byte offsets; a plan whose FORWARD
    trie is large (8.4 M states: 4 x 16 x 8 bytes each = 4.3 GB of dF) takes their 64-bit instantiation instead of being
    refused (rounds 1-2: GST_EUNSUPPORTED) -- and must not be answered with wrapped offsets: the block's rows for the
    lexicographically last circuits (whose states sit beyond the 4 GB mark) and for a spread of others equal what a small
    plan holding only those circuits returns through the default instantiation.  8,400 circuits of depth 1,000: distinct
    random prefixes of 200 gates in front of one common suffix of 800."""
    from pygsti_amd import _lib
    rng = np.random.default_rng(0)
    D, nG, nEl, nC, L = 16, 6, 4, 8400, 1000
    suffix = rng.integers(0, nG, 800)
    circs = [np.concatenate([rng.integers(0, nG, L - 800), suffix]) for _ in range(nC)]
    gates = np.eye(D)[None] * 0.98 + 0.01 * rng.standard_normal((nG, D, D))
    rho = np.zeros((1, D)); rho[0, 0] = 0.5
    eff = 0.05 * rng.standard_normal((nEl, D)); eff[:, 0] += 0.5
    nP = D + nEl * D + nG * D * D
    kind = np.concatenate([np.full(D, 1), np.full(nEl * D, 2), np.full(nG * D * D, 0)]).astype(np.int32)
    obj = np.concatenate([np.zeros(D), np.repeat(np.arange(nEl), D), np.repeat(np.arange(nG), D * D)]).astype(np.int32)
    elem = np.concatenate([np.arange(D), np.tile(np.arange(D), nEl), np.tile(np.arange(D * D), nG)]).astype(np.int32)

    def make(sel):
        n = len(sel)
        ptr = np.arange(n + 1, dtype=np.int64) * L
        g = np.concatenate([circs[c] for c in sel]).astype(np.int32)
        pl = _lib.Plan.from_circuits(D, nG, 1, nEl, n * nEl, np.zeros(n, np.int32), ptr, g, np.arange(n + 1, dtype=np.int64) * nEl,
                                     np.tile(np.arange(nEl, dtype=np.int32), n), np.arange(n * nEl, dtype=np.int32))
        pl.set_model(gates, rho, eff)
        pl.set_param_map(kind, obj, elem)
        return pl
    pl = make(np.arange(nC))
    st = pl.stats()
    assert st["trie_nodes"] > 8.3e6
    rows, cols = np.array([80, 7]), np.array([81, 82, 400, 3])          # gate x gate, rho x gate, gate x rho
    H = pl.fill_hprobs(idx1=rows, idx2=cols, mode=_lib.DERIV_ANALYTIC)
    J = pl.fill_dprobs(param_idx=cols, mode=_lib.DERIV_ANALYTIC)
    order = sorted(range(nC), key=lambda c: circs[c].tolist())
    sel = np.array(order[-24:] + order[:8] + order[1000:7000:500])
    small = make(sel)
    Hs = small.fill_hprobs(idx1=rows, idx2=cols, mode=_lib.DERIV_ANALYTIC)
    Js = small.fill_dprobs(param_idx=cols, mode=_lib.DERIV_ANALYTIC)
    el = (sel[:, None] * nEl + np.arange(nEl)[None, :]).ravel()
    assert np.abs(Hs).max() > 1e-3
    assert np.abs(H[el] - Hs).max() < 1e-11 * max(1.0, np.abs(Hs).max())
    assert np.abs(J[el] - Js).max() < 1e-12 * max(1.0, np.abs(Js).max())
    assert np.isfinite(H).all()
    pl.close(); small.close()


def test_analytic_fill_overwrites_every_requested_entry():
    """Circuits that never apply some gate must get exact zeros in that gate's columns, not whatever the buffer held:
    the destination is pre-filled with NaN and a sentinel."""
    from pygsti_amd import _lib
    fx = load_fixture("smq2Q_XYICNOT_L2_depol")
    pl = plan_from_fixture(fx)
    nE, nP = int(fx["nE"]), int(fx["nP"])
    ld = nP + 3
    d_J = pl.device_malloc(nE * ld * 8)
    fill = np.full((nE, ld), np.nan); fill[:, nP:] = -7.0
    pl.memcpy_h2d(d_J, fill)
    pl.fill_dprobs_dev(d_J, ld, np.arange(nP), None, 1e-7, None, _lib.DERIV_ANALYTIC); pl.sync()
    J = pl.memcpy_d2h(np.empty((nE, ld)), d_J)
    assert np.isfinite(J[:, :nP]).all(), "entries left unwritten: %d" % int((~np.isfinite(J[:, :nP])).sum())
    assert (J[:, nP:] == -7.0).all()
    cols = fx["dprobs_cols"]
    assert np.abs(J[fx["matrix_rows"]][:, cols] - fx["dprobs_matrix"]).max() < TOL
    # where the FD Jacobian is exactly zero (a circuit that never applies the column's gate, among others) this one is ~0
    zero = (fx["dprobs_map"] == 0.0)
    assert np.abs(J[:, cols][zero]).max() <= 1e-8
    pl.device_free(d_J)


def test_analytic_keep_zeros_option():
    """GST_OPT_ANALYTIC_KEEP_ZEROS: a repeated analytic fill into the SAME destination with the SAME columns does not
    store the structural zeros again.  First fill: everything is written (the destination is pre-filled with NaN);
    second fill into the same device buffer: same bits; the zeros really are skipped (a sentinel planted in a
    structural-zero entry survives the second fill, and is overwritten again as soon as the columns, the destination or
    the option change)."""
    from pygsti_amd import _lib
    fx = load_fixture("smq2Q_XYICNOT_L2_depol")
    pl = plan_from_fixture(fx)
    nE, nP = int(fx["nE"]), int(fx["nP"])
    cols = np.arange(nP)
    ref = pl.fill_dprobs(param_idx=cols, mode=_lib.DERIV_ANALYTIC)
    pl.set_option(_lib.OPT_ANALYTIC_KEEP_ZEROS, 1)
    d = pl.device_malloc(nE * nP * 8)
    pl.memcpy_h2d(d, np.full(nE * nP, np.nan))
    pl.fill_dprobs_dev(d, nP, cols, None, 1e-7, None, _lib.DERIV_ANALYTIC); pl.sync()
    J1 = pl.memcpy_d2h(np.empty((nE, nP)), d)
    assert np.array_equal(J1, ref)
    # a structural zero: an element whose circuit never applies some gate (a whole 256-column block of exact zeros)
    # (a sentinel in every such block: the ones of work items none of whose circuits applies the gate are skipped; a
    #  circuit paired with one that does apply it still gets its zeros stored with the partner's block)
    blocks = np.abs(J1[:, 80:]).reshape(nE, -1, 256).max(2)
    zb = np.argwhere(blocks == 0.0)
    assert len(zb) > 100
    poisoned = J1.copy()
    pos = zb[:, 0] * nP + 80 + 256 * zb[:, 1] + 5
    poisoned.ravel()[pos] = 123.456
    pl.memcpy_h2d(d, poisoned)
    pl.fill_dprobs_dev(d, nP, cols, None, 1e-7, None, _lib.DERIV_ANALYTIC); pl.sync()
    J2 = pl.memcpy_d2h(np.empty((nE, nP)), d)
    kept = J2.ravel()[pos] == 123.456
    assert kept.sum() > 0.3 * len(pos), "structural zeros of whole items must not have been stored again (%d of %d kept)" % (kept.sum(), len(pos))
    J2.ravel()[pos[kept]] = 0.0
    assert np.array_equal(J2, ref)
    pos = int(pos[kept][0]); planted = np.array([123.456])
    # another column request: everything is written again
    pl.fill_dprobs_dev(d, nP, cols[::-1].copy(), np.arange(nP)[::-1].copy(), 1e-7, None, _lib.DERIV_ANALYTIC); pl.sync()
    assert np.array_equal(pl.memcpy_d2h(np.empty((nE, nP)), d), ref)
    # option off: zeros stored every time
    pl.memcpy_h2d(d + pos * 8, planted)
    pl.set_option(_lib.OPT_ANALYTIC_KEEP_ZEROS, 0)
    pl.fill_dprobs_dev(d, nP, cols[::-1].copy(), np.arange(nP)[::-1].copy(), 1e-7, None, _lib.DERIV_ANALYTIC); pl.sync()
    assert np.array_equal(pl.memcpy_d2h(np.empty((nE, nP)), d), ref)
    pl.device_free(d)
    with pytest.raises(ValueError):
        pl.set_option(99, 1)


def test_resident_zeros_by_default_on_tracked_destinations():
    """Default (GST_OPT_ANALYTIC_KEEP_ZEROS = 2): memory the caller allocated as TRACKED (gst_device_malloc_tracked: an
    explicit statement per allocation; plain gst_device_malloc memory is never claimed) lets a repeated exact fill into it
    skip the structural zeros WITHOUT a per-fill promise -- and every library write in between (h2d copy, an FD fill, an objective
    map, the caller's gst_device_touch, a row scaling with a non-finite factor) makes the next fill store everything again.
    Sentinels cannot be planted through the library here (that is a tracked write): the decision is read from
    gst_stats.last_zeros_resident and the results are compared with a fresh host fill every time."""
    from pygsti_amd import _lib
    fx = load_fixture("smq2Q_XYICNOT_L2_depol")
    pl = plan_from_fixture(fx)
    nE, nP = int(fx["nE"]), int(fx["nP"])
    cols = np.arange(nP)
    ref = pl.fill_dprobs(param_idx=cols, mode=_lib.DERIV_ANALYTIC)
    d = pl.device_malloc(nE * nP * 8, tracked=True)

    def fill(dst=d, ld=nP, c=cols, dest=None):
        pl.fill_dprobs_dev(dst, ld, c, dest, 1e-7, None, _lib.DERIV_ANALYTIC); pl.sync()
        return bool(pl.stats()["last_zeros_resident"])

    def check():
        assert np.array_equal(pl.memcpy_d2h(np.empty((nE, nP)), d), ref)

    pl.memcpy_h2d(d, np.full(nE * nP, np.nan))
    assert not fill(); check()                      # first fill of the destination: everything stored over the NaNs
    assert fill(); check()                          # second: zeros resident
    assert fill(); check()
    pl.memcpy_h2d(d, np.full(nE * nP, np.nan))      # a library write in between
    assert not fill(); check()
    assert fill()
    pl.memcpy_h2d(d + 8 * (nE * nP - 1), np.array([np.nan]))      # ... of a single entry
    assert not fill(); check()
    assert fill()
    pl.fill_dprobs_dev(d, nP, cols, None, 1e-7, None, _lib.DERIV_FD); pl.sync()       # an FD Jacobian over it (no exact zeros in general)
    assert not fill(); check()
    assert fill()
    pl.device_touch(d + 800, 16)                   # the caller's own kernel wrote there
    assert not fill(); check()
    assert fill()
    # another column request / leading dimension: nothing is assumed
    assert not fill(c=cols[::-1].copy(), dest=np.arange(nP)[::-1].copy()); check()
    assert fill(c=cols[::-1].copy(), dest=np.arange(nP)[::-1].copy()); check()
    assert not fill(); check()
    # row scalings keep zeros zero ...
    w = pl.device_malloc(nE * 8); jtj = pl.device_malloc(nP * nP * 8)
    scale = np.linspace(0.5, 2.0, nE)
    pl.memcpy_h2d(w, scale)
    assert fill()
    pl.fill_jtj_dev(d, nE, nP, nP, jtj, w); pl.sync()
    assert np.allclose(pl.memcpy_d2h(np.empty((nE, nP)), d), ref * scale[:, None], rtol=1e-15, atol=0)
    assert fill(); check()
    # ... unless a factor is not finite: 0 * inf = NaN in a structural zero, which the next fill must overwrite.  The host
    # cannot know (the factors live on the device): the claim's device word is cleared on the stream and the kernel reads it
    scale[3] = np.inf
    pl.memcpy_h2d(w, scale)
    pl.fill_jtj_dev(d, nE, nP, nP, jtj, w); pl.sync()
    assert np.isnan(pl.memcpy_d2h(np.empty((nE, nP)), d)[3]).any()
    fill(); check()                                # (stats say "resident": the host's view; the kernel stored everything)
    assert fill(); check()
    # foreign memory (a pointer straight from the HIP runtime: its owner may write it with kernels the library never sees)
    # is never trusted
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    raw = C.c_void_p()
    assert hip.hipMalloc(C.byref(raw), C.c_size_t(nE * nP * 8)) == 0
    assert not fill(dst=raw.value)
    assert not fill(dst=raw.value)
    assert np.array_equal(pl.memcpy_d2h(np.empty((nE, nP)), raw.value), ref)
    hip.hipFree(raw)
    # ... and neither is the library's own PLAIN allocation: its caller never said that only the library writes it
    plain = pl.device_malloc(nE * nP * 8)
    assert not fill(dst=plain) and not fill(dst=plain)
    assert np.array_equal(pl.memcpy_d2h(np.empty((nE, nP)), plain), ref)
    pl.device_free(plain)
    # a row scaling issued through ANOTHER plan (another stream) ends the claim instead of racing with its device word
    other = plan_from_fixture(fx)
    scale[3] = 1.0
    pl.memcpy_h2d(w, scale)
    assert fill()
    other.fill_jtj_dev(d, nE, nP, nP, jtj, w); other.sync()
    assert not fill(); check()
    assert fill()
    # freed and re-allocated tracked memory starts untrusted
    pl.device_free(d)
    d2 = pl.device_malloc(nE * nP * 8, tracked=True)
    pl.memcpy_h2d(d2, np.full(nE * nP, np.nan))
    assert not fill(dst=d2)
    assert np.array_equal(pl.memcpy_d2h(np.empty((nE, nP)), d2), ref)
    # host destinations go through the plan's staging buffer, which only the library writes: resident from the second fill
    # on, and an FD fill (same staging buffer) in between resets it
    pl2 = plan_from_fixture(fx)
    a1 = pl2.fill_dprobs(param_idx=cols, mode=_lib.DERIV_ANALYTIC); r1 = pl2.stats()["last_zeros_resident"]
    a2 = pl2.fill_dprobs(param_idx=cols, mode=_lib.DERIV_ANALYTIC); r2 = pl2.stats()["last_zeros_resident"]
    pl2.fill_dprobs(param_idx=cols, mode=_lib.DERIV_FD)
    a3 = pl2.fill_dprobs(param_idx=cols, mode=_lib.DERIV_ANALYTIC); r3 = pl2.stats()["last_zeros_resident"]
    assert (r1, r2, r3) == (0, 1, 0) and np.array_equal(a1, ref) and np.array_equal(a2, ref) and np.array_equal(a3, ref)
    # option 0: never
    pl.set_option(_lib.OPT_ANALYTIC_KEEP_ZEROS, 0)
    assert not fill(dst=d2) and not fill(dst=d2)
    for p_ in (w, jtj, d2):
        pl.device_free(p_)


def test_state_caches_beyond_32bit_offsets_take_the_wide_contraction(monkeypatch):
    """The two-cache MFMA contraction addresses its state caches with 32-bit per-lane byte offsets (4 GB each).  A plan
    whose caches are larger is no longer refused (round 2: GST_EUNSUPPORTED) or degraded: the same kernels, instantiated
    with 64-bit lane offsets, run it -- Jacobians and exact Hessian blocks, D = 16 and D = 64.  GST_TEST_FORCE cache_limit=... stands
    in for the 4 GB so that small plans take that form: bit-identical to the default instantiation."""
    from pygsti_amd import _lib
    fx = load_fixture("smq2Q_XYICNOT_L2_depol")
    cols = fx["dprobs_cols"]
    pl0 = plan_from_fixture(fx)
    ref = pl0.fill_dprobs(param_idx=cols, mode=_lib.DERIV_ANALYTIC)
    Href = pl0.fill_hprobs(idx1=cols[:6], idx2=cols[:40], mode=_lib.DERIV_ANALYTIC)
    st = pl0.stats()
    for limit in (st["trie_nodes"] * 16 * 8 * 2,      # forward cache fits, the 4-effect backward cache does not
                  1024.0):                            # neither fits
        force(monkeypatch, cache_limit=limit)
        pl = plan_from_fixture(fx)
        J = pl.fill_dprobs(param_idx=cols, mode=_lib.DERIV_ANALYTIC)
        assert np.array_equal(J, ref)
        assert np.abs(J[fx["matrix_rows"]] - fx["dprobs_matrix"]).max() < 1e-8
        assert np.array_equal(pl.fill_hprobs(idx1=cols[:6], idx2=cols[:40], mode=_lib.DERIV_ANALYTIC), Href)
    # D = 64
    force(monkeypatch, cache_limit=None)
    fx3 = load_fixture("3q_explicit_L64")
    c3 = fx3["dprobs_cols"][:96]
    p3 = plan_from_fixture(fx3)
    ref3 = p3.fill_dprobs(param_idx=c3, mode=_lib.DERIV_ANALYTIC)
    H3 = p3.fill_hprobs(idx1=c3[:3], idx2=c3[:24], mode=_lib.DERIV_ANALYTIC)
    force(monkeypatch, cache_limit=1024)
    p3w = plan_from_fixture(fx3)
    assert np.array_equal(p3w.fill_dprobs(param_idx=c3, mode=_lib.DERIV_ANALYTIC), ref3)
    assert np.array_equal(p3w.fill_hprobs(idx1=c3[:3], idx2=c3[:24], mode=_lib.DERIV_ANALYTIC), H3)


def test_d64_analytic_jacobian_and_hessian_vs_matrix_simulator_directly():
    """The D = 64 analytic kernels (analytic_mfma64_kernel, dwalk64_kernel) against vectors taken STRAIGHT from the
    reference's MatrixForwardSimulator on the 3-qubit model (tests/golden/3q_explicit_matrix.npz, round 4; two 16-column
    Jacobian blocks covering the preparation / effect / gate boundaries and two exact Hessian blocks) -- until now D = 64
    was compared with the numpy restatement only.  Absolute 1e-8; the FD mode on the same plan stays bit-identical to
    the Map simulator's columns."""
    from conftest import matrix_rows_by_circuit
    fx = load_fixture("3q_explicit_matrix")
    pl = plan_from_fixture(fx)
    rows = matrix_rows_by_circuit(fx)
    cols = fx["matrix_cols"]
    pr = np.empty(int(fx["nE"]))
    J = pl.fill_dprobs(param_idx=cols, probs_out=pr, mode=_lib.DERIV_ANALYTIC)
    ref = fx["matrix_by_circuit_dprobs"][rows]
    assert np.abs(J - ref).max() < TOL, np.abs(J - ref).max()
    assert np.abs(ref).max() > 0.1
    assert np.abs(pr - fx["matrix_by_circuit_probs"][rows]).max() < 1e-10
    for b in (0, 1):
        H = pl.fill_hprobs(idx1=fx["mh%d_idx1" % b], idx2=fx["mh%d_idx2" % b], mode=_lib.DERIV_ANALYTIC)
        href = fx["mh%d_by_circuit_hprobs" % b][rows]
        assert np.abs(H - href).max() < TOL, (b, np.abs(H - href).max())
        assert np.abs(href).max() > 1e-3
    assert_bitwise(pl.fill_dprobs(param_idx=fx["dprobs_cols"], eps=float(fx["derivative_eps"])), fx["dprobs_map"], "D = 64 FD columns")
