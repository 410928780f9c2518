"""GPU parity tests proper: the HIP path, called through the C ABI, against (a) the golden vectors
generated from the reference and (b) the CPU oracle on the same inputs.  Bar: BIT-EXACT for probs,
FD dprobs and FD-of-FD hprobs (the device computes in the reference's arithmetic order)."""
import numpy as np
import pytest

from conftest import force, load_fixture, assert_bitwise, plan_from_fixture, page_locked_candidate

pytestmark = pytest.mark.gpu

FIXTURES = ["smq1Q_XYI_L4_depol", "smq1Q_XYI_L4_kick", "smq1Q_XYI_L128_depol",
            "smq2Q_XYICNOT_L2_depol", "smq2Q_XYICNOT_L1024_deep",
            "smq1Q_multispam_L2",       # two preparations, two POVMs (2 and 3 effects), an empty gate string
            "3q_explicit_L64"]          # 3 qubits, D = 64, 41,536 parameters: vectors of the reference's Cython Map path


@pytest.mark.parametrize("name", FIXTURES)
@pytest.mark.parametrize("target_tasks", [0, 7])
def test_probs_bitwise_vs_reference(name, target_tasks):
    fx = load_fixture(name)
    pl = plan_from_fixture(fx, target_tasks=target_tasks)
    p = pl.fill_probs()
    assert_bitwise(p, fx['probs'], "probs " + name)


@pytest.mark.parametrize("name", FIXTURES)
def test_dprobs_fd_bitwise_vs_reference(name):
    fx = load_fixture(name)
    pl = plan_from_fixture(fx)
    pr = np.empty(int(fx['nE']))
    J = pl.fill_dprobs(param_idx=fx['dprobs_cols'], eps=float(fx['derivative_eps']), probs_out=pr)
    assert_bitwise(pr, fx['probs'], "probs (pr_array_to_fill) " + name)
    assert_bitwise(J, fx['dprobs_map'], "dprobs " + name)


@pytest.mark.parametrize("fd_split", [2, 4])
def test_dprobs_fd_row_split_variants_bitwise(fd_split):
    """gst_options.fd_split: the rows of every mat-vec split over 2 / 4 wavefronts with an LDS exchange -- same bits."""
    for name in ("smq2Q_XYICNOT_L2_depol", "smq2Q_XYICNOT_L1024_deep"):
        fx = load_fixture(name)
        pl = plan_from_fixture(fx, fd_split=fd_split)
        J = pl.fill_dprobs(param_idx=fx['dprobs_cols'], eps=float(fx['derivative_eps']))
        assert_bitwise(J, fx['dprobs_map'], "dprobs fd_split=%d %s" % (fd_split, name))


@pytest.mark.parametrize("overlap,handover", [("1", "1"), ("1", "2"), ("1", "0"), ("0", "1")])
def test_dprobs_fd_base_pass_inside_persistent_launch_bitwise(overlap, handover, monkeypatch):
    """Persistent launch with the base pass walked INSIDE it (GST_TEST_FORCE overlap=1, the default for D = 16): designated
    wavefronts publish states and probabilities with write-through stores, the finite-difference walks wait on the
    sentinel the buffers were pre-filled with.  Same arithmetic, same bits -- including probs_out, which the chains fill
    -- whatever the hand-over setting; destinations pre-filled with NaN; repeated fills (stale cache contents)."""
    force(monkeypatch, persist=2, fused=0, overlap=overlap, handover=handover)   # (fused=0: plans this small would otherwise take the one-launch fused-lane form)
    forms = set()
    for name in ("smq2Q_XYICNOT_L1024_deep", "smq2Q_XYICNOT_L2_depol"):
        fx = load_fixture(name)
        for tt in (0, 3, 40):
            pl = plan_from_fixture(fx, target_tasks=tt)
            for rep in range(2):
                pr = np.full(int(fx['nE']), np.nan)
                J = np.full((int(fx['nE']), len(fx['dprobs_cols'])), np.nan)
                pl.fill_dprobs(out=J, param_idx=fx['dprobs_cols'], eps=float(fx['derivative_eps']), probs_out=pr)
                st = pl.stats()
                forms.add(st["last_fd_form"])
                assert st["last_fd_aborted"] == 0, (name, tt, st)
                assert_bitwise(pr, fx['probs'], "probs_out overlap=%s %s tasks=%d rep %d" % (overlap, name, tt, rep))
                assert_bitwise(J, fx['dprobs_map'], "dprobs overlap=%s handover=%s %s tasks=%d rep %d" % (overlap, handover, name, tt, rep))
            # a different model on the same plan: nothing of the previous fill's cache may survive
            g2 = np.array(fx['gates']) * 0.999
            pl.set_model(g2, fx['rhos'], fx['effects'])
            J2 = pl.fill_dprobs(param_idx=fx['dprobs_cols'], eps=float(fx['derivative_eps']))
            force(monkeypatch, persist=0)
            pl0 = plan_from_fixture(fx, target_tasks=tt)
            pl0.set_model(g2, fx['rhos'], fx['effects'])
            assert_bitwise(J2, pl0.fill_dprobs(param_idx=fx['dprobs_cols'], eps=float(fx['derivative_eps'])), "second model " + name)
            force(monkeypatch, persist=2)
    assert (2 in forms) == (overlap == "1"), forms


def test_dprobs_fd_bounded_waits_fall_back_to_the_standby_launches(monkeypatch):
    """A wait that cannot end -- here: the overlap launch is told to walk no chain at all (GST_TEST_FORCE skip_chains=1), as
    when the producing workgroup is not resident on a shared device -- runs out after ~0.1 s, raises the abort flag, and
    the stand-by launches enqueued behind the persistent one (separate base pass, one workgroup per pair, guarded by that
    flag) produce the reference's Jacobian bit for bit.  Nothing hangs; the next fill is unaffected."""
    import time
    force(monkeypatch, persist=2, fused=0, skip_chains=1)
    fx = load_fixture("smq2Q_XYICNOT_L1024_deep")
    pl = plan_from_fixture(fx)
    pr = np.full(int(fx['nE']), np.nan)
    t0 = time.perf_counter()
    J = pl.fill_dprobs(param_idx=fx['dprobs_cols'], eps=float(fx['derivative_eps']), probs_out=pr)
    dt = time.perf_counter() - t0
    st = pl.stats()
    assert st["last_fd_form"] == 2 and st["last_fd_aborted"] == 1, st
    assert dt < 20.0, "bounded waits must end within a fraction of a second each (took %.1f s)" % dt
    assert_bitwise(pr, fx['probs'], "probs after the fall-back")
    assert_bitwise(J, fx['dprobs_map'], "dprobs after the fall-back")
    force(monkeypatch, skip_chains=None)
    pl2 = plan_from_fixture(fx)
    J2 = pl2.fill_dprobs(param_idx=fx['dprobs_cols'], eps=float(fx['derivative_eps']))
    assert pl2.stats()["last_fd_aborted"] == 0
    assert_bitwise(J2, fx['dprobs_map'], "dprobs, next plan")


@pytest.mark.parametrize("handover", ["0", "1", "2"])
def test_dprobs_fd_walk_handover_bitwise(handover, monkeypatch):
    """Persistent launch with walks cut at their task's slot-free middle and handed from one SIMD to another
    (GST_TEST_FORCE handover=2: every walk that can be cut; 1: only to balance the queues; 0: never): the second half picks up
    the 64 lane states the first half stored -- same program words, same arithmetic, same bits."""
    force(monkeypatch, persist=2, handover=handover)
    for name in ("smq2Q_XYICNOT_L1024_deep", "smq2Q_XYICNOT_L2_depol", "smq1Q_XYI_L128_depol"):
        fx = load_fixture(name)
        for tt in (0, 3):
            pl = plan_from_fixture(fx, target_tasks=tt)
            pr = np.empty(int(fx['nE']))
            J = pl.fill_dprobs(param_idx=fx['dprobs_cols'], eps=float(fx['derivative_eps']), probs_out=pr)
            assert_bitwise(J, fx['dprobs_map'], "dprobs handover=%s %s tasks=%d" % (handover, name, tt))
            assert_bitwise(pl.fill_dprobs(param_idx=fx['dprobs_cols'], eps=float(fx['derivative_eps'])), J, "repeat " + name)


@pytest.mark.parametrize("mode", ["0", "2"])
def test_dprobs_fd_launch_forms_bitwise(mode, monkeypatch):
    """The two forms of the FD launch -- one workgroup per (task, wavefront) pair placed by the dispatcher
    (GST_TEST_FORCE persist=0), and persistent workgroups popping pairs from per-SIMD queues (=2: always) -- run the same
    arithmetic: same bits.  (The default picks by the number of pairs per SIMD.)"""
    force(monkeypatch, persist=mode)
    for name in ("smq1Q_XYI_L128_depol", "smq2Q_XYICNOT_L2_depol", "smq2Q_XYICNOT_L1024_deep"):
        fx = load_fixture(name)
        pl = plan_from_fixture(fx)
        J = pl.fill_dprobs(param_idx=fx['dprobs_cols'], eps=float(fx['derivative_eps']))
        assert_bitwise(J, fx['dprobs_map'], "dprobs persist=%s %s" % (mode, name))
        assert_bitwise(pl.fill_dprobs(param_idx=fx['dprobs_cols'], eps=float(fx['derivative_eps'])), J, "repeat " + name)


def test_dprobs_column_window_and_dest_indices():
    """dest_param_slice semantics (distforwardsim.py:130-144): fill a column window of a wider 'ep' array."""
    fx = load_fixture("smq1Q_XYI_L4_depol")
    pl = plan_from_fixture(fx)
    nE, nP = int(fx['nE']), int(fx['nP'])
    full = np.full((nE, nP + 5), -7.0)
    cols = np.arange(10, 30)
    pl.fill_dprobs(out=full, param_idx=cols, dest_idx=cols + 2, eps=1e-7)
    assert_bitwise(full[:, 12:32], fx['dprobs_map'][:, 10:30], "window")
    assert (full[:, :12] == -7.0).all() and (full[:, 32:] == -7.0).all()
    view = full[:, 3:40]      # non-contiguous rows (leading dimension nP+5)
    pl.fill_dprobs(out=view, param_idx=np.arange(0, 8), dest_idx=None, eps=1e-7)
    assert_bitwise(full[:, 3:11], fx['dprobs_map'][:, 0:8], "view")


@pytest.mark.parametrize("name", ["smq1Q_XYI_L4_depol", "smq2Q_XYICNOT_L2_depol"])
@pytest.mark.parametrize("direct", ["2", "1", "0"])      # 2: kernel-written whatever the window's width
def test_dprobs_into_page_locked_array_bitwise(name, direct, monkeypatch):
    """A destination registered with gst_host_register is written by the FD kernel itself (default; GST_TEST_FORCE host_direct=0 selects the copy) or
    by a copy from HBM (=0): same bits as the reference either way, (ld, dest_idx) window honoured, nothing outside it
    touched, and a later pageable destination still works."""
    from pygsti_amd import _lib
    force(monkeypatch, host_direct=direct)
    fx = load_fixture(name)
    pl = plan_from_fixture(fx)
    cols = fx["dprobs_cols"]; nE, n = int(fx["nE"]), len(cols)
    full = page_locked_candidate((nE, n + 7), -7.0)
    assert _lib.pin_host_array(full)
    pr = np.empty(nE)
    try:
        pl.fill_dprobs(out=full, param_idx=cols, dest_idx=np.arange(n) + 3, eps=float(fx["derivative_eps"]), probs_out=pr)
        assert_bitwise(full[:, 3:3 + n], fx["dprobs_map"], "page-locked destination")
        assert_bitwise(pr, fx["probs"], "probs_out")
        assert (full[:, :3] == -7.0).all() and (full[:, 3 + n:] == -7.0).all()
        # a window that starts inside the registered region
        pl.fill_dprobs(out=full[:, 1:], param_idx=cols[:5], dest_idx=None, eps=1e-7)
        assert_bitwise(full[:, 1:6], fx["dprobs_map"][:, :5], "view into the registered array")
    finally:
        _lib.unpin_host_array(full)
    assert_bitwise(pl.fill_dprobs(param_idx=cols, eps=1e-7), fx["dprobs_map"], "pageable destination afterwards")


@pytest.mark.parametrize("name", ["smq1Q_XYI_L4_depol", "smq2Q_XYICNOT_L2_depol"])
def test_analytic_dprobs_into_page_locked_array(name, monkeypatch):
    """The analytic contraction writes a page-locked destination itself as well (round 3): bit for bit what the staged route
    (host_direct=0: HBM, then a copy) returns, window honoured, NaN pre-fill fully overwritten, nothing outside touched;
    with GST_OPT_ANALYTIC_KEEP_ZEROS set the staged route is taken (the kernel then skips stores the host array never got)."""
    from pygsti_amd import _lib
    fx = load_fixture(name)
    cols = fx["dprobs_cols"]; nE, n = int(fx["nE"]), len(cols)
    force(monkeypatch, host_direct=0)
    ref = plan_from_fixture(fx).fill_dprobs(param_idx=cols, mode=_lib.DERIV_ANALYTIC)
    force(monkeypatch, host_direct=2)
    pl = plan_from_fixture(fx)
    full = page_locked_candidate((nE, n + 7), np.nan)
    assert _lib.pin_host_array(full)
    pr = np.empty(nE)
    try:
        full[:, :3] = -7.0; full[:, 3 + n:] = -7.0
        pl.fill_dprobs(out=full, param_idx=cols, dest_idx=np.arange(n) + 3, probs_out=pr, mode=_lib.DERIV_ANALYTIC)
        assert_bitwise(full[:, 3:3 + n], ref, "page-locked destination, analytic")
        assert (full[:, :3] == -7.0).all() and (full[:, 3 + n:] == -7.0).all()
        assert np.abs(pr - fx["probs"]).max() < 1e-13
        pl.set_option(_lib.OPT_ANALYTIC_KEEP_ZEROS, 1)
        full[:, 3:3 + n] = np.nan
        for _ in range(2):      # same destination, same request, twice: every entry must still be there
            pl.fill_dprobs(out=full, param_idx=cols, dest_idx=np.arange(n) + 3, mode=_lib.DERIV_ANALYTIC)
            assert_bitwise(full[:, 3:3 + n], ref, "page-locked destination with the keep-zeros option")
    finally:
        _lib.unpin_host_array(full)


@pytest.mark.parametrize("name", ["smq1Q_XYI_L4_depol", "smq1Q_multispam_L2", "smq2Q_XYICNOT_L2_depol", "3q_explicit_L64"])
@pytest.mark.parametrize("mode", ["fd", "analytic"])
def test_device_fills_overwrite_every_requested_entry(name, mode):
    """Fresh device memory is zero, and structural zeros are a third of a GST Jacobian: an entry a kernel forgets to write
    goes unnoticed unless the destination starts out as NaN (that is how a skipped block of the analytic contraction was
    found).  Device destination pre-filled with NaN and a sentinel beyond the requested columns, every mode."""
    from pygsti_amd import _lib
    fx = load_fixture(name)
    pl = plan_from_fixture(fx)
    cols = fx["dprobs_cols"]; nE, n = int(fx["nE"]), len(cols)
    if mode == "analytic" and int(fx["D"]) == 64 and n > 4096:
        cols = cols[:4096]; n = len(cols)
    ld = n + 2
    d_J = pl.device_malloc(nE * ld * 8)
    fill = np.full((nE, ld), np.nan); fill[:, n:] = -7.0
    pl.memcpy_h2d(d_J, fill)
    pl.fill_dprobs_dev(d_J, ld, cols, None, float(fx["derivative_eps"]), None, _lib.DERIV_FD if mode == "fd" else _lib.DERIV_ANALYTIC)
    pl.sync()
    J = pl.memcpy_d2h(np.empty((nE, ld)), d_J)
    assert np.isfinite(J[:, :n]).all(), "entries left unwritten: %d" % int((~np.isfinite(J[:, :n])).sum())
    assert (J[:, n:] == -7.0).all()
    if mode == "fd":
        assert_bitwise(J[:, :n], fx["dprobs_map"][:, :n], "FD into a NaN-prefilled device array " + name)
    pl.device_free(d_J)


@pytest.mark.parametrize("name", FIXTURES)
def test_device_probs_overwrite_every_element(name):
    fx = load_fixture(name)
    pl = plan_from_fixture(fx)
    nE = int(fx["nE"])
    d_p = pl.device_malloc((nE + 3) * 8)
    fill = np.full(nE + 3, np.nan); fill[nE:] = -7.0
    pl.memcpy_h2d(d_p, fill)
    pl.fill_probs_dev(d_p); pl.sync()
    p = pl.memcpy_d2h(np.empty(nE + 3), d_p)
    assert_bitwise(p[:nE], fx["probs"], "probabilities into a NaN-prefilled device array " + name)
    assert (p[nE:] == -7.0).all()
    pl.device_free(d_p)


@pytest.mark.parametrize("name", ["smq1Q_XYI_L4_depol", "smq1Q_multispam_L2", "3q_explicit_L64"])
def test_hprobs_fd_bitwise_vs_reference(name):
    fx = load_fixture(name)
    pl = plan_from_fixture(fx)
    H = pl.fill_hprobs(idx1=fx['hprobs_rows'], idx2=fx['hprobs_cols'], eps=float(fx['hessian_eps']))
    assert_bitwise(H, fx['hprobs_map'], "hprobs")


def test_hprobs_same_row_and_same_element(oracle_built):
    """Blocks that overlap (same element, same row) exercise the merged-special-row path."""
    O = oracle_built
    fx = load_fixture("smq1Q_XYI_L4_kick")
    pl = plan_from_fixture(fx)
    idx = np.array([0, 1, 4, 5, 12, 13, 14, 16, 17, 28, 29, 44, 59])
    H = pl.fill_hprobs(idx1=idx, idx2=idx, eps=1e-5)
    Ho = O.from_fixture(fx).hprobs(idx, idx, eps=1e-5)
    assert_bitwise(H, Ho, "hprobs overlap")


def test_2q_hprobs_block_vs_oracle(oracle_built):
    O = oracle_built
    fx = load_fixture("smq2Q_XYICNOT_L2_depol")
    pl = plan_from_fixture(fx)
    i1 = np.array([3, 20, 80, 81, 97, 336, 600, 1615])
    i2 = np.concatenate([np.arange(0, 20), np.arange(80, 150), np.arange(1500, 1530)])
    H = pl.fill_hprobs(idx1=i1, idx2=i2, eps=1e-5)
    Ho = O.from_fixture(fx).hprobs(i1, i2, eps=1e-5)
    assert_bitwise(H, Ho, "2Q hprobs")


def test_random_model_vs_oracle(oracle_built):
    """Seeded random dense model (no structure, no zeros) on the 2Q L<=2 plan: full Jacobian."""
    O = oracle_built
    fx = load_fixture("smq2Q_XYICNOT_L2_depol")
    rng = np.random.default_rng(11)
    fx = dict(fx)
    fx['gates'] = fx['gates'] + 0.05 * rng.standard_normal(fx['gates'].shape)
    fx['rhos'] = fx['rhos'] + 0.05 * rng.standard_normal(fx['rhos'].shape)
    fx['effects'] = fx['effects'] + 0.05 * rng.standard_normal(fx['effects'].shape)
    pl = plan_from_fixture(fx)
    orc = O.from_fixture(fx)
    assert_bitwise(pl.fill_probs(), orc.probs(), "random probs")
    cols = np.arange(int(fx['nP']))
    assert_bitwise(pl.fill_dprobs(param_idx=cols), orc.dprobs(cols), "random dprobs")


def test_kind_none_columns_are_exact_zero():
    fx = load_fixture("smq2Q_XYICNOT_L1024_deep")   # the idle gate and Gypi2:0 never occur in this atom
    pl = plan_from_fixture(fx)
    none_cols = np.flatnonzero(fx['pkind'] == -1)[:70]
    assert len(none_cols) == 70
    J = pl.fill_dprobs(param_idx=none_cols)
    assert (J == 0).all() and not np.signbit(J).any()


def test_stats_report_device_time():
    fx = load_fixture("smq1Q_XYI_L128_depol")
    pl = plan_from_fixture(fx, timing=1)         # (launch-bound plans record no timing events unless asked: gst_options.timing)
    pl.fill_dprobs()
    st = pl.stats()
    assert st['last_kernel_ms'] > 0 and st['last_total_ms'] >= st['last_kernel_ms']
    quiet = plan_from_fixture(fx)
    quiet.fill_dprobs()
    assert quiet.stats()['last_kernel_ms'] == 0
