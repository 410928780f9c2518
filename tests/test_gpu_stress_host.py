"""Soak of the host-array boundary ("the rare abort"): thousands of fills into caller-owned HEAP arrays of assorted
sizes, alternating finite-difference Jacobians and exact Hessian blocks, while a second thread keeps the allocator busy and
page-locks / releases regions of its own.  The reference's contract for these buffers is "valid for the call only"
(mapforwardsim_calc_densitymx.pyx:170-190: numpy owns them, the C loop borrows the pointer): the library must therefore never
leave the device a reason to touch a caller address after -- or, for pageable memory, during -- the call.  A regression shows
as a dead process ("Memory access fault by GPU ..."), not as a failed assertion; the assertions pin the results bit for bit
so that a wrong staging offset cannot hide either."""
import gc
import threading

import numpy as np
import pytest

from conftest import load_fixture, plan_from_fixture, assert_bitwise, page_locked_candidate

pytestmark = pytest.mark.gpu

N_CALLS = 2400


def _pressure(stop, errors):
    """Allocation pressure + gst_host_register / unregister churn beside the fills (own mmap regions: what callers are told to
    register), plus plain heap garbage of the sizes glibc serves from brk and from mmap."""
    from pygsti_amd import _lib
    rng = np.random.default_rng(7)
    held = []
    try:
        while not stop.is_set():
            kind = int(rng.integers(0, 3))
            if kind == 0:
                a = page_locked_candidate((int(rng.integers(1, 1 << 16)),), 0.0)
                _lib.pin_host_array(a)
                held.append(("pin", a))
            elif kind == 1:
                held.append(("heap", np.empty(int(rng.integers(1, 1 << 18)))))          # brk-sized and mmap-sized
            else:
                held.append(("heap", bytearray(int(rng.integers(1, 1 << 12)))))
            while len(held) > int(rng.integers(2, 24)):
                k, a = held.pop(int(rng.integers(0, len(held))))
                if k == "pin":
                    _lib.unpin_host_array(a)
                del a
    except Exception as e:          # noqa: BLE001 -- reported by the test body
        errors.append(e)
    finally:
        for k, a in held:
            if k == "pin":
                _lib.unpin_host_array(a)


def test_heap_destinations_under_allocation_pressure():
    from pygsti_amd import _lib
    fx1 = load_fixture("smq1Q_XYI_L4_depol")
    fx2 = load_fixture("smq2Q_XYICNOT_L2_depol")
    p1, p2 = plan_from_fixture(fx1), plan_from_fixture(fx2)
    nP1, nP2, nE1, nE2 = int(fx1["nP"]), int(fx2["nP"]), int(fx1["nE"]), int(fx2["nE"])
    J1_ref = p1.fill_dprobs(eps=1e-7)
    assert_bitwise(J1_ref, fx1["dprobs_map"], "reference Jacobian (1Q)")
    J2_ref = p2.fill_dprobs(eps=1e-7, param_idx=np.arange(nP2))
    H1_ref = p1.fill_hprobs(idx1=fx1["hprobs_rows"], idx2=fx1["hprobs_cols"], mode=_lib.DERIV_ANALYTIC)
    H2_ref = p2.fill_hprobs(idx1=fx2["mh0_idx1"], idx2=fx2["mh0_idx2"], mode=_lib.DERIV_ANALYTIC)

    stop, errors = threading.Event(), []
    th = threading.Thread(target=_pressure, args=(stop, errors), daemon=True)
    th.start()
    rng = np.random.default_rng(11)
    garbage = []
    try:
        for it in range(N_CALLS):
            which = it % 4
            if which == 0:      # a window of columns into a wider heap array (leading dimension > columns)
                c0 = int(rng.integers(0, nP1 - 1)); n = int(rng.integers(1, nP1 - c0 + 1))
                ld = n + int(rng.integers(0, 9))
                out = np.full((nE1, ld), -7.0)
                p1.fill_dprobs(out=out[:, :n], param_idx=np.arange(c0, c0 + n), eps=1e-7)
                assert_bitwise(out[:, :n], J1_ref[:, c0:c0 + n], "call %d: 1Q Jacobian window" % it)
                assert (out[:, n:] == -7.0).all(), "call %d: wrote outside its columns" % it
            elif which == 1:    # exact Hessian block, small
                H = p1.fill_hprobs(idx1=fx1["hprobs_rows"], idx2=fx1["hprobs_cols"], mode=_lib.DERIV_ANALYTIC)
                assert_bitwise(H, H1_ref, "call %d: 1Q exact Hessian" % it)
            elif which == 2:    # 2Q Jacobian columns: up to 2.5 MB destinations (mmap-served) and tiny ones (brk-served)
                n = int(rng.choice([1, 3, 16, 64, 200]))
                c0 = int(rng.integers(0, nP2 - n))
                out = np.empty((nE2, n))
                p2.fill_dprobs(out=out, param_idx=np.arange(c0, c0 + n), eps=1e-7)
                assert_bitwise(out, J2_ref[:, c0:c0 + n], "call %d: 2Q Jacobian columns" % it)
            else:
                H = p2.fill_hprobs(idx1=fx2["mh0_idx1"], idx2=fx2["mh0_idx2"], mode=_lib.DERIV_ANALYTIC)
                assert_bitwise(H, H2_ref, "call %d: 2Q exact Hessian" % it)
            garbage.append(np.empty(int(rng.integers(1, 1 << 15))))
            if len(garbage) > 16:
                del garbage[:int(rng.integers(1, 16))]
            if it % 97 == 0:
                gc.collect()
            if it % 300 == 150:     # fresh plans now and then: their buffers come and go with everything else
                p1.close(); p2.close()
                p1, p2 = plan_from_fixture(fx1), plan_from_fixture(fx2)
            assert not errors, errors
    finally:
        stop.set()
        th.join(60)
    assert not errors, errors
    assert not th.is_alive()
