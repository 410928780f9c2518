import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# Diagnostics for a process that dies through abort(): the library prints the native stack (gst_abi.cpp), and glibc sends its
# own message ("free(): invalid size", ...) to stderr instead of the controlling terminal, where a test log never sees it.
os.environ.setdefault("LIBC_FATAL_STDERR_", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # pytest captures fd 2 while a test runs, and what a dying process wrote into the capture is lost: the library gets a
    # duplicate of the REAL stderr (this hook runs before the per-test capture, as pytest's own faulthandler plugin relies on)
    if "GST_ABORT_BACKTRACE" not in os.environ:
        try:
            os.environ["GST_ABORT_BACKTRACE"] = "%d:%d" % (os.dup(2), os.getpid())
        except OSError:
            os.environ["GST_ABORT_BACKTRACE"] = "1"


def page_locked_candidate(shape, fill):
    """A float64 array with pages of its own (anonymous mmap: page-aligned, shares no page with the heap) -- what a test hands
    to _lib.pin_host_array.  Registering ranges INSIDE the brk heap is legal but is what a rare GPU memory fault on a
    host-heap address was traced to; the library's own callers only page-lock mmap-backed arrays."""
    import mmap
    count = int(np.prod(shape))
    mm = mmap.mmap(-1, max(count * 8, mmap.PAGESIZE))
    a = np.frombuffer(mm, dtype=np.float64, count=count).reshape(shape)
    a[...] = fill
    return a


def load_fixture(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def assert_bitwise(a, b, what=""):
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    assert a.shape == b.shape, "%s: shape %s vs %s" % (what, a.shape, b.shape)
    neq = bits(a) != bits(b)
    if neq.any():
        idx = np.argwhere(neq.reshape(a.shape))[0]
        raise AssertionError("%s: %d of %d values differ bitwise; first at %s: %r vs %r (max abs diff %.3e)" % (
            what, int(neq.sum()), a.size, tuple(idx), a[tuple(idx)], b[tuple(idx)], np.nanmax(np.abs(a - b))))


@pytest.fixture(scope="session")
def oracle_built():
    from oracle import oracle as O
    O.build(ref=True)
    return O


def plan_from_fixture(fx, **kw):
    from pygsti_amd import _lib
    pl = _lib.Plan.from_table(fx['D'], len(fx['gates']), len(fx['rhos']), len(fx['effects']), fx['nE'],
                              fx['cache_size'], fx['t_dest'], fx['t_start'], fx['t_cache'], fx['t_rho'],
                              fx['row_ptr'], fx['gate_idx'], fx['eff_ptr'], fx['eff_label'], fx['eff_dest'], **kw)
    pl.set_model(fx['gates'], fx['rhos'], fx['effects'])
    pl.set_param_map(fx['pkind'], fx['pobj'], fx['pelem'])
    return pl


def matrix_rows_by_circuit(fx):
    """Round-4 fixtures store the Matrix simulator's vectors ordered [circuit][outcome name sorted]
    (`matrix_by_circuit_*`, `matrix_outcome_names`): index array that brings them into the Map layout's element order."""
    names = [str(n) for n in fx["outcome_names"]]
    mn = [str(n) for n in fx["matrix_outcome_names"]]
    return np.array([int(fx["el_circuit"][k]) * len(mn) + mn.index(names[int(fx["el_outcome"][k])])
                     for k in range(int(fx["nE"]))], np.int64)


def element_depth(fx):
    """Gate count of the circuit every element belongs to."""
    return np.diff(fx["circ_ptr"])[fx["el_circuit"]]


def design_checker(oracle_module, pack, model, circuits, layout, kind=None):
    """The CPU checker for a WHOLE design of the host mirror (bench.py's workload): the reference-format prefix table of
    `circuits` (oracle/prefix_table.py, array-equal to the reference's own table on every design fixture) with the model's
    arrays and the `full` parameter map, walked by the reference's own C++ reps when oracle/_ref is built.  Element order =
    the 1-atom layout's (circuit-major, outcomes in model.effect_labels order)."""
    from oracle import prefix_table as PT
    lookup = {l: i for i, l in enumerate(model.operations.keys())}
    ptr = np.zeros(len(circuits) + 1, np.int64)
    ptr[1:] = np.cumsum([len(c) for c in circuits])
    gates = np.fromiter((lookup[g] for c in circuits for g in c), np.int32, count=int(ptr[-1]))
    tbl = PT.build_table(ptr, gates, len(model.effect_labels))
    tbl["D"] = model.dim
    G, R, E = layout.model_arrays(model)
    pk, po, pe = layout.param_map(model)
    mdl = dict(gates=G, rhos=R, effects=E, pkind=pk, pobj=po, pelem=pe)
    if kind is None:
        kind = "reference" if os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libgst_ref.so")) else "port"
    return oracle_module.Oracle(tbl, mdl, kind)


def force(monkeypatch, **kw):
    """Set keys of GST_TEST_FORCE (the library's one test hook: launch-form selectors read when a plan is created;
    include/gstfwd.h); value None removes a key."""
    cur = dict(item.split("=", 1) for item in os.environ.get("GST_TEST_FORCE", "").split(",") if item)
    for k, v in kw.items():
        if v is None:
            cur.pop(k, None)
        else:
            cur[k] = str(v)
    if cur:
        monkeypatch.setenv("GST_TEST_FORCE", ",".join("%s=%s" % kv for kv in cur.items()))
    else:
        monkeypatch.delenv("GST_TEST_FORCE", raising=False)
