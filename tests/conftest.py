import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


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
