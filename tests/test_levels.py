"""Log-depth level programs (csrc/gst_levels.cpp; SURVEY 8(f) row f2, second half) checked WITHOUT a GPU: the programs the
host scheduler emits -- periodic (germ-power) paths of the tries evaluated by matrix squaring and doubling, everything
else level by level -- are interpreted in numpy and must reproduce EVERY state of the trie that the sequential walk
produces (forward plan: F = G_k ... G_1 rho; reversed plan: B = G_{k+1}^T ... G_n^T E for every effect), each written
exactly once, in far fewer dependent stages than the walk has steps."""
import numpy as np
import pytest

from conftest import load_fixture
from test_plan_compiler import make

KIND_ROWS, KIND_MAT, IDENT = 0, 1, -1


def interpret(lp, bmats, starts, nv, D=16):
    """Run a level program.  bmats[g] = the gate in ROW form (out_row = in_row @ bmats[g]); starts[s] = [nv][D] start vectors.
    Returns (cache {id: [D][nv]}, stages per task, scratch matrices used)."""
    words, ids, off = lp["words"], lp["ids"], lp["task_off"]
    cache, n_stages, writes = {}, [], {}
    for t in range(len(off) - 1):
        pc = int(off[t])
        ns = int(words[pc]); pc += 1
        n_stages.append(ns)
        mats = {}

        def matrix(ref):
            if ref == IDENT:
                return np.eye(D)
            return bmats[ref] if ref >= 0 else mats[-(ref + 2)]
        for s in range(ns):
            nt = int(words[pc]); pc += 1
            produced = []                                     # (a stage's tiles read only what EARLIER stages wrote)
            for k in range(nt):
                w0, mref, a, b = (int(x) for x in words[pc + 4 * k: pc + 4 * k + 4])
                kind, n_nodes = w0 & 255, w0 >> 8
                M = matrix(mref)
                if kind == KIND_MAT:
                    assert n_nodes == 16 and b not in mats and all(p[0] != ("m", b) for p in produced)
                    produced.append((("m", b), matrix(a) @ M))
                else:
                    assert kind == KIND_ROWS and 1 <= n_nodes * nv <= 16
                    for q in range(n_nodes):
                        src, dst = int(ids[a + q]), int(ids[b + q])
                        X = starts[-(src + 1)] if src < 0 else cache[src].T          # [nv][D] rows
                        produced.append((("s", dst), (X @ M).T))
            for key, val in produced:
                if key[0] == "m":
                    mats[key[1]] = val
                else:
                    assert key[1] not in cache, "state %d written twice" % key[1]
                    cache[key[1]] = val
            pc += 4 * nt
        assert pc == off[t + 1]
        assert len(mats) <= lp["max_mats"]
    return cache, n_stages


def sequential_states(node_parent, node_sym, produced_ids, bmats, starts):
    """The same states by following the state graph (parents have smaller ids)."""
    out = {}
    for i in sorted(produced_ids):
        p = int(node_parent[i])
        X = starts[int(node_sym[i])] if p < 0 else out[p].T @ bmats[int(node_sym[i])]
        out[i] = X.T
    return out


@pytest.mark.parametrize("name", ["smq2Q_XYICNOT_L1024_deep", "smq2Q_XYICNOT_L2_depol", "smq2Q_XYICNOT_L1024_CPTPLND_deep"])
@pytest.mark.parametrize("which", [0, 1])
def test_level_programs_reproduce_every_state(name, which):
    fx = load_fixture(name)
    pl = make(fx)
    lp = pl.level_program(which)
    assert lp["usable"], lp
    G, R, E = fx["gates"], fx["rhos"], fx["effects"]
    if which == 0:
        nv, bmats, starts = 1, np.transpose(G, (0, 2, 1)), R[:, None, :]              # forward: out_row = in_row @ G^T
    else:
        nv, bmats, starts = len(E), G, E[None, :, :]                                 # backward: B'^T = B^T G; start = all effects
    assert lp["nv"] == nv
    cache, n_stages = interpret(lp, bmats, starts, nv)
    assert len(cache) == lp["n_states"] - lp["n_tasks"], "every state (all ids but the tasks' virtual roots) is produced"
    seq = sequential_states(lp["node_parent"], lp["node_sym"], cache.keys(), bmats, starts)
    err = max(np.abs(cache[i] - seq[i]).max() for i in cache)
    assert err < 1e-12, err
    assert max(n_stages) == lp["max_stages"] and sum(n_stages) == lp["n_stages"]
    if "L1024" in name:
        # depth-1,030 germ-power families: a handful of stages per task instead of ~1,000 dependent steps
        assert lp["worthwhile"] and lp["n_chains"] >= 9 and lp["max_stages"] <= 48, lp
        assert lp["chain_nodes"] > 0.8 * (lp["n_states"] - lp["n_tasks"])
        assert 20 * lp["n_stages"] < lp["sum_task_depth"]
    else:
        assert not lp["worthwhile"]            # (L <= 2: nothing periodic; the sequential walk stays)
        assert lp["n_chains"] == 0


def test_level_program_of_the_bench_design_is_shallow():
    """The benchmarked workload (2Q L<=1024 full design, 136,275 circuits): every task's ~1,150-step walk becomes at most a
    few dozen stages; the scheduler itself is quick."""
    import time
    from pygsti_amd import modelpacks as MP
    from pygsti_amd.layout import HipCOPALayout
    pack = MP.smq2Q_XYICNOT
    model = pack.target_model().depolarize(0.01, 0.01)
    layout = HipCOPALayout(pack.create_gst_circuits(1024, lite=False), model, num_atoms=1, devices=[0], rank=0, size=1)
    plan = layout.atoms[0].plan()
    for which in (0, 1):
        t0 = time.perf_counter()
        lp = plan.level_program(which)
        dt = time.perf_counter() - t0
        assert lp["usable"] and lp["worthwhile"], {k: v for k, v in lp.items() if not hasattr(v, "shape")}
        assert lp["max_stages"] <= 64 and lp["n_chains"] >= 1000
        assert 10 * lp["n_stages"] < lp["sum_task_depth"]
        assert dt < 20.0, dt
        print("which=%d: %.2f s; %s" % (which, dt, {k: v for k, v in lp.items() if not hasattr(v, "shape")}))


@pytest.mark.parametrize("name", ["smq2Q_XYICNOT_L1024_deep", "smq2Q_XYICNOT_L2_depol"])
def test_probability_only_program_produces_the_final_states(name):
    """which = 2: only the circuits' final states and what they are computed from (doubling sources, period boundaries,
    the paths to them) -- every final state equals the sequential walk's, far fewer states are formed on deep families."""
    fx = load_fixture(name)
    pl = make(fx)
    full, lp = pl.level_program(0), pl.level_program(2)
    assert lp["usable"]
    G, R = fx["gates"], fx["rhos"]
    bmats, starts = np.transpose(G, (0, 2, 1)), R[:, None, :]
    cache, n_stages = interpret(lp, bmats, starts, 1)
    assert len(cache) == lp["n_produced"] <= full["n_produced"]
    par, sym = lp["node_parent"], lp["node_sym"]
    # the final state of every circuit: the state graph's leaves of the fixture's circuits, walked sequentially
    _, _, leaf = pl_state_graph(pl)
    assert set(int(x) for x in leaf) <= set(cache.keys())
    for c in range(0, len(leaf), max(1, len(leaf) // 60)):
        path, i = [], int(leaf[c])
        while i >= 0:
            path.append(i); i = int(par[i])
        X = None
        for i in reversed(path):
            X = starts[int(sym[i])] if par[i] < 0 else X @ bmats[int(sym[i])]
        assert np.abs(cache[int(leaf[c])].T - X).max() < 1e-12
    if "L1024" in name:
        assert lp["n_produced"] < 0.5 * full["n_produced"], (lp["n_produced"], full["n_produced"])
        assert lp["max_stages"] <= full["max_stages"]


def pl_state_graph(pl):
    import ctypes as C
    from pygsti_amd import _lib
    n = C.c_int64(0)
    _lib.check(_lib.lib().gst_get_state_graph(pl._h, None, None, 0, None, 0, C.byref(n)))
    par = np.empty(n.value, np.int32); sym = np.empty(n.value, np.int32)
    nc = pl.stats()["n_circuits"]
    leaf = np.empty(nc, np.int32)
    _lib.check(_lib.lib().gst_get_state_graph(pl._h, par.ctypes.data_as(C.c_void_p), sym.ctypes.data_as(C.c_void_p), n.value,
                                              leaf.ctypes.data_as(C.c_void_p), nc, C.byref(n)))
    return par, sym, leaf


def test_non_periodic_and_small_plans_have_no_level_program_or_decline():
    fx = load_fixture("smq1Q_XYI_L128_depol")           # D = 4
    assert not make(fx).level_program(0)["usable"]
    fx = load_fixture("3q_explicit_matrix")             # D = 64
    assert not make(fx).level_program(0)["usable"]
