"""Plan compiler (host C++ inside libgstfwd.so) checked WITHOUT a GPU: the walk programs it emits are
interpreted in numpy with the reference's arithmetic order and must reproduce the reference's
probabilities bit for bit, for every slot budget / task granularity."""
import numpy as np
import pytest

from conftest import load_fixture, assert_bitwise, plan_from_fixture
from _interp import run_programs
from pygsti_amd import _lib

FIXTURES = ["smq1Q_XYI_L4_depol", "smq1Q_XYI_L128_depol", "smq2Q_XYICNOT_L2_depol", "smq2Q_XYICNOT_L1024_deep"]


def make(fx, **kw):
    return _lib.Plan.from_table(fx['D'], len(fx['gates']), len(fx['rhos']), len(fx['effects']), fx['nE'],
                                fx['cache_size'], fx['t_dest'], fx['t_start'], fx['t_cache'], fx['t_rho'],
                                fx['row_ptr'], fx['gate_idx'], fx['eff_ptr'], fx['eff_label'], fx['eff_dest'], **kw)


@pytest.mark.parametrize("name", FIXTURES)
@pytest.mark.parametrize("max_slots,target_tasks", [(0, 0), (1, 3), (2, 16), (8, 1000)])
def test_programs_reproduce_reference_probs(name, max_slots, target_tasks):
    fx = load_fixture(name)
    pl = make(fx, max_slots=max_slots, target_tasks=target_tasks)
    st = pl.stats()
    words, off = pl.program()
    out, written, s2 = run_programs(words, off, fx['gates'], fx['rhos'], fx['effects'], fx['eff_ptr'],
                                    fx['eff_label'], fx['eff_dest'], int(fx['nE']))
    assert (written == 1).all(), "every element is produced exactly once"
    assert_bitwise(out, fx['probs'], "interpreted program")
    assert s2['applies'] == st['applies_per_pass']
    assert s2['max_slot'] == st['max_slots'] <= (max_slots or 64)
    assert st['trie_nodes'] - len(fx['rhos']) <= st['applies_per_pass']       # at least one apply per trie node
    assert st['applies_per_pass'] <= st['sum_depth'] + 1                        # never worse than no sharing
    assert st['sum_depth'] == int(np.diff(fx['circ_ptr']).sum())


def test_from_circuits_equals_from_table():
    fx = load_fixture("smq2Q_XYICNOT_L2_depol")
    nO = len(fx['outcome_names'])
    n = len(fx['circ_ptr']) - 1
    # circuits in caller order, elements circuit-major; effect index = the fixture's label order per element
    lab = fx['eff_label'].reshape(n, nO)   # rows are indexed by expanded circuit == circuit here
    pc = _lib.Plan.from_circuits(fx['D'], len(fx['gates']), 1, len(fx['effects']), n * nO, np.zeros(n, np.int32),
                                 fx['circ_ptr'], fx['circ_gates'], np.arange(n + 1) * nO, lab.ravel(),
                                 np.arange(n * nO, dtype=np.int32))
    w, off = pc.program()
    out, written, _ = run_programs(w, off, fx['gates'], fx['rhos'], fx['effects'], np.arange(n + 1) * nO,
                                   lab.ravel(), np.arange(n * nO), n * nO)
    assert_bitwise(out, fx['probs'], "from_circuits")


def test_ragged_and_edge_inputs():
    D = 4
    I = np.eye(4)
    gates = np.array([I, I * 0.5])
    # empty circuit, duplicates, a circuit that is a prefix of another, different rhos
    circs = [[], [0], [0], [0, 1], [0, 1, 1], [1], []]
    rho = np.array([0, 0, 0, 0, 0, 1, 1], np.int32)
    ptr = np.zeros(len(circs) + 1, np.int64); ptr[1:] = np.cumsum([len(c) for c in circs])
    g = np.array([x for c in circs for x in c], np.int32)
    n = len(circs)
    pl = _lib.Plan.from_circuits(D, 2, 2, 1, n, rho, ptr, g, np.arange(n + 1), np.zeros(n, np.int32), np.arange(n, dtype=np.int32))
    w, off = pl.program()
    rhos = np.array([[1., 2, 3, 4], [5., 6, 7, 8]]); eff = np.array([[1., 1, 1, 1]])
    out, written, _ = run_programs(w, off, gates, rhos, eff, np.arange(n + 1), np.zeros(n, np.int32), np.arange(n), n)
    assert (written == 1).all()
    assert out.tolist() == [10.0, 10.0, 10.0, 5.0, 2.5, 13.0, 26.0]
    # zero circuits
    pl0 = _lib.Plan.from_circuits(D, 2, 2, 1, 0, np.zeros(0, np.int32), np.zeros(1, np.int64), np.zeros(0, np.int32),
                                  np.zeros(1, np.int64), np.zeros(0, np.int32), np.zeros(0, np.int32))
    assert pl0.stats()['n_tasks'] == 0


def test_bad_descriptions_are_rejected_not_crashed():
    fx = load_fixture("smq1Q_XYI_L4_depol")
    bad = dict(fx); bad['gate_idx'] = fx['gate_idx'].copy(); bad['gate_idx'][3] = 99
    with pytest.raises(ValueError):
        make(bad)
    bad = dict(fx); bad['eff_dest'] = fx['eff_dest'].copy(); bad['eff_dest'][1] = bad['eff_dest'][0]
    with pytest.raises(ValueError):
        make(bad)
    bad = dict(fx); bad['t_start'] = fx['t_start'].copy(); bad['t_start'][0] = 5     # cache slot not yet written
    with pytest.raises(ValueError):
        make(bad)
    bad = dict(fx); bad['D'] = 65                  # (2 .. 64 run, padded to 4 / 16 / 64 where needed: tests/test_general_dimension.py)
    with pytest.raises(_lib.GstError):
        make(bad)


@pytest.mark.parametrize("name", FIXTURES)
def test_state_graph_reproduces_every_circuit(name):
    """The state-id graph the analytic mode walks backwards: from each circuit's final state id the parent chain
    spells the circuit's gates in reverse and ends in its state preparation; every NODE marker id of the programs
    is covered, and replayed states carry the same id (checked bitwise inside the interpreter)."""
    from oracle import oracle as O
    fx = load_fixture(name)
    pl = make(fx, max_slots=1)
    par, sym, leaf = pl.state_graph()
    full, rho = O.expand_table_circuits(fx)
    for c in range(len(full)):
        i = leaf[c]; s = []
        while par[i] >= 0:
            s.append(int(sym[i])); i = par[i]
        assert s[::-1] == [int(g) for g in full[c]] and sym[i] == rho[c]
    words, off = pl.program()
    ids = (words[(words >> 28) == 6] & 0x0FFFFFFF)
    assert ids.max() < len(par)
    out, written, st = run_programs(words, off, fx['gates'], fx['rhos'], fx['effects'], fx['eff_ptr'],
                                    fx['eff_label'], fx['eff_dest'], int(fx['nE']))
    assert set(leaf.tolist()) <= set(st['node_states'].keys())


def test_fd_queue_packing_and_handovers():
    """gst_get_fd_queues (host side): longest-first packing of (task, 64 columns) pairs into per-SIMD queues, then
    hand-overs -- walks of the fullest queues cut in two, the second part given to the emptiest -- narrow the spread of
    the estimated loads; cutting every walk (handover = 2, the tests' setting) cuts more."""
    fx = load_fixture("smq2Q_XYICNOT_L1024_deep")
    from conftest import plan_from_fixture
    pl = plan_from_fixture(fx, target_tasks=0)
    cols = np.asarray(fx["dprobs_cols"], np.int64)
    l0, n_pairs, h0 = pl.fd_queues(cols, n_queues=8, handover=0)
    l1, _, h1 = pl.fd_queues(cols, n_queues=8, handover=1)
    l2, _, h2 = pl.fd_queues(cols, n_queues=8, handover=2)
    assert h0 == 0 and n_pairs > 0 and l0.sum() > 0
    assert l1.max() <= l0.max() and (l1.max() - l1.min()) <= (l0.max() - l0.min())
    assert h2 >= h1 >= 0
    assert l1.sum() >= l0.sum()                       # every hand-over adds its bookkeeping cost to the estimate


@pytest.mark.parametrize("name", ["smq1Q_XYI_L4_depol", "smq1Q_multispam_L2", "smq2Q_XYICNOT_L2_depol"])
def test_dirty_programs_reproduce_the_perturbed_walk(name):
    """gst_get_dirty_programs (finite differences over whole-object perturbations, gst_set_lindblad): for every task and
    object class the compiled fragment -- entered from the base pass's cached state at the object's first application, or
    from the perturbed preparation -- must give, for the circuits it emits, BIT FOR BIT the probabilities of the full walk
    with the perturbed object, and every circuit it does not emit must be untouched by the perturbation.  Numpy
    interpreter on the CPU; the device kernel that runs these programs is tests/test_gpu_lindblad.py's subject."""
    from _interp import run_programs, run_dirty_program
    fx = load_fixture(name)
    pl = make(fx)
    words, off = pl.program()
    dwords, doff, nC = pl.dirty_programs()
    G, R, E = fx["gates"], fx["rhos"], fx["effects"]
    nG, nR = len(G), len(R)
    assert nC == nG + nR and len(doff) == (len(off) - 1) * nC + 1
    nE = int(fx["nE"])
    base, _, st = run_programs(words, off, G, R, E, fx["eff_ptr"], fx["eff_label"], fx["eff_dest"], nE)
    cache = st["node_states"]
    rng = np.random.default_rng(3)
    n_tasks = len(off) - 1
    checked_gate = checked_rho = 0
    for cls in range(nC):
        Gp, Rp = G.copy(), R.copy()
        if cls < nG:
            Gp[cls] = G[cls] + 1e-3 * rng.standard_normal(G[cls].shape)          # the WHOLE object moves
        else:
            Rp[cls - nG] = R[cls - nG] + 1e-3 * rng.standard_normal(R[cls - nG].shape)
        full, _, _ = run_programs(words, off, Gp, Rp, E, fx["eff_ptr"], fx["eff_label"], fx["eff_dest"], nE)
        got = base.copy()
        for t in range(n_tasks):
            a, b = int(doff[t * nC + cls]), int(doff[t * nC + cls + 1])
            if b == a:
                continue
            emitted = run_dirty_program(dwords[a:b], Gp, Rp, E, fx["eff_ptr"], fx["eff_label"], fx["eff_dest"], cache, got)
            assert len(emitted) > 0 and len(set(emitted)) == len(emitted)
            if cls < nG: checked_gate += 1
            else: checked_rho += 1
        assert np.array_equal(got.view(np.uint64), full.view(np.uint64)), "class %d" % cls
        assert (full != base).any()
    assert checked_gate > 0 and checked_rho > 0


@pytest.mark.parametrize("name", ["smq1Q_XYI_L4_depol", "smq2Q_XYICNOT_L2_depol"])
def test_fd_work_counts_what_the_kernel_executes(name):
    """gst_get_fd_work (bench.py's `roofline.executed`): the plan's programs walked with the FD kernel's clean/dirty rule --
    checked here against a direct per-wavefront Python walk of the same programs (one wavefront = 64 consecutive lanes of
    the packed columns: preparation / effect parameters first, then each gate's on wavefronts of its own)."""
    from pygsti_amd import _lib
    fx = load_fixture(name)
    pl = plan_from_fixture(fx)
    nP, D = int(fx["nP"]), int(fx["D"])
    got = pl.fd_work(np.arange(nP))
    words, off = pl.program()
    # the lane packing of gst_fill_fd.cpp::pack_lanes for the `full` element map
    spam = [p for p in range(nP) if fx["pkind"][p] != 0]
    waves = [spam[i:i + 64] for i in range(0, len(spam), 64)]
    for g in range(len(fx["gates"])):
        mine = [p for p in range(nP) if fx["pkind"][p] == 0 and fx["pobj"][p] == g]
        if len(mine) >= 32 or not waves or len(waves[-1]) == 64:
            waves += [mine[i:i + 64] for i in range(0, len(mine), 64)]
        else:
            room = 64 - len(waves[-1]); waves[-1] = waves[-1] + mine[:room]
            waves += [mine[i:i + 64] for i in range(room, len(mine), 64)]
    assert got["n_waves"] == len(waves) and got["n_tasks"] == len(off) - 1
    ex_w = ex_c = dots_w = dots_c = sched = 0
    n_out = np.diff(fx["eff_ptr"])
    for cols in waves:
        gates = {int(fx["pobj"][p]) for p in cols if fx["pkind"][p] == 0}
        rho = any(fx["pkind"][p] == 1 for p in cols); eff = any(fx["pkind"][p] == 2 for p in cols)
        for t in range(len(off) - 1):
            dirty = False; slot = {}
            for w in words[off[t]:off[t + 1]]:
                op, arg = int(w) >> 28, int(w) & 0x0FFFFFFF
                if op == _lib.OP_RHO: dirty = rho
                elif op == _lib.OP_APPLY:
                    dirty = dirty or arg in gates
                    sched += 1
                    if dirty: ex_w += 1; ex_c += len(cols)
                elif op == _lib.OP_SAVE: slot[arg] = dirty
                elif op == _lib.OP_LOAD: dirty = slot[arg]
                elif op == _lib.OP_EMIT and (dirty or eff):
                    dots_w += int(n_out[arg]); dots_c += int(n_out[arg]) * len(cols)
    assert (got["wave_applies_executed"], got["col_applies_executed"]) == (ex_w, ex_c)
    assert (got["wave_dots_executed"], got["col_dots_executed"]) == (dots_w, dots_c)
    assert got["wave_applies_schedule"] == sched == len(waves) * pl.stats()["applies_per_pass"]
    # (1Q: all 60 parameters share ONE wavefront that perturbs the preparation -- nothing is clean; 2Q: 26 wavefronts)
    assert 0 < got["col_applies_executed"] <= got["col_applies_schedule"]
    assert (got["col_applies_executed"] < got["col_applies_schedule"]) == (len(waves) > 1)
