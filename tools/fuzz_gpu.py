"""Development aid: randomized parity sweep on the GPU box -- many random plans (shapes, slot budgets, task targets, both
FD launch forms) through the C ABI against the CPU oracle, bit for bit (probs, FD dprobs, FD-of-FD hprobs), and the
analytic Jacobian against the numpy forward/backward oracle.  Usage: python tools/fuzz_gpu.py [n_cases] [seed0] [d64]
(third argument "d64": 3-qubit shapes, D = 64, through the register-blocked / shared-tile / row kernels)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_gpu_ragged import _random_case
from oracle import oracle as O
from pygsti_amd import _lib

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
d64 = len(sys.argv) > 3 and sys.argv[3] == "d64"
big = len(sys.argv) > 3 and sys.argv[3] == "big"       # thousands of circuits: the default (size-dependent) launch form
t0 = time.time()
bad = 0
for k in range(n_cases):
    rng = np.random.default_rng(seed0 + k)
    D = 4 if rng.random() < 0.35 else 16
    n_circ = int(rng.choice([20, 80, 200, 500]))
    nG = int(rng.integers(1, 8)); nR = int(rng.integers(1, 4)); nEl = int(rng.integers(1, 7))
    max_len = int(rng.choice([6, 40, 150, 400]))
    max_slots = int(rng.integers(0, 5)); target_tasks = int(rng.choice([0, 0, 1, 3, 17, 200]))
    if big:
        D = 16; n_circ = int(rng.choice([1500, 3000])); nG = int(rng.integers(3, 8)); max_len = int(rng.choice([100, 600]))
        persist = ""
    if d64:
        D = 64; n_circ = int(rng.choice([10, 40, 120])); nG = int(rng.integers(1, 5)); nEl = int(rng.integers(1, 9))
        max_len = int(rng.choice([4, 20, 60])); max_slots = int(rng.choice([0, 1, 2, 3, 6, 12]))
    persist = str(rng.choice(["0", "2"]))
    if persist:
        os.environ["GST_TEST_FORCE"] = "persist=%s" % persist
    else:
        os.environ.pop("GST_TEST_FORCE", None)
    a, tbl, mdl, nP = _random_case(D, seed0 + k, n_circ=n_circ, nG=nG, nR=nR, nEl=nEl, max_len=max_len)
    tag = "case %d: D=%d circuits=%d nG=%d nR=%d nEl=%d max_len=%d slots=%d tasks=%d persist=%s" % (
        k, D, n_circ, nG, nR, nEl, max_len, max_slots, target_tasks, persist)
    try:
        pl = _lib.Plan.from_circuits(D, a["nG"], a["nR"], a["nEl"], a["nE"], a["rho"], a["ptr"], a["g"], a["eff_ptr"],
                                     a["eff_label"], a["eff_dest"], max_slots=max_slots, target_tasks=target_tasks)
        pl.set_model(mdl["gates"], mdl["rhos"], mdl["effects"])
        pl.set_param_map(mdl["pkind"], mdl["pobj"], mdl["pelem"])
        orc = O.Oracle(tbl, mdl)
        assert np.array_equal(pl.fill_probs(), orc.probs()), "probs"
        cols = rng.permutation(nP)[: min(nP, int(rng.choice([1, 30, 200, nP if not d64 else 700])))]
        J = pl.fill_dprobs(param_idx=cols, eps=1e-7)
        assert np.array_equal(J, orc.dprobs(cols, eps=1e-7)), "dprobs"
        assert np.array_equal(pl.fill_dprobs(param_idx=cols, eps=1e-7), J), "dprobs (repeat)"
        i1 = rng.permutation(nP)[:5]; i2 = rng.permutation(nP)[:40]
        if a["nE"] > 0:
            H = pl.fill_hprobs(idx1=i1, idx2=i2, eps=1e-5)
            assert np.array_equal(H, orc.hprobs(i1, i2, eps=1e-5)), "hprobs"
        fx = dict(tbl); fx.update(mdl)
        Ja = pl.fill_dprobs(param_idx=cols, mode=_lib.DERIV_ANALYTIC)
        Jo, _ = O.analytic_dprobs(fx, cols)
        assert np.abs(Ja - Jo).max() <= 1e-8 * max(1.0, np.abs(Jo).max()), "analytic dprobs"
        if a["nE"] > 0 and n_circ <= 80 and max_len <= 60:       # exact Hessian block against the numpy derivative-state oracle
            Ha = pl.fill_hprobs(idx1=i1[:3], idx2=i2[:12], mode=_lib.DERIV_ANALYTIC)
            Ho = O.analytic_hprobs(fx, i1[:3], i2[:12])
            assert np.abs(Ha - Ho).max() <= 1e-8 * max(1.0, np.abs(Ho).max()), "analytic hprobs"
        pl.close()
    except Exception as e:           # noqa
        bad += 1
        print("FAIL", tag, "->", repr(e)[:300], flush=True)
print("%d cases, %d failures, %.1f s" % (n_cases, bad, time.time() - t0))
sys.exit(1 if bad else 0)
