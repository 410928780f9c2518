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
anyd = len(sys.argv) > 3 and sys.argv[3] == "anyd"     # state dimensions that are none of 4 / 16 / 64 (zero-padded behind the C ABI)
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
    if anyd:
        D = int(rng.choice([2, 3, 5, 8, 9, 12, 15, 17, 25, 36, 49, 63]))
        if D > 16:
            n_circ = int(rng.choice([10, 40, 120])); nG = int(rng.integers(1, 5)); max_len = int(rng.choice([4, 20, 60])); max_slots = int(rng.choice([0, 1, 2, 3]))
    persist = str(rng.choice(["0", "2"]))
    extra = os.environ.get("GST_FUZZ_FORCE", "")          # further test-hook keys for every case, e.g. chain_resident=1
    if persist:
        os.environ["GST_TEST_FORCE"] = "persist=%s" % persist + ("," + extra if extra else "")
    elif extra:
        os.environ["GST_TEST_FORCE"] = extra
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
        cols = rng.permutation(nP)[: min(nP, int(rng.choice([1, 30, 200, nP if not (d64 or (anyd and D > 16)) else 700])))]
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
        # --- round 4: the modes without an ordering contract and the bookkeeping around them ---------------------------
        if a["nE"] > 0:
            nE = a["nE"]
            allc = np.arange(nP)
            exact = pl.fill_probs()
            # fast probabilities (level passes forced at D = 16, the matrix-core walk at D = 64): <= 1e-10
            pl.set_option(_lib.OPT_FAST_PROBS, 1); pl.set_option(_lib.OPT_FAST_CHAINS, 2)
            fast = pl.fill_probs()
            # (the random models are not contractive: values grow with depth, so the bars are relative to the largest one)
            pscale = max(1.0, np.abs(exact).max())
            assert np.abs(fast - exact).max() <= 1e-10 * pscale, "fast probs %g (scale %g)" % (np.abs(fast - exact).max(), pscale)
            Jl = pl.fill_dprobs(param_idx=cols, mode=_lib.DERIV_ANALYTIC)
            assert np.abs(Jl - Jo).max() <= 1e-8 * max(1.0, np.abs(Jo).max()), "analytic dprobs through the level / matrix-core passes"
            pl.set_option(_lib.OPT_FAST_PROBS, 0); pl.set_option(_lib.OPT_FAST_CHAINS, 1)
            # resident zeros on tracked memory: repeated exact fills, a finite-difference fill and a new model in between
            if nP * nE <= 4_000_000:
                Jfull = pl.fill_dprobs(param_idx=allc, mode=_lib.DERIV_ANALYTIC)
                d = pl.device_malloc(nE * nP * 8, tracked=True)
                pl.memcpy_h2d(d, np.full(nE * nP, np.nan))
                for rep in range(3):
                    pl.fill_dprobs_dev(d, nP, allc, None, 1e-7, None, _lib.DERIV_ANALYTIC); pl.sync()
                    assert np.array_equal(pl.memcpy_d2h(np.empty((nE, nP)), d), Jfull), "repeated exact fill %d" % rep
                pl.fill_dprobs_dev(d, nP, allc, None, 1e-7, None, _lib.DERIV_FD); pl.sync()
                pl.fill_dprobs_dev(d, nP, allc, None, 1e-7, None, _lib.DERIV_ANALYTIC); pl.sync()
                assert np.array_equal(pl.memcpy_d2h(np.empty((nE, nP)), d), Jfull), "exact fill after an FD fill into the same memory"
                g2 = mdl["gates"] + 1e-3 * rng.standard_normal(mdl["gates"].shape)
                pl.set_model(g2, mdl["rhos"], mdl["effects"])
                J2 = pl.fill_dprobs(param_idx=allc, mode=_lib.DERIV_ANALYTIC)
                pl.fill_dprobs_dev(d, nP, allc, None, 1e-7, None, _lib.DERIV_ANALYTIC); pl.sync()
                assert np.array_equal(pl.memcpy_d2h(np.empty((nE, nP)), d), J2), "exact fill (resident zeros) after a model change"
                pl.set_model(mdl["gates"], mdl["rhos"], mdl["effects"])
                pl.device_free(d)
                # a random general parameterisation over the element Jacobian (chain-rule products of both tilings)
                if D <= 16:
                    # the whole element Jacobian [rhos | effects | gates] through an identity parameter map
                    nR_, nEl_, nG_ = mdl["rhos"].shape[0], mdl["effects"].shape[0], mdl["gates"].shape[0]
                    fk = np.concatenate([np.full(nR_ * D, 1), np.full(nEl_ * D, 2), np.full(nG_ * D * D, 0)]).astype(np.int32)
                    fo = np.concatenate([np.repeat(np.arange(nR_), D), np.repeat(np.arange(nEl_), D), np.repeat(np.arange(nG_), D * D)]).astype(np.int32)
                    fe = np.concatenate([np.tile(np.arange(D), nR_), np.tile(np.arange(D), nEl_), np.tile(np.arange(D * D), nG_)]).astype(np.int32)
                    pl.set_param_map(fk, fo, fe)
                    Jall = pl.fill_dprobs(param_idx=np.arange(len(fk)), mode=_lib.DERIV_ANALYTIC)
                    objs, nQ, exp_cols, exp_abs = [], 0, [], []
                    for kind, n_obj in ((0, nG_), (1, nR_), (2, nEl_)):
                        for o in range(n_obj):
                            if rng.random() < 0.2:
                                continue                        # an object without parameters
                            Jel = Jall[:, (fk == kind) & (fo == o)]
                            nq = int(rng.choice([1, 7, 45, 97]))
                            W = rng.standard_normal((Jel.shape[1], nq))
                            objs.append((kind, int(o), nQ + np.arange(nq), W)); exp_cols.append(Jel @ W); exp_abs.append(np.abs(Jel) @ np.abs(W)); nQ += nq
                    if not objs:
                        objs.append((1, 0, np.arange(3), np.zeros((D, 3)))); exp_cols.append(np.zeros((nE, 3))); exp_abs.append(np.zeros((nE, 3))); nQ = 3
                    pl.set_derivs(nQ, objs)
                    Jg = pl.fill_dprobs(param_idx=np.arange(nQ), mode=_lib.DERIV_ANALYTIC)
                    Je = np.concatenate(exp_cols, axis=1)
                    cscale = max(1.0, np.concatenate(exp_abs, axis=1).max())
                    assert np.abs(Jg - Je).max() <= 1e-12 * cscale, "chain rule %g (scale %g)" % (np.abs(Jg - Je).max(), cscale)
                    Jg2 = pl.fill_dprobs(param_idx=np.arange(nQ), mode=_lib.DERIV_ANALYTIC)
                    assert np.array_equal(Jg, Jg2), "chain rule (repeat)"
                    pl.set_derivs(nP, [])
                    pl.set_param_map(mdl["pkind"], mdl["pobj"], mdl["pelem"])
        pl.close()
    except Exception as e:           # noqa
        bad += 1
        print("FAIL", tag, "->", repr(e)[:300], flush=True)
print("%d cases, %d failures, %.1f s" % (n_cases, bad, time.time() - t0))
sys.exit(1 if bad else 0)
