"""Measure the BASELINE.json configurations other than the bench line (configs[1] 1Q L<=128, configs[4] 3-qubit D=64)
on one MI355X: probabilities, FD Jacobian, analytic Jacobian (where implemented), one Hessian block.  One JSON line
per configuration on stdout; the committed copy is profiles/r01_configs.json.   python tools/bench_configs.py"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygsti_amd import _lib, modelpacks
from pygsti_amd.layout import HipCOPALayout


F64_PEAK_TFLOPS, HBM_PEAK_GBS = 78.6, 8000.0


def _roof_compute(flops, seconds):
    a = flops / seconds / 1e12
    return {"bound": "mfma", "achieved": a, "peak": F64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": a / F64_PEAK_TFLOPS}


def _roof_hbm(nbytes, seconds):
    a = nbytes / seconds / 1e9
    return {"bound": "hbm", "achieved": a, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": a / HBM_PEAK_GBS}


def timed(fn, plan, reps):
    fn(); plan.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    plan.sync()
    return (time.perf_counter() - t0) / reps


def _releasing(fn):
    """Free what `fn` allocated on the device and close the plans it created: bench.py runs these legs BEFORE it builds
    the 2Q workload and wants the process clean again afterwards."""
    def run(*args, **kw):
        made = []
        orig = _lib.Plan.device_malloc

        def rec(self, nbytes, tracked=False):
            ptr = orig(self, nbytes, tracked)
            made.append((self, ptr))
            return ptr
        _lib.Plan.device_malloc = rec
        try:
            return fn(*args, **kw)
        finally:
            _lib.Plan.device_malloc = orig
            plans = []
            for pl, ptr in made:
                try:
                    pl.device_free(ptr)
                except Exception:
                    pass
                if not any(pl is q for q in plans):
                    plans.append(pl)
            for pl in plans:
                pl.close()
    run.__name__, run.__doc__ = fn.__name__, fn.__doc__
    return run


@_releasing
def one_q():
    pack = modelpacks.smq1Q_XYI
    model = pack.target_model().depolarize(op_noise=0.01, spam_noise=0.01)
    circuits = pack.create_gst_circuits(128)
    layout = HipCOPALayout(circuits, model, num_atoms=1, devices=[0], rank=0, size=1)
    plan = layout.atoms[0].plan()
    plan.set_model(*layout.model_arrays(model)); plan.set_param_map(*layout.param_map(model))
    nE, nP = layout.num_elements, model.num_params
    d_J = plan.device_malloc(nE * nP * 8); d_p = plan.device_malloc(nE * 8)
    pidx = np.arange(nP, dtype=np.int64)
    t_p = timed(lambda: plan.fill_probs_dev(d_p), plan, 200)
    t_fd = timed(lambda: plan.fill_dprobs_dev(d_J, nP, pidx, None, 1e-7, d_p, _lib.DERIV_FD), plan, 200)
    t_an = timed(lambda: plan.fill_dprobs_dev(d_J, nP, pidx, None, 1e-7, d_p, _lib.DERIV_ANALYTIC), plan, 200)
    st = plan.stats()
    # one blocking fill at a time (what a caller that needs the result sees), host arrays included
    J = layout.allocate_local_array("ep"); pr = layout.allocate_local_array("e")     # (what bulk_fill_dprobs(array, layout) is handed: page-locked from 256 KB, on pages of its own)
    def lat(fn, reps=200):
        fn()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        return (time.perf_counter() - t0) / reps
    l_p = lat(lambda: plan.fill_probs(pr))
    l_fd = lat(lambda: plan.fill_dprobs(J, pidx, None, 1e-7, pr, _lib.DERIV_FD))
    l_an = lat(lambda: plan.fill_dprobs(J, pidx, None, 1e-7, pr, _lib.DERIV_ANALYTIC))
    layout.free_local_array(J); layout.free_local_array(pr)
    # one Levenberg-Marquardt evaluation, blocking: the four-call composition against gst_lm_step_dev (one call; a HIP graph from the
    # third call on: a 1Q iteration is seven launches around microseconds of work)
    bufs = [plan.device_malloc(n * 8) for n in (nE, nE, nE, nE, nP * nP, nP)]
    d_c, d_N, d_ls, d_w, d_jtj, d_jtf = bufs
    cnt = np.random.default_rng(3).binomial(1000, np.clip(plan.fill_probs(), 0, 1)).astype(np.float64)
    plan.memcpy_h2d(d_c, cnt); plan.memcpy_h2d(d_N, np.full(nE, 1000.0))
    G_, R_, E_ = layout.model_arrays(model)
    def four_calls():
        plan.set_model(G_, R_, E_)
        plan.fill_dprobs_dev(d_J, nP, pidx, None, 1e-7, d_p, _lib.DERIV_FD)
        plan.objective_rows_dev("logl", d_p, d_c, d_N, nE, d_ls, d_w)
        plan.fill_normal_eqs_dev(d_J, nE, nP, nP, d_w, d_ls, d_jtj, d_jtf)
        plan.sync()
    def one_call():
        plan.set_model(G_, R_, E_)
        plan.lm_step_dev(nP, d_c, d_N, d_J, nP, d_p, d_ls, d_w, d_jtj, d_jtf, "logl")
    l_lm4 = lat(four_calls)
    one_call(); one_call()
    l_lm1 = lat(one_call)
    D = 4
    fl_pass = 2.0 * D * D * st["applies_per_pass"] + 2.0 * D * nE           # flops of one probability pass
    roof = {"probs": _roof_compute(fl_pass, t_p), "dprobs_fd": _roof_compute(nP * fl_pass, t_fd),
            "dprobs_analytic": _roof_hbm(8.0 * nE * nP, t_an),
            "note": "launch / latency bound at this size (a fill is two ~30 us dependent chains): the fractions say how far a 2,240-element problem is from a roof sized for 10^5 circuits"}
    return {"config": "smq1Q_XYI L<=128 (BASELINE configs[1]): %d circuits, nE=%d, nP=%d, D=4" % (len(circuits), nE, nP),
            "roofline": roof,
            "blocking_host_fill_us": {"probs": 1e6 * l_p, "dprobs_fd": 1e6 * l_fd, "dprobs_analytic": 1e6 * l_an},
            "lm_step_blocking_us": {"four_calls": 1e6 * l_lm4, "gst_lm_step_dev_hip_graph": 1e6 * l_lm1,
                                    "note": "model upload + FD Jacobian + objective rows + JtJ + Jtf + sum(terms), results on the device"},
            "probs_us": 1e6 * t_p, "probs_per_s": nE / t_p,
            "dprobs_fd_us": 1e6 * t_fd, "dprobs_fd_el_per_s": nE * nP / t_fd,
            "dprobs_analytic_us": 1e6 * t_an, "dprobs_analytic_el_per_s": nE * nP / t_an,
            "note": "latency-bound: %d tasks, %d gate applications per pass; times include the launch (~10 us each kernel)" % (st["n_tasks"], st["applies_per_pass"])}


@_releasing
def three_q(n_circ=4000, max_len=256, n_cols=4096):
    # (rounds 1-4 measured 400 circuits: 314 tasks on 256 CUs, i.e. chain latency, not throughput; SURVEY's C5 allows more)
    rng = np.random.default_rng(0)
    D, nG, nEl = 64, 10, 8
    gates = np.eye(D)[None] + 0.04 * rng.standard_normal((nG, D, D))
    rhos = np.zeros((1, D)); rhos[0, 0] = 1.0 / np.sqrt(8)
    effects = 0.1 * rng.standard_normal((nEl, D)); effects[:, 0] += 1.0 / np.sqrt(8)
    circs = [rng.integers(0, nG, L) for L in rng.integers(1, max_len + 1, n_circ)]
    ptr = np.zeros(n_circ + 1, np.int64); ptr[1:] = np.cumsum([len(c) for c in circs])
    g = np.concatenate(circs).astype(np.int32)
    nE = n_circ * nEl
    nP = D + nEl * D + nG * D * D
    kind = np.concatenate([np.full(D, 1), np.full(nEl * D, 2), np.full(nG * D * D, 0)]).astype(np.int32)
    obj = np.concatenate([np.zeros(D), np.repeat(np.arange(nEl), D), np.repeat(np.arange(nG), D * D)]).astype(np.int32)
    elem = np.concatenate([np.arange(D), np.tile(np.arange(D), nEl), np.tile(np.arange(D * D), nG)]).astype(np.int32)
    plan = _lib.Plan.from_circuits(D, nG, 1, nEl, nE, np.zeros(n_circ, np.int32), ptr, g, np.arange(n_circ + 1, dtype=np.int64) * nEl,
                                   np.tile(np.arange(nEl, dtype=np.int32), n_circ), np.arange(nE, dtype=np.int32))
    plan.set_model(gates, rhos, effects); plan.set_param_map(kind, obj, elem)
    d_p = plan.device_malloc(nE * 8)
    t_p = timed(lambda: plan.fill_probs_dev(d_p), plan, 20)
    exact_p = plan.memcpy_d2h(np.empty(nE), d_p)
    plan.set_option(_lib.OPT_FAST_PROBS, 1)                       # (the walk on the matrix cores: <= 1e-10, not bit-identical)
    t_pf = timed(lambda: plan.fill_probs_dev(d_p), plan, 20)
    fast_err = float(np.abs(plan.memcpy_d2h(np.empty(nE), d_p) - exact_p).max() / max(1.0, np.abs(exact_p).max()))
    plan.set_option(_lib.OPT_FAST_PROBS, 0)
    plan.fill_probs_dev(d_p); plan.sync()
    cols = np.arange(576, 576 + n_cols, dtype=np.int64)          # one whole gate's parameters (64 x 64)
    d_J = plan.device_malloc(nE * n_cols * 8)
    t_fd = timed(lambda: plan.fill_dprobs_dev(d_J, n_cols, cols, None, 1e-7, d_p, _lib.DERIV_FD), plan, 2)
    st = plan.stats()
    t_an = timed(lambda: plan.fill_dprobs_dev(d_J, n_cols, cols, None, 1e-7, d_p, _lib.DERIV_ANALYTIC), plan, 5)
    plan.device_free(d_J)
    allc = np.arange(nP, dtype=np.int64)
    d_Jf = plan.device_malloc(nE * nP * 8)
    t_anf = timed(lambda: plan.fill_dprobs_dev(d_Jf, nP, allc, None, 1e-7, d_p, _lib.DERIV_ANALYTIC), plan, 3)
    plan.device_free(d_Jf)
    i1 = np.arange(576, 576 + 16, dtype=np.int64); i2 = np.arange(576, 576 + 256, dtype=np.int64)
    t0 = time.perf_counter(); H = plan.fill_hprobs(idx1=i1, idx2=i2, eps=1e-5); t_h = time.perf_counter() - t0
    plan.fill_hprobs(idx1=i1[:4], idx2=i2, mode=_lib.DERIV_ANALYTIC)
    t0 = time.perf_counter(); Ha = plan.fill_hprobs(idx1=i1, idx2=i2, mode=_lib.DERIV_ANALYTIC); t_ha = time.perf_counter() - t0
    h_diff = float(np.abs(H - Ha).max() / max(np.abs(Ha).max(), 1e-300))
    flops = n_cols * 2.0 * D * D * st["applies_per_pass"]
    roof = {"probs": _roof_compute(2.0 * D * D * st["applies_per_pass"] + 2.0 * D * nE, t_p),
            "dprobs_fd_block": dict(_roof_compute(flops + n_cols * 2.0 * D * nE, t_fd), compute_unit="valu_f64",
                                    note="un-fused multiply/add (bit parity): its own ceiling is half the FMA peak"),
            "dprobs_analytic_full": _roof_hbm(8.0 * nE * nP, t_anf),
            "hprobs_fd_block_16x256": dict(_roof_compute(16 * 256 * (2.0 * D * D * st["applies_per_pass"] + 2.0 * D * nE), t_h),
                                           note="incl. the D2H copy of the block (host output)")}
    return {"roofline": roof, "config": "3-qubit explicit dense model (BASELINE configs[4] shape): D=64, 10 gates, 8 outcomes, nP=%d; %d seeded random circuits, "
                      "lengths 1..%d, nE=%d" % (nP, n_circ, max_len, nE),
            "probs_ms": 1e3 * t_p, "probs_per_s": nE / t_p,
            "fast_probs_ms": 1e3 * t_pf, "fast_probs_max_diff_rel_to_largest": fast_err, "largest_abs_prob": float(np.abs(exact_p).max()),
            "dprobs_fd_cols": int(n_cols), "dprobs_fd_ms": 1e3 * t_fd, "dprobs_fd_el_per_s": nE * n_cols / t_fd,
            "dprobs_fd_TFLOPs_algorithmic": flops / t_fd / 1e12,
            "dprobs_analytic_same_block_ms": 1e3 * t_an, "dprobs_analytic_same_block_el_per_s": nE * n_cols / t_an,
            "dprobs_analytic_full_ms": 1e3 * t_anf, "dprobs_analytic_full_el_per_s": nE * nP / t_anf,
            "dprobs_analytic_full_GBps": 8.0 * nE * nP / t_anf / 1e9,
            "hprobs_block": "16 x 256", "hprobs_ms_incl_d2h": 1e3 * t_h, "hprobs_el_per_s": nE * 16 * 256 / t_h,
            "hprobs_analytic_ms_incl_d2h": 1e3 * t_ha, "hprobs_fd_vs_exact_rel": h_diff,
            "applies_per_pass": st["applies_per_pass"], "n_tasks": st["n_tasks"],
            "note": "FD: bit-exact, register-blocked kernel (16 models per wavefront), no MFMA -- separate multiply/add is what parity "
                    "with the reference Map path requires; analytic: backward states over the suffix trie + 64x64 MFMA fp64 blocks"}


def two_q_hessian_block(plan, nE, nP, applies_per_pass, D=16, n1=2, n2=256):
    """One (n1 x n2) rectangle of the objective's Hessian on the bench design, Map-simulator semantics (FD of FD,
    eps 1e-5): hprobs block + both Jacobian blocks produced and contracted with the objective's dterms / hterms on the
    device (gst_objective_hessian_block); only n1 * n2 numbers come back."""
    rng = np.random.default_rng(7)
    d_c = plan.device_malloc(nE * 8); d_N = plan.device_malloc(nE * 8)
    plan.memcpy_h2d(d_c, rng.integers(0, 1000, nE).astype(np.float64)); plan.memcpy_h2d(d_N, np.full(nE, 1000.0))
    rows = np.arange(80, 80 + n1, dtype=np.int64); cols = np.arange(80, 80 + n2, dtype=np.int64)
    try:
        plan.objective_hessian_block("logl", d_c, d_N, rows, cols, 1e-5, 1e-4, 1e-4, None, _lib.DERIV_FD)
        t0 = time.perf_counter()
        reps = 2
        for _ in range(reps):
            plan.objective_hessian_block("logl", d_c, d_N, rows, cols, 1e-5, 1e-4, 1e-4, None, _lib.DERIV_FD)
        t = (time.perf_counter() - t0) / reps
    finally:
        plan.device_free(d_c); plan.device_free(d_N)
    flops = n1 * n2 * (2.0 * D * D * applies_per_pass + 2.0 * D * nE)
    return {"config": "2Q bench design, objective Hessian rectangle %d x %d (FD of FD, eps 1e-5; gst_objective_hessian_block)" % (n1, n2),
            "ms": 1e3 * t, "hessian_elements_per_s": nE * n1 * n2 / t,
            "roofline": dict(_roof_compute(flops, t), compute_unit="valu_f64",
                             note="flops of the n1*n2 doubly perturbed passes of the reference's schedule; includes the two FD Jacobian blocks, the contraction and the blocking read-back of n1*n2 numbers")}


if __name__ == "__main__":
    print(json.dumps(one_q()))
    print(json.dumps(three_q()))
