"""The secondary legs of bench.py (everything the JSON line reports beside `value`): each takes the namespace `B` that bench.py
fills after the timed region -- the plan with the headline Jacobian resident in d_out, the rank context, the workload --
and returns its entry of the JSON line.  A leg leaves the plan as it found it: model, parameter map and the (unscaled)
headline Jacobian.  Split out of bench.py in round 5 (the way tools/bench_configs.py holds the other configurations);
nothing here is part of the timed region."""
import time

import numpy as np

from pygsti_amd import _lib


def _unpack(B):
    return (B.args, B.plan, B.ctx, B.comm, B.world, B.rank, B.lay_world, B.layout, B.model, B.pack, B.gates, B.rhos, B.effects, B.d_out, B.d_probs, B.d_pfull, B.pidx, B.nP, B.nP_local, B.nE_local, B.nE_total, B.mode, B.blocks, B.row0, B.exchange, B.grid, B.col_split, B.barrier_sync, B.exchange_probs, B.log, B.n_pr, B.HBM_PEAK_GBS)


def fast_probs(B):
    (args, plan, ctx, comm, world, rank, lay_world, layout, model, pack, gates, rhos, effects, d_out, d_probs, d_pfull, pidx, nP, nP_local, nE_local, nE_total, mode, blocks, row0, exchange, grid, col_split, barrier_sync, exchange_probs, log, n_pr, HBM_PEAK_GBS) = _unpack(B)
    # secondary: the same fill through the log-depth level pass (GST_OPT_FAST_PROBS: germ-power paths by matrix squaring and
    # doubling on the MFMA cores, <= 1e-10 against the bit-exact probabilities -- what a line search of the optimizer may use)
    fast_info = None
    try:
        exact = plan.memcpy_d2h(np.empty(nE_local), d_probs)
        plan.set_option(_lib.OPT_FAST_PROBS, 1)
        for _ in range(2):
            plan.fill_probs_dev(d_probs)
        barrier_sync(plan)
        used = plan.stats()["last_levels"]
        tf0 = time.perf_counter()
        for _ in range(n_pr):
            plan.set_model(gates, rhos, effects)
            plan.fill_probs_dev(d_probs)
            exchange_probs()
        barrier_sync(plan)
        dtf = ctx.max_over_ranks(time.perf_counter() - tf0)
        fast = plan.memcpy_d2h(np.empty(nE_local), d_probs)
        fast_info = {"ms": 1e3 * dtf / n_pr, "probs_per_s": nE_total * n_pr / dtf, "level_pass_used": bool(used),
                     "max_abs_vs_bit_exact_probs": float(np.abs(fast - exact).max()),
                     "note": "gst_fill_probs_dev under GST_OPT_FAST_PROBS; bar 1e-10; the default fill stays bit-identical to the reference"}
    except Exception as e:
        fast_info = {"error": "%s: %s" % (type(e).__name__, e)}
    finally:
        plan.set_option(_lib.OPT_FAST_PROBS, 0)
        plan.fill_probs_dev(d_probs)
        barrier_sync(plan)
    return fast_info



def store_only_ceiling(device, nbytes=4 << 30, reps=10):
    """What a store-only kernel reaches on this part: hipMemsetAsync (the runtime's fill kernel) over a scratch buffer far
    larger than L2 + the 256 MB memory-side cache.  The exact Jacobian is a store stream (8 nE nP bytes written once, reads
    a few per cent of that), so THIS is its practical roof; the 8 TB/s figure is the read peak.  Returns GB/s or None."""
    import ctypes
    try:
        hip = ctypes.CDLL("libamdhip64.so")
        hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
        hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
        hip.hipFree.argtypes = [ctypes.c_void_p]
        if hip.hipSetDevice(int(device)) != 0:
            return None
        p = ctypes.c_void_p()
        if hip.hipMalloc(ctypes.byref(p), nbytes) != 0:
            return None
        for _ in range(3):
            hip.hipMemsetAsync(p, 0, nbytes, None)
        hip.hipDeviceSynchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            hip.hipMemsetAsync(p, 0, nbytes, None)
        hip.hipDeviceSynchronize()
        dt = (time.perf_counter() - t0) / reps
        hip.hipFree(p)
        return nbytes / dt / 1e9
    except Exception:
        return None

def analytic(B):
    (args, plan, ctx, comm, world, rank, lay_world, layout, model, pack, gates, rhos, effects, d_out, d_probs, d_pfull, pidx, nP, nP_local, nE_local, nE_total, mode, blocks, row0, exchange, grid, col_split, barrier_sync, exchange_probs, log, n_pr, HBM_PEAK_GBS) = _unpack(B)
    # secondary: the same Jacobian by analytic derivatives (MatrixForwardSimulator semantics, <= 1e-8 vs that simulator)
    ana_info = None
    if args.deriv == "fd" and not args.no_analytic:
        for _ in range(2):
            plan.fill_dprobs_dev(d_out, nP, pidx, None, 1e-7, d_probs, _lib.DERIV_ANALYTIC)
        barrier_sync(plan)
        ta0 = time.perf_counter()
        n_an = max(3, args.steps)
        for _ in range(n_an):
            plan.set_model(gates, rhos, effects)
            plan.fill_dprobs_dev(d_out, nP, pidx, None, 1e-7, d_probs, _lib.DERIV_ANALYTIC)
            exchange_probs()
        barrier_sync(plan)
        dta = ctx.max_over_ranks(time.perf_counter() - ta0)
        ana_kms = plan.stats()["last_kernel_ms"]
        ana_bytes = 8.0 * nE_local * nP
        ana_info = {"value": nE_total * nP * n_an / dta, "unit": "Jacobian-elements/s", "ms_per_step": 1e3 * dta / n_an,
                    "kernel_ms": ana_kms,
                    "roofline": {"bound": "hbm", "kernel": "analytic_mfma_kernel (+ the backward chain pass that feeds it)",
                                 "achieved": ana_bytes / (ana_kms * 1e-3) / 1e9 if ana_kms > 0 else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": ana_bytes / (ana_kms * 1e-3) / 1e9 / HBM_PEAK_GBS if ana_kms > 0 else None,
                                 "whole_step_frac": ana_bytes / (dta / n_an) / 1e9 / HBM_PEAK_GBS,
                                 "bytes_per_launch": ana_bytes,
                                 "note": "algorithmic bytes = the Jacobian write 8*nE*nP; `frac` over the contraction kernel, `whole_step_frac` over the step (chain passes included)"},
                    "chain_passes": "log-depth level passes" if plan.stats()["last_levels"] else "sequential walks",
                    "structural_zeros": ("resident (the destination is tracked device memory whose previous contents were this fill's: "
                                         "the zero blocks of gates a circuit never applies are not stored again)"
                                         if plan.stats()["last_zeros_resident"] else "stored by every fill"),
                    "note": "analytic derivatives (reference MatrixForwardSimulator semantics); secondary figure, not `value`"}
        plan.fill_dprobs_dev(d_out, nP, pidx, None, 1e-7, d_probs, mode)      # leave the FD Jacobian resident
        barrier_sync(plan)
    return ana_info


def cptplnd(B):
    (args, plan, ctx, comm, world, rank, lay_world, layout, model, pack, gates, rhos, effects, d_out, d_probs, d_pfull, pidx, nP, nP_local, nE_local, nE_total, mode, blocks, row0, exchange, grid, col_split, barrier_sync, exchange_probs, log, n_pr, HBM_PEAK_GBS) = _unpack(B)
    # secondary (N=1): the same design under the CPTPLND parameterisation StandardGST fits by default -- every member a
    # static target composed with an exponentiated Lindblad error generator (1,920 parameters), the dense members built
    # ON THE DEVICE for the base model and for every finite-difference step (gst_set_lindblad)
    cptp_info = None
    if world == 1 and lay_world == 1 and not args.no_cptplnd:
        from pygsti_amd import lindblad as LBM
        lmodel = LBM.LindbladModel.from_target(pack.target_model(), layout.model_gate_labels, layout.effect_labels, "CPTPLND")
        theta = 0.003 * np.random.default_rng(9).standard_normal(lmodel.num_params)
        nPl = lmodel.num_params
        plan.set_lindblad(lmodel)
        d_outl = plan.device_malloc(nE_local * nPl * 8, tracked=True)
        pidx_l = np.arange(nPl, dtype=np.int64)
        try:
            t_set0 = time.perf_counter()
            plan.set_lindblad_params(theta)
            t_set = time.perf_counter() - t_set0
            n_c = max(2, min(args.steps, 3))
            # exact derivatives: element Jacobian (analytic contraction) x the members' derivative matrices (Frechet derivative
            # of the exponential), both computed on the device -- MatrixForwardSimulator semantics for this parameterisation.
            # (FD over device-built members is not offered at this depth: GST_LINDBLAD_FD_MAX_DEPTH, include/gstfwd.h)
            plan.fill_dprobs_dev(d_outl, nPl, pidx_l, None, 1e-7, d_probs, _lib.DERIV_ANALYTIC)
            barrier_sync(plan)
            ta0 = time.perf_counter()
            for _ in range(n_c):
                plan.set_lindblad_params(theta)
                plan.fill_dprobs_dev(d_outl, nPl, pidx_l, None, 1e-7, d_probs, _lib.DERIV_ANALYTIC)
            barrier_sync(plan)
            dtc = (time.perf_counter() - ta0) / n_c
            chk = plan.memcpy_d2h(np.empty(nPl), d_outl + ((nE_local - 1) * nPl) * 8)
            assert np.isfinite(chk).all()
            refused = None
            try:
                plan.fill_dprobs_dev(d_outl, nPl, pidx_l, None, 1e-7, d_probs, _lib.DERIV_FD)
            except _lib.GstUnsupported as e:
                refused = str(e)[:160]
            cptp_info = {"value": nE_local * nPl / dtc, "unit": "Jacobian-elements/s", "ms_per_step": 1e3 * dtc, "n_params": nPl,
                         "set_params_ms": 1e3 * t_set, "derivative": "exact (GST_DERIV_ANALYTIC)",
                         "fd_over_device_built_members": "refused at this depth" if refused else "NOT refused (unexpected)",
                         "parity": "exact route vs the Matrix simulator: <= 1e-8 at every depth (1.2e-11 at depth 1,030; tests/test_gpu_lindblad.py); "
                                   "FD at depth = gst_fill_dprobs_models over the reference's own stepped members (<= 1e-8)",
                         "note": "bulk_fill_dprobs of the CPTPLND-parameterised model: base members and every member's d(dense)/d(parameter) "
                                 "built on the device from the parameter vector, element Jacobian x chain-rule GEMM on the matrix cores; "
                                 "secondary figure, not `value`"}
        finally:
            plan.device_free(d_outl)
            plan.set_lindblad(None)
            plan.set_model(gates, rhos, effects)
            plan.set_param_map(*layout.param_map(model))
        plan.fill_dprobs_dev(d_out, nP, pidx, None, 1e-7, d_probs, mode)      # leave the headline Jacobian resident
        barrier_sync(plan)
        log("CPTPLND leg done: %.1f ms per Jacobian" % (1e3 * dtc))
    return cptp_info


def jacobian_gather(B):
    (args, plan, ctx, comm, world, rank, lay_world, layout, model, pack, gates, rhos, effects, d_out, d_probs, d_pfull, pidx, nP, nP_local, nE_local, nE_total, mode, blocks, row0, exchange, grid, col_split, barrier_sync, exchange_probs, log, n_pr, HBM_PEAK_GBS) = _unpack(B)
    # secondary (N>1): the fan-in of the Jacobian row blocks to rank 0 between device buffers (Gatherv,
    # resourceallocation.py:329-348 -- what `gather_local_array('ep', ...)` is for), alone and behind a fill
    if world > 1 and comm is not None and not args.no_jacobian_gather and args.scaling == "strong":
        d_jfull = plan.device_malloc(nE_total * nP * 8) if rank == 0 else None
        try:
            if rank == 0:       # the root's own rows in place (one device-to-device copy per fill in a real run)
                plan.fill_dprobs_dev(d_jfull + row0 * nP * 8, nP, pidx, None, 1e-7, d_probs, mode)
            comm.gather_rows(d_out, d_jfull, nP, blocks, 0, plan)                       # warm-up (maps peers, opens links)
            barrier_sync(plan)
            tg = time.perf_counter()
            n_g = 3
            for _ in range(n_g):
                comm.gather_rows(d_out, d_jfull, nP, blocks, 0, plan)
            barrier_sync(plan)
            t_gather = ctx.max_over_ranks(time.perf_counter() - tg) / n_g
            tg = time.perf_counter()
            for _ in range(n_g):
                plan.set_model(gates, rhos, effects)
                dst = d_jfull + row0 * nP * 8 if rank == 0 else d_out
                plan.fill_dprobs_dev(dst, nP, pidx, None, 1e-7, d_probs, mode)
                comm.gather_rows(d_out, d_jfull, nP, blocks, 0, plan)
            barrier_sync(plan)
            t_both = ctx.max_over_ranks(time.perf_counter() - tg) / n_g
            moved = 8.0 * (nE_total - nE_local) * nP if rank == 0 else 0.0
            moved = ctx.max_over_ranks(moved)
            exchange["jacobian_gather_to_rank0"] = {
                "ms": 1e3 * t_gather, "GB": moved / 1e9, "GBps_into_rank0": moved / t_gather / 1e9,
                "fill_plus_gather_ms": 1e3 * t_both,
                "elements_per_s_with_gather": nE_total * nP / t_both,
                "note": "secondary: rows of the other ranks written into rank 0's assembled [nE][nP] array, each block "
                        "over its own xGMI link (grouped point-to-point under RCCL, peer copies under IPC)"}
            if rank == 0:
                chk = plan.memcpy_d2h(np.empty(nP), d_jfull + ((nE_total - 1) * nP) * 8)
                assert np.isfinite(chk).all()
        finally:
            if d_jfull is not None:
                plan.device_free(d_jfull)


def lm_step(B):
    (args, plan, ctx, comm, world, rank, lay_world, layout, model, pack, gates, rhos, effects, d_out, d_probs, d_pfull, pidx, nP, nP_local, nE_local, nE_total, mode, blocks, row0, exchange, grid, col_split, barrier_sync, exchange_probs, log, n_pr, HBM_PEAK_GBS) = _unpack(B)
    # secondary (every N): one Levenberg-Marquardt iteration's device work end to end -- model upload, Jacobian fill,
    # objective maps (lsvec, dlsvec row scale), J_s^T J_s and J_s^T lsvec on the resident Jacobian, and for N > 1 the
    # all-reduce of the nP^2 + nP partial sums between device buffers (the path's one real exchange step, SURVEY 8(e) /
    # row f1).  This is the figure multi-GPU scaling of a FIT hinges on: no Jacobian ever leaves its GPU.
    lm_info = None
    if not col_split and not args.no_lm_step:
        bufs = [plan.device_malloc(n) for n in (nP * nP * 8, nP * 8, nE_local * 8, nE_local * 8, nE_local * 8, nE_local * 8)]
        d_jtj, d_jtf, d_ls, d_w, d_c, d_N = bufs
        try:
            pb = plan.memcpy_d2h(np.empty(nE_local), d_probs)
            plan.memcpy_h2d(d_c, np.random.default_rng(1234 + rank).binomial(1000, np.clip(pb, 0.0, 1.0)).astype(np.float64))
            plan.memcpy_h2d(d_N, np.full(nE_local, 1000.0))

            lm_mode = [mode]
            lm_in_place = [False]

            def lm_step():
                plan.set_model(gates, rhos, effects)
                plan.fill_dprobs_dev(d_out, nP_local, pidx, None, 1e-7, d_probs, lm_mode[0])
                plan.objective_rows_dev("logl", d_probs, d_c, d_N, nE_local, d_ls, d_w, want_sum=False)
                if lm_in_place[0]:      # rounds 2-4: the rows of J scaled in place by a streaming pass, then contracted
                    plan.fill_jtj_dev(d_out, nE_local, nP, nP, d_jtj, d_w)
                    plan.fill_jtf_dev(d_out, nE_local, nP, nP, d_ls, d_jtf)
                else:                   # round 5: the dlsvec row factors applied while the rows are staged (J only read)
                    plan.fill_normal_eqs_dev(d_out, nE_local, nP, nP, d_w, d_ls, d_jtj, d_jtf)
                if world > 1 and comm is not None:
                    comm.allreduce_sum(d_jtj, nP * nP, plan)
                    comm.allreduce_sum(d_jtf, nP, plan)
            lm_step()
            barrier_sync(plan)
            n_lm = max(3, min(args.steps, 5))
            tl = time.perf_counter()
            for _ in range(n_lm):
                lm_step()
            barrier_sync(plan)
            t_lm = ctx.max_over_ranks(time.perf_counter() - tl) / n_lm
            chk = plan.memcpy_d2h(np.empty(nP), d_jtf)
            assert np.isfinite(chk).all()
            lm_info = {"ms_per_step": 1e3 * t_lm, "elements_per_s": nE_total * nP / t_lm, "unit": "Jacobian-elements/s",
                       "allreduce_MB": (nP * nP + nP) * 8 / 1e6 if world > 1 else 0.0,
                       "allreduce_transport": ctx.transport if world > 1 else None,
                       "note": "fill + Poisson-picture dlogl maps + J_s^T J_s (block-sparse split-K MFMA fp64 SYRK) + J_s^T f, dlsvec row "
                               "factors applied while the rows are staged (gst_fill_normal_eqs_dev)"
                               + (" + all-reduce of nP^2 + nP doubles between device buffers" if world > 1 else "")
                               + "; the Jacobian never leaves HBM; secondary figure, not `value`"}
            if mode == _lib.DERIV_FD and not args.no_analytic:
                # the same iteration with the exact Jacobian (what an optimizer that does not insist on the Map simulator's
                # finite differences would run; the structural zeros of the re-used Jacobian stay resident under the scaling)
                lm_mode[0] = _lib.DERIV_ANALYTIC
                lm_step(); lm_step()
                barrier_sync(plan)
                tl = time.perf_counter()
                for _ in range(n_lm):
                    lm_step()
                barrier_sync(plan)
                t_lma = ctx.max_over_ranks(time.perf_counter() - tl) / n_lm
                assert np.isfinite(plan.memcpy_d2h(np.empty(nP), d_jtf)).all()
                lm_info["exact_jacobian_ms_per_step"] = 1e3 * t_lma
                lm_info["exact_jacobian_zeros_resident"] = bool(plan.stats()["last_zeros_resident"])
                jtf_new = plan.memcpy_d2h(np.empty(nP), d_jtf); jtj_new = plan.memcpy_d2h(np.empty((nP, nP)), d_jtj)
                lm_in_place[0] = True
                lm_step(); lm_step()
                barrier_sync(plan)
                tl = time.perf_counter()
                for _ in range(n_lm):
                    lm_step()
                barrier_sync(plan)
                lm_info["exact_jacobian_in_place_scaling_ms_per_step"] = 1e3 * ctx.max_over_ranks(time.perf_counter() - tl) / n_lm
                jtf_old = plan.memcpy_d2h(np.empty(nP), d_jtf)
                lm_info["weights_on_the_fly_jtj_same_bits_as_in_place"] = bool(np.array_equal(jtj_new, plan.memcpy_d2h(np.empty((nP, nP)), d_jtj)))
                lm_info["weights_on_the_fly_jtf_max_rel_diff"] = float(np.abs(jtf_new - jtf_old).max() / max(np.abs(jtf_old).max(), 1e-300))
                lm_info["jtf_max_abs"] = float(np.abs(jtf_old).max())
                lm_in_place[0] = False
                lm_mode[0] = mode
        finally:
            for d in bufs:
                plan.device_free(d)
        plan.set_model(gates, rhos, effects)
        plan.fill_dprobs_dev(d_out, nP_local, pidx, None, 1e-7, d_probs, mode)      # leave the headline Jacobian resident (unscaled)
        barrier_sync(plan)
        log("LM-step leg done: %.2f ms" % (1e3 * t_lm))
    return lm_info


def normal_equations(B):
    (args, plan, ctx, comm, world, rank, lay_world, layout, model, pack, gates, rhos, effects, d_out, d_probs, d_pfull, pidx, nP, nP_local, nE_local, nE_total, mode, blocks, row0, exchange, grid, col_split, barrier_sync, exchange_probs, log, n_pr, HBM_PEAK_GBS) = _unpack(B)
    jtj_info = None
    if args.jtj and col_split and world > 1 and comm is not None:
        # Normal equations with the columns distributed (distlayout.py:1306-1346): the ranks of an atom-processor split
        # the atom's ROWS, exchange each other's column blocks between device buffers (gst_comm_exchange_blocks), and
        # each contracts its [rows/NP x nP] share; then the usual all-reduce of nP^2 doubles.
        from pygsti_amd.layout import _slice_up_range
        G = grid[1]
        share = _slice_up_range(nE_local, G)[layout.param_proc_index]
        n_my = share.stop - share.start
        xblocks = layout.column_exchange_blocks(0)
        d_stage = plan.device_malloc(max(n_my * nP, 1) * 8); d_T = plan.device_malloc(max(n_my * nP, 1) * 8)
        d_jtj = plan.device_malloc(nP * nP * 8)

        def assemble():
            comm.exchange_blocks(d_out, d_stage, xblocks, plan)
            for cs in layout.param_slices:
                c = cs.stop - cs.start
                plan.copy_block_dev(d_T + cs.start * 8, nP, d_stage + n_my * cs.start * 8, c, n_my, c)
        assemble(); plan.fill_jtj_dev(d_T, n_my, nP, nP, d_jtj); comm.allreduce_sum(d_jtj, nP * nP, plan)      # warm-up
        barrier_sync(plan)
        tj = time.perf_counter()
        for _ in range(3):
            assemble()
        barrier_sync(plan)
        t_x = ctx.max_over_ranks((time.perf_counter() - tj) / 3)
        tj = time.perf_counter()
        for _ in range(3):
            plan.fill_jtj_dev(d_T, n_my, nP, nP, d_jtj)
        barrier_sync(plan)
        t_jtj = ctx.max_over_ranks((time.perf_counter() - tj) / 3)
        tj = time.perf_counter()
        for _ in range(3):
            comm.allreduce_sum(d_jtj, nP * nP, plan)
        barrier_sync(plan)
        t_ar = ctx.max_over_ranks((time.perf_counter() - tj) / 3)
        sent = 8.0 * nE_local * nP_local * (G - 1) / G
        jtj_info = {"grid": "%dx%d" % grid, "column_exchange_ms": 1e3 * t_x, "column_exchange_GB_sent_per_rank": sent / 1e9,
                    "column_exchange_GBps_per_rank": sent / t_x / 1e9, "jtj_ms": 1e3 * t_jtj, "allreduce_ms": 1e3 * t_ar,
                    "note": "columns distributed over the parameter-processors of an atom-processor: every rank sends the other "
                            "ranks' row shares of its column slice and receives their slices of its own share (grouped "
                            "point-to-point under RCCL, peer copies under IPC), contracts [rows/NP x nP] with the MFMA SYRK, "
                            "then the nP^2 all-reduce"}
        for d in (d_stage, d_T, d_jtj):
            plan.device_free(d)
    elif args.jtj and not col_split:
        # One Levenberg-Marquardt iteration's worth of data reduction on the resident Jacobian (row f1):
        # probabilities -> lsvec and dlsvec row scale (Poisson-picture dlogl, synthetic counts N=1000 drawn around the
        # model's own probabilities) -> J_s^T J_s (split-K MFMA fp64) and J_s^T lsvec (streaming).
        d_jtj = plan.device_malloc(nP * nP * 8); d_jtf = plan.device_malloc(nP * 8)
        d_ls = plan.device_malloc(nE_local * 8); d_w = plan.device_malloc(nE_local * 8)
        d_c = plan.device_malloc(nE_local * 8); d_N = plan.device_malloc(nE_local * 8)
        pb = plan.memcpy_d2h(np.empty(nE_local), d_probs)
        rngc = np.random.default_rng(1234 + rank)
        plan.memcpy_h2d(d_c, rngc.binomial(1000, np.clip(pb, 0.0, 1.0)).astype(np.float64))
        plan.memcpy_h2d(d_N, np.full(nE_local, 1000.0))
        obj = plan.objective_rows_dev("logl", d_probs, d_c, d_N, nE_local, d_ls, d_w)                                # warm-up
        plan.fill_jtj_dev(d_out, nE_local, nP, nP, d_jtj); plan.fill_jtf_dev(d_out, nE_local, nP, nP, d_ls, d_jtf)
        barrier_sync(plan)
        tj = time.perf_counter()
        for _ in range(10):
            plan.objective_rows_dev("logl", d_probs, d_c, d_N, nE_local, d_ls, d_w, want_sum=False)
        barrier_sync(plan)
        t_obj = (time.perf_counter() - tj) / 10
        tj = time.perf_counter()
        for _ in range(3):
            plan.fill_jtj_dev(d_out, nE_local, nP, nP, d_jtj)      # (row scaling in place is a separate 2-pass stream; timed below)
        barrier_sync(plan)
        t_jtj = (time.perf_counter() - tj) / 3
        tj = time.perf_counter()
        for _ in range(3):
            plan.fill_jtf_dev(d_out, nE_local, nP, nP, d_ls, d_jtf)
        barrier_sync(plan)
        t_jtf = (time.perf_counter() - tj) / 3
        tj = time.perf_counter()
        plan.fill_jtj_dev(d_out, nE_local, nP, nP, d_jtj, d_w)     # once, with the row scale (J is scaled in place)
        barrier_sync(plan)
        t_jtj_scaled = time.perf_counter() - tj
        jtj_info = {"objective": "Poisson-picture dlogl, min_prob_clip=radius=1e-4", "objective_value": obj,
                    "objective_rows_ms": 1e3 * t_obj,
                    "jtj_ms": 1e3 * t_jtj, "jtj_TFLOPs": 2.0 * nE_local * nP * nP / 2 / t_jtj / 1e12,
                    "jtj_with_row_scale_ms": 1e3 * t_jtj_scaled,
                    "jtf_ms": 1e3 * t_jtf, "jtf_GBps": 8.0 * nE_local * nP / t_jtf / 1e9,
                    "note": "element-wise objective kernel + hand-written split-K MFMA fp64 SYRK / streaming GEMV on the device-resident Jacobian of this rank (flops counted for the triangle: nE*nP^2)"}
        if world > 1 and comm is not None:
            # the path's one real exchange step: every rank holds the partial J^T J of its rows; the optimizer needs
            # the sum -> one all-reduce of nP^2 (+ nP) doubles between the device buffers, cf. distlayout.py:1259,1355
            comm.allreduce_sum(d_jtj, nP * nP, plan)               # warm-up
            barrier_sync(plan)
            ta = time.perf_counter()
            for _ in range(5):
                comm.allreduce_sum(d_jtj, nP * nP, plan)
                comm.allreduce_sum(d_jtf, nP, plan)
            barrier_sync(plan)
            jtj_info["allreduce_ms"] = 1e3 * ctx.max_over_ranks(time.perf_counter() - ta) / 5
            jtj_info["allreduce_MB"] = (nP * nP + nP) * 8 / 1e6
            jtj_info["allreduce_transport"] = ctx.transport
        for d in (d_jtj, d_jtf, d_ls, d_w, d_c, d_N):
            plan.device_free(d)
    return jtj_info


def host_fill(B):
    (args, plan, ctx, comm, world, rank, lay_world, layout, model, pack, gates, rhos, effects, d_out, d_probs, d_pfull, pidx, nP, nP_local, nE_local, nE_total, mode, blocks, row0, exchange, grid, col_split, barrier_sync, exchange_probs, log, n_pr, HBM_PEAK_GBS) = _unpack(B)
    # secondary (N=1): the reference API end to end -- bulk_fill_dprobs into the caller's HOST 'ep' array, PCIe included
    host_fill = None
    if world == 1 and lay_world == 1 and not args.no_host_fill:
        J_host = layout.allocate_local_array("ep", "d")            # page-locked (registered) by the layout
        log("host array allocated, pinned=%s" % layout.last_array_pinned)
        pr_host = np.empty(nE_local)
        plan.fill_dprobs(J_host, pidx, None, 1e-7, pr_host, mode)  # warm-up
        th = time.perf_counter()
        n_h = 3
        for _ in range(n_h):
            plan.set_model(gates, rhos, effects)
            plan.fill_dprobs(J_host, pidx, None, 1e-7, pr_host, mode)
        t_host = (time.perf_counter() - th) / n_h
        host_fill = {"ms": 1e3 * t_host, "elements_per_s": nE_local * nP / t_host, "GBps": 8.0 * nE_local * nP / t_host / 1e9,
                     "pinned": bool(getattr(layout, "last_array_pinned", False)),
                     "note": "gst_fill_dprobs into a host numpy 'ep' array from layout.allocate_local_array (what "
                             "bulk_fill_dprobs(array, layout) returns in the reference): the FD kernel stores 7 GB straight into the page-locked array over PCIe; never `value`"}
        layout.free_local_array(J_host)
        log("host-fill leg done: %.1f ms per fill" % (1e3 * t_host))
    return host_fill
