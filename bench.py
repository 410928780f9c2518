#!/usr/bin/env python3
"""bench.py -- the BASELINE.json metric on MI355X.

Metric : dprobs Jacobian-elements/s (+ circuit-outcome probs/s as a secondary figure), 2Q GST L<=1024.
Workload: smq2Q_XYICNOT target model .depolarize(0.01, 0.01), `create_gst_experiment_design(1024, lite=False)`
          = 136,275 circuits, nE = 545,100 elements, nP = 1,616 parameters, D = 16 (BASELINE configs[2], the
          "~10^5 circuits" case).  One "step" = one bulk_fill_dprobs over the whole design (base
          probabilities + all 1,616 finite-difference columns, derivative_eps = 1e-7) exactly as the reference's
          MapForwardSimulator computes it (mapforwardsim_calc_densitymx.pyx:290-383) -- bit-identical results --
          with the Jacobian left resident in HBM (7.05 GB).  The model arrays are re-uploaded every step, as an
          optimizer iteration does after model.from_vector().
Multi-GPU: the path shards by layout atoms (contiguous runs of circuits), one atom per rank, no data-path
          collective (the reference's bulk_fill_dprobs leaves rows distributed, too).  `--scaling strong` (default,
          the mode BASELINE.json's north_star names): the ONE design is dealt to N atoms, total work fixed.
          `--scaling weak`: every rank's atom is one full design's worth of circuits (the global layout is N
          designs, as in a bootstrap / multi-dataset GST run where each replica carries its own model estimate --
          rank r's model is depolarized by 0.01 + 0.001 r), per-GPU work fixed.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (walk_kernel<16,1,*>, fp64 VALU bound);
`cpu_baseline` times the CPU oracle on a bounded sample on this box's host cores (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

F64_VALU_PEAK_TFLOPS = 78.6   # MI355X fp64 vector peak (FMA = 2 flop); separate mul+add can reach half of it
HBM_PEAK_GBS = 8000.0


def build_workload(design, max_len, world, rank, device, target_tasks, max_slots=0, scaling="strong", grid=None, deriv="fd"):
    from pygsti_amd import modelpacks
    from pygsti_amd.layout import HipCOPALayout
    pack = modelpacks.smq2Q_XYICNOT
    circuits = pack.create_gst_circuits(max_len, lite=(design == "lite"))
    if scaling == "weak":
        # rank r owns atom r of an N-design layout: a full design's worth of circuits and its own model estimate
        model = pack.target_model().depolarize(op_noise=0.01 + 0.001 * rank, spam_noise=0.01)
        layout = HipCOPALayout(circuits, model, num_atoms=1, devices=[device], rank=0, size=1,
                               target_tasks=target_tasks, max_slots=max_slots)
    else:
        model = pack.target_model().depolarize(op_noise=0.01, spam_noise=0.01)
        layout = HipCOPALayout(circuits, model, num_atoms=grid[0] if grid else world, devices=[device], rank=rank, size=world,
                               target_tasks=target_tasks, max_slots=max_slots, processor_grid=grid,
                               partition_cost="depth" if deriv == "analytic" else "fd")
    return pack, model, circuits, layout


def usable_cores():
    """CPUs this process may actually use: scheduler affinity, capped by the cgroup CPU quota when one is set."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:                   # cgroup v2: "<quota|max> <period>"
            q, per = f.read().split()[:2]
            if q != "max":
                n = min(n, max(1, int(float(q) / float(per) + 0.5)))
    except (OSError, ValueError):
        try:                                                          # cgroup v1
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                q = int(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                per = int(f.read())
            if q > 0 and per > 0:
                n = min(n, max(1, int(q / per + 0.5)))
        except (OSError, ValueError):
            pass
    return max(1, n)


def parity_columns(model_dim, n_gates, n_effects, n_params):
    """The finite-difference columns of the parity leg: spread over the preparation, the effects and every gate (`full`
    parameterisation: rho | effects | gates in the parameter vector, one parameter per dense element)."""
    D = model_dim
    cols = [0, D // 2 + 1, D, D + (n_effects * D) // 2 - 1, D + n_effects * D - 1]
    g0 = D + n_effects * D
    for g in range(n_gates):
        cols += [g0 + g * D * D + (3 * g + 1) * D % (D * D) + (5 * g + 2) % D, g0 + g * D * D + D * D - 1 - 7 * g]
    cols = sorted(set(c for c in cols if 0 <= c < n_params))
    return np.array(cols, np.int64)


def reference_parity(orc, parity_ctx):
    """The benchmarked workload itself against the CPU checker (the reference's own C++ reps when oracle/_ref is built):
    ALL probabilities of the design and K full finite-difference columns of the HBM-resident Jacobian (one more step,
    identical to the timed ones), compared bit for bit."""
    plan, nE, nP = parity_ctx["plan"], parity_ctx["nE"], parity_ctx["nP"]
    cols = parity_ctx["cols"]
    t0 = time.perf_counter()
    Jo, po = orc.dprobs(cols, eps=1e-7, return_probs=True)
    t_cpu = time.perf_counter() - t0
    pd = plan.memcpy_d2h(np.empty(nE), parity_ctx["d_probs"])
    Jd = np.empty((nE, len(cols)))
    d_col = plan.device_malloc(nE * 8)
    tmp = np.empty(nE)
    try:
        for j, c in enumerate(cols):            # column c of the resident [nE][nP] Jacobian -> a contiguous device vector
            plan.copy_block_dev(d_col, 1, parity_ctx["d_out"] + int(c) * 8, nP, nE, 1)
            plan.memcpy_d2h(tmp, d_col)
            Jd[:, j] = tmp
    finally:
        plan.device_free(d_col)
    same_p = np.array_equal(pd.view(np.uint64), po.view(np.uint64))
    same_j = np.array_equal(Jd.view(np.uint64), np.ascontiguousarray(Jo).view(np.uint64))
    return {"bitwise": bool(same_p and same_j), "bitwise_probs": bool(same_p), "bitwise_dprobs": bool(same_j),
            "max_abs_probs": float(np.abs(pd - po).max()), "max_abs_dprobs": float(np.abs(Jd - Jo).max()),
            "n_rows": int(nE), "n_cols": int(len(cols)), "cols": [int(c) for c in cols], "checker": orc.kind,
            "checker_seconds": t_cpu, "max_abs_J_checked": float(np.abs(Jo).max()),
            "what": "every probability of the benchmarked design and %d whole columns (preparation, effects, all gates) of the "
                    "HBM-resident Jacobian of one more step identical to the timed ones vs the CPU checker's bulk_fill_dprobs on the reference-format prefix "
                    "table of the same design (checker = %s)" % (len(cols), "the reference's own C++ reps, oracle/_ref" if orc.kind == "reference" else "the C restatement, oracle/liboracle.so")}


def cpu_baseline(pack, model, max_len, design, parity_ctx=None):
    """Time the CPU checker on a bounded sample of THE SAME workload: the reference-format prefix table of the
    benchmarked design (restated PrefixTable, oracle/prefix_table.py) walked for a fixed number of passes -- one pass =
    one finite-difference column = one `dm_mapfill_probs` (mapforwardsim_calc_densitymx.pyx:194-287).  Returns the
    1-core figure (`cpu_baseline`) and the all-host-cores figure (`cpu_baseline_allcores`: the FD columns are
    independent, so every core walks the table for its own columns -- what OpenMP over columns / the reference's
    parameter-processor MPI split, distlayout.py:478-506, would do)."""
    import concurrent.futures as cf
    from oracle import oracle as O, prefix_table as PT
    circuits = pack.create_gst_circuits(max_len, lite=(design == "lite"))
    lookup = {l: i for i, l in enumerate(model.operations.keys())}
    ptr = np.zeros(len(circuits) + 1, np.int64)
    ptr[1:] = np.cumsum([len(c) for c in circuits])
    gates = np.fromiter((lookup[g] for c in circuits for g in c), np.int32, count=int(ptr[-1]))
    nO = len(model.effect_labels)
    tbl = PT.build_table(ptr, gates, nO)
    tbl["D"] = model.dim
    G = np.array([model.operations[l] for l in model.operations])
    R = np.array([next(iter(model.preps.values()))])
    E = np.array([model.effect_vector(l) for l in model.effect_labels])
    nP = model.num_params
    if parity_ctx is not None:
        pk, po_, pe = parity_ctx["param_map"]
    else:
        pk, po_, pe = -np.ones(nP, np.int32), np.zeros(nP, np.int32), np.zeros(nP, np.int32)
    mdl = dict(gates=G, rhos=R, effects=E, pkind=pk, pobj=po_, pelem=pe)
    have_ref = os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libgst_ref.so"))
    kind = "reference" if have_ref else "port"
    orc = O.Oracle(tbl, mdl, kind)
    nE = tbl["nE"]
    parity = None
    if parity_ctx is not None:
        assert nE == parity_ctx["nE"]
        parity = reference_parity(orc, parity_ctx)
        log("parity leg done: bitwise=%s" % parity["bitwise"])
    t1 = orc.time_passes(2)                      # calibrate
    n_pass = int(max(4, min(400, 10.0 / max(t1 / 2, 1e-6))))
    t = orc.time_passes(n_pass)
    what = "%s germs design itself (%d circuits, nE=%d, %d gate applications per pass on the reference-format prefix table)" % (
        design, len(circuits), nE, len(tbl["gate_idx"]))
    one = {"value": nE * n_pass / t, "unit": "Jacobian-elements/s", "cores": 1, "kind": kind,
           "sample": "%d FD columns (probability passes) over the %s, %.1f s" % (n_pass, what, t),
           "probs_per_s": nE * n_pass / t}
    # the restatement next to the reference's own C++ on the same table (maps "port" numbers back to "reference")
    if have_ref:
        n_r = max(2, n_pass // 8)
        port = O.Oracle(tbl, mdl, "port")
        tp = port.time_passes(n_r); tr = orc.time_passes(n_r)
        one["port_over_reference_time_ratio"] = tp / tr
    # all host cores: one thread per usable core, each with its own workspace (ctypes releases the GIL).  "Usable" =
    # the scheduler affinity capped by the container's CPU quota (a box may show 256 CPUs and grant 8)
    cores = usable_cores()
    workers = [O.Oracle(tbl, mdl, kind) for _ in range(cores)]
    with cf.ThreadPoolExecutor(cores) as ex:
        t0 = time.perf_counter()
        list(ex.map(lambda w: w.time_passes(1), workers))          # calibrate under full load
        t_conc = time.perf_counter() - t0
        per = int(max(1, min(n_pass, 10.0 / max(t_conc, 1e-6))))
        t0 = time.perf_counter()
        list(ex.map(lambda w: w.time_passes(per), workers))
        ta = time.perf_counter() - t0
    cpu_model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    cpu_model = line.split(":", 1)[1].strip(); break
    except OSError:
        pass
    allc = {"value": nE * per * cores / ta, "unit": "Jacobian-elements/s", "cores": cores, "kind": kind,
            "cpu_model": cpu_model, "nproc": os.cpu_count(), "cgroup_limited": cores < (os.cpu_count() or 1),
            "sample": "%d threads x %d FD columns each over the %s, %.1f s" % (cores, per, what, ta),
            "speedup_over_1_core": (nE * per * cores / ta) / (nE * n_pass / t)}
    return one, allc, parity


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: start N copies of this script, one rank per GPU (RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_ADDR / MASTER_PORT exported as torch.distributed.run would), let rank 0 own stdout (the one JSON
    line), send the other ranks' stdout to stderr, and return the worst exit code.  A rank that fails takes the others
    down (exact PIDs); the whole job is bounded by GST_BENCH_TIMEOUT seconds (default 1500)."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else sys.stderr))
    deadline = time.monotonic() + float(os.environ.get("GST_BENCH_TIMEOUT", "1500"))
    rc = 0
    alive = list(procs)
    while alive:
        time.sleep(0.2)
        for p in list(alive):
            code = p.poll()
            if code is not None:
                alive.remove(p)
                if code != 0 and rc == 0:
                    rc = code
        if alive and (rc != 0 or time.monotonic() > deadline):
            if rc == 0:
                rc = 124
                print("[bench] ranks still running at the deadline: stopping them", file=sys.stderr, flush=True)
            for p in alive:
                p.terminate()
            t_kill = time.monotonic() + 10
            for p in alive:
                try:
                    p.wait(max(0.1, t_kill - time.monotonic()))
                except subprocess.TimeoutExpired:
                    p.kill()
            break
    return rc


def fd_roofline(flops_schedule, fd_exec, k_ms, jac_bytes, traffic):
    """The `roofline` object of the FD headline.  `frac` prices what the kernel ISSUES (gst_get_fd_work walks the plan's
    programs with the kernel's own clean/dirty rule) against the dense fp64 peak; the reference schedule's flops (SURVEY 8(d):
    nP*(2*D^2*A + 2*D*nE), 42 % of which the kernel skips as provably bit-identical to the base pass) are `frac_schedule`.
    Flat scalars and short strings only: the driver's record keeps those."""
    sec = k_ms * 1e-3
    issued = fd_exec["issued_flops_per_launch"] if fd_exec else flops_schedule
    r = {"bound": "mfma", "compute_unit": "valu_f64", "kernel": "walk_kernel<16,1>",
         "achieved": issued / sec / 1e12, "peak": F64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
         "frac": issued / sec / 1e12 / F64_VALU_PEAK_TFLOPS,
         "frac_schedule": flops_schedule / sec / 1e12 / F64_VALU_PEAK_TFLOPS,
         "frac_of_unfused_ceiling": issued / sec / 1e12 / (0.5 * F64_VALU_PEAK_TFLOPS),
         "kernel_ms": k_ms, "flops_per_launch": issued, "flops_schedule_per_launch": flops_schedule,
         "executed_over_schedule": (fd_exec or {}).get("executed_over_schedule"),
         "mul_add_share_of_pmc_valu": (fd_exec or {}).get("mul_add_share_of_pmc_valu"),
         "hbm_write_GBps": jac_bytes / sec / 1e9, "hbm_peak_GBps": HBM_PEAK_GBS, "traffic": traffic,
         "bound_note": "compute roof = dense fp64 78.6 TFLOP/s (MFMA and vector FMA alike); the walk runs on the fp64 VALU",
         "note": "frac = issued mul+add flops (no FMA: bit parity, own ceiling 0.5); frac_schedule = SURVEY 8(d) flops incl. skipped",
         "executed": fd_exec}
    return r


_T0 = time.perf_counter()


def log(msg):
    """Progress on stderr (stdout carries the one JSON line)."""
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench %7.1fs] %s" % (time.perf_counter() - _T0, msg), file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--design", default="full", choices=["full", "lite"])
    ap.add_argument("--max-len", type=int, default=1024)
    ap.add_argument("--target-tasks", type=int, default=0)
    ap.add_argument("--max-slots", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--jtj", action="store_true", help="also time J^T J and J^T f on the resident Jacobian (row f1); "
                    "N>1: plus the all-reduce of the partial products between device buffers (RCCL)")
    ap.add_argument("--no-jacobian-gather", action="store_true",
                    help="N>1: skip the secondary timing of the fan-in of the Jacobian row blocks to rank 0")
    ap.add_argument("--no-host-fill", action="store_true", help="N=1: skip the secondary end-to-end fill into a host array")
    ap.add_argument("--scaling", default="strong", choices=["weak", "strong"],
                    help="N>1: weak = one full design per rank (N designs in all); strong = one design dealt to N atoms")
    ap.add_argument("--no-analytic", action="store_true", help="skip the secondary analytic-derivative timing")
    ap.add_argument("--keep-zeros", action="store_true", help="analytic legs: GST_OPT_ANALYTIC_KEEP_ZEROS = 1 (the caller's promise for ANY destination; the default, 2, already covers the tracked device memory this bench fills)")
    ap.add_argument("--store-zeros", action="store_true", help="analytic legs: GST_OPT_ANALYTIC_KEEP_ZEROS = 0 (every fill stores the structural zeros again: the round-3 default)")
    ap.add_argument("--no-other-configs", action="store_true", help="N=1: skip the secondary legs of the other BASELINE configurations (1Q, 3Q, Hessian block)")
    ap.add_argument("--no-lm-step", action="store_true", help="skip the secondary LM-iteration timing (fill + objective maps + J^T J + J^T f [+ all-reduce])")
    ap.add_argument("--no-fit-replay", action="store_true", help="N=1: skip the replay of the recorded 2Q L<=64 GST fit (the reference's only published end-to-end workload)")
    ap.add_argument("--no-cptplnd", action="store_true", help="N=1: skip the secondary CPTPLND (Lindblad-parameterised) Jacobian timing")
    ap.add_argument("--emulate-ranks", type=int, default=0,
                    help="development aid (N=1 only): time rank 0's atom of an N-atom strong-scaling layout on this one "
                         "GPU; the printed value counts only that atom's elements")
    ap.add_argument("--grid", default="", help="processor grid NAxNP (strong scaling): NA atom-processors, each split into NP "
                    "parameter-processors that take a column slice of the same atom (distforwardsim.py:445-485); NA*NP = --gpus")
    ap.add_argument("--emulate-rank", type=int, default=0, help="with --emulate-ranks: which rank's share to run")
    ap.add_argument("--deriv", default="fd", choices=["fd", "analytic"],
                    help="fd: finite differences, bit-identical to the reference Map path (headline); "
                         "analytic: exact derivatives (MatrixForwardSimulator semantics)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))           # plain `python bench.py --gpus N`: become the launcher of N ranks
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit("bench.py --gpus %d started inside a job of WORLD_SIZE=%d: launch it with --nproc-per-node %d, or with no "
                 "launcher at all (it then starts its own ranks)" % (args.gpus, world, args.gpus))

    from pygsti_amd import _lib, dist as gdist
    device = local_rank % max(_lib.device_count(), 1)      # (ranks share a GPU only in 1-GPU plumbing runs)
    # control plane: a gloo group (rendezvous, barriers, max-over-ranks); data plane: the C ABI's device communicator
    # (RCCL over xGMI; the IPC transport when RCCL cannot start, e.g. ranks sharing a GPU)
    ctx = gdist.init(device=device, transport="auto") if world > 1 else gdist.DistContext(0, 1, local_rank, None, None)
    comm = ctx.comm

    def barrier_sync(plan):
        plan.sync()
        ctx.barrier()

    lay_world, lay_rank = (args.emulate_ranks, args.emulate_rank) if (world == 1 and args.emulate_ranks > 1) else (world, rank)
    # The other BASELINE configurations (1Q L<=128, the 3-qubit D = 64 model) run FIRST, while theirs are the only plans of
    # the process: blocking fills of a launch-bound plan measure 2-3x longer (130 / 288 us instead of 50 / 105 us) once
    # the 2Q design's plan, its streams and its 7 GB of buffers exist beside them -- a property of the runtime's queues,
    # not of the kernels (tools/ab_1q.sh).
    other_configs, bc = None, None
    if world == 1 and lay_world == 1 and not args.no_other_configs:
        import importlib.util
        spec = importlib.util.spec_from_file_location("bench_configs", os.path.join(ROOT, "tools", "bench_configs.py"))
        bc = importlib.util.module_from_spec(spec); spec.loader.exec_module(bc)
        other_configs = {}
        for key, fn in (("smq1Q_XYI_L128", bc.one_q), ("3Q_D64", bc.three_q)):
            try:
                other_configs[key] = fn()
            except Exception as e:                       # a secondary leg must not take the headline down with it
                other_configs[key] = {"error": "%s: %s" % (type(e).__name__, e)}
            log("config %s done" % key)
    grid = tuple(int(x) for x in args.grid.lower().split("x")) if args.grid else None
    if grid and (len(grid) != 2 or grid[0] * grid[1] != lay_world or args.scaling != "strong"):
        sys.exit("--grid NAxNP needs NA*NP == %d ranks and strong scaling" % lay_world)
    col_split = bool(grid) and grid[1] > 1
    if col_split:         # the secondary legs below assume whole rows on every rank
        args.no_analytic = args.no_jacobian_gather = True
    pack, model, circuits, layout = build_workload(args.design, args.max_len, lay_world, lay_rank, device,
                                                    args.target_tasks, args.max_slots, args.scaling, grid, args.deriv)
    log("workload built: %d circuits" % len(circuits))
    atom = layout.atoms[0]
    plan = atom.plan()
    log("plan compiled")
    gates, rhos, effects = layout.model_arrays(model)
    plan.set_model(gates, rhos, effects)
    plan.set_param_map(*layout.param_map(model))
    nP = model.num_params
    nE_local = atom.num_elements
    nE_total = layout.global_num_elements * (world if args.scaling == "weak" else 1)
    if lay_world != world:
        nE_total = nE_local                       # --emulate-ranks: one atom's share only
    n_circ_total = len(circuits) * (world if args.scaling == "weak" else 1)
    # row blocks of the assembled element dimension: (owner rank, first row, rows)
    if world > 1 and args.scaling == "strong":
        blocks = gdist.row_blocks(layout, world)
        row0 = atom.element_slice.start
    elif world > 1:
        blocks = [(r, r * nE_local, nE_local) for r in range(world)]
        row0 = rank * nE_local
    else:
        blocks, row0 = [(0, 0, nE_local)], 0
    # this rank's parameter columns: all of them, or its parameter-processor's slice of the atom it shares (--grid)
    gps = layout.global_param_slice
    nP_local = gps.stop - gps.start
    d_out = plan.device_malloc(nE_local * nP_local * 8, tracked=True)    # this rank's Jacobian block (stays distributed, as in the reference); TRACKED: only the library writes it
    d_pfull = plan.device_malloc(nE_total * 8)             # the ASSEMBLED probabilities: own rows filled in place
    d_probs = d_pfull + row0 * 8
    pidx = np.arange(gps.start, gps.stop, dtype=np.int64)

    mode = _lib.DERIV_ANALYTIC if args.deriv == "analytic" else _lib.DERIV_FD
    if args.keep_zeros or args.store_zeros:
        plan.set_option(_lib.OPT_ANALYTIC_KEEP_ZEROS, 0 if args.store_zeros else 1)
    exchange = None
    if world > 1:
        exchange = {"transport": ctx.transport, "rccl_ranks": world if ctx.transport == "rccl" else 0,
                    "rccl_version": comm.info()["rccl_version"] if comm is not None else 0,
                    "what": "every step: probability row blocks all-gathered between device buffers (Allgatherv of "
                            "resourceallocation.py:316-348); Jacobian rows stay on their ranks as bulk_fill_dprobs leaves them",
                    "fallback_reason": ctx.comm_error}
    host_stage = np.empty(nE_total) if (world > 1 and comm is None) else None
    # N > 1, what `north_star` asks for: the Jacobian blocks end up on rank 0.  Rank 0's assembled [nE][nP] array is mapped into
    # every rank (gst_comm_map_root_buffer) and each fill writes its rows THERE -- the kernel's stores cross xGMI while it runs,
    # there is no separate gather pass.  When the mapping is not available the rows stay on their ranks (the reference's own
    # bulk_fill_dprobs leaves them distributed, too) and the line says so.
    d_jfull, d_direct = None, None
    if world > 1 and comm is not None and not args.no_jacobian_gather and not col_split:
        ok = 1.0
        try:
            if rank == 0:
                d_jfull = plan.device_malloc(nE_total * nP * 8)
            mapped = comm.map_root_buffer(d_jfull or 0, 0)
            d_direct = mapped + row0 * nP * 8
        except Exception as e:
            ok = 0.0
            exchange["jacobian_fan_in_error"] = "%s: %s" % (type(e).__name__, str(e)[:200])
        if ctx.max_over_ranks(1.0 - ok) > 0.0:          # (all ranks or none)
            d_direct = None
        exchange["jacobian_fan_in"] = ("every fill writes its row block straight into rank 0's assembled array (peer-mapped over xGMI, "
                                       "gst_comm_map_root_buffer): included in `value`") if d_direct else "not available: rows stay on their ranks"

    def exchange_probs():
        if world == 1:
            return
        if comm is not None:
            comm.allgather_rows(d_pfull, 1, blocks, plan)          # stream-ordered behind the fill (RCCL)
        elif args.scaling == "strong":                             # no device transport at all: host-staged, flagged
            plan.memcpy_d2h(host_stage[row0:row0 + nE_local], d_probs)
            plan.memcpy_h2d(d_pfull, gdist.gather_elements(host_stage, layout))

    def step(dst=None):
        plan.set_model(gates, rhos, effects)          # from_vector -> new dense arrays -> H2D
        plan.fill_dprobs_dev(dst or d_direct or d_out, nP_local, pidx, None, 1e-7, d_probs, mode)
        exchange_probs()

    for _ in range(args.warmup):
        step()
    barrier_sync(plan)
    log("warm-up done")
    t0 = time.perf_counter()
    kernel_ms = []
    for _ in range(args.steps):
        step()
        if world == 1:
            plan.sync()                                # N=1: per-step HIP-event read-out (stream stays ordered)
            kernel_ms.append(plan.stats()["last_kernel_ms"])
    plan.sync()
    t_local = time.perf_counter() - t0             # this rank's own K steps (the max over ranks, after the barrier, is `dt`)
    barrier_sync(plan)
    dt = ctx.max_over_ranks(time.perf_counter() - t0)
    if not kernel_ms:
        kernel_ms = [plan.stats()["last_kernel_ms"]]
    per_rank_ms = ctx.all_floats(1e3 * t_local / args.steps)
    per_rank_kernel_ms = ctx.all_floats(float(np.mean(kernel_ms)))

    if world > 1:
        # every rank must now hold every circuit's probabilities: they sum to 1 per circuit
        pf = plan.memcpy_d2h(np.empty(nE_total), d_pfull)
        assert abs(pf.sum() - n_circ_total) < 1e-6 * n_circ_total, "assembled probabilities must sum to 1 per circuit"

    log("timed steps done: %.2f ms/step" % (1e3 * dt / args.steps))
    value_rows_distributed = None
    if d_direct:
        if rank == 0:               # the assembled Jacobian is complete on rank 0: every row finite, the last block included
            chk = plan.memcpy_d2h(np.empty(nP), d_jfull + ((nE_total - 1) * nP) * 8)
            assert np.isfinite(chk).all(), "the assembled Jacobian on rank 0 has unwritten rows"
        for _ in range(2):
            step(d_out)
        barrier_sync(plan)
        td0 = time.perf_counter()
        for _ in range(args.steps):
            step(d_out)
        barrier_sync(plan)
        value_rows_distributed = nE_total * nP * args.steps / ctx.max_over_ranks(time.perf_counter() - td0)
    # secondary: probabilities only (+ their all-gather)
    for _ in range(2):
        plan.fill_probs_dev(d_probs)
    barrier_sync(plan)
    tp0 = time.perf_counter()
    n_pr = max(3, args.steps)
    for _ in range(n_pr):
        plan.set_model(gates, rhos, effects)
        plan.fill_probs_dev(d_probs)
        exchange_probs()
    barrier_sync(plan)
    dtp = ctx.max_over_ranks(time.perf_counter() - tp0)
    # ---- secondary legs (tools/bench_legs.py): nothing below is part of the timed region -----------------------------------
    from types import SimpleNamespace
    import importlib.util
    _spec = importlib.util.spec_from_file_location("bench_legs", os.path.join(ROOT, "tools", "bench_legs.py"))
    legs = importlib.util.module_from_spec(_spec); _spec.loader.exec_module(legs)
    B = SimpleNamespace(args=args, plan=plan, ctx=ctx, comm=comm, world=world, rank=rank, lay_world=lay_world, layout=layout, model=model,
                        pack=pack, gates=gates, rhos=rhos, effects=effects, d_out=d_out, d_probs=d_probs, d_pfull=d_pfull, pidx=pidx, nP=nP,
                        nP_local=nP_local, nE_local=nE_local, nE_total=nE_total, mode=mode, blocks=blocks, row0=row0, exchange=exchange,
                        grid=grid, col_split=col_split, barrier_sync=barrier_sync, exchange_probs=exchange_probs, log=log, n_pr=n_pr,
                        HBM_PEAK_GBS=HBM_PEAK_GBS)
    fast_info = legs.fast_probs(B)          # the probability fill through the log-depth level pass (GST_OPT_FAST_PROBS)
    ana_info = legs.analytic(B)             # the same Jacobian by exact derivatives (MatrixForwardSimulator semantics)
    cptp_info = legs.cptplnd(B)             # the same design under the CPTPLND parameterisation (device-built Lindblad members)
    log("probs / analytic legs done")
    legs.jacobian_gather(B)                 # N > 1: the Jacobian fan-in to rank 0 (adds exchange['jacobian_gather_to_rank0'])
    # secondary (N=1): the other BASELINE configurations, each with its own roofline fraction -- 1Q L<=128 (configs[1]),
    # the 3-qubit D = 64 model (configs[4]: FD block, full analytic Jacobian, a Hessian block) and one rectangle of the 2Q
    # objective Hessian (FD of FD) -- so that they are driver-timed figures, not builder-only ones
    if other_configs is not None:
        for key, fn in (("2Q_objective_hessian_block", lambda: bc.two_q_hessian_block(plan, nE_local, nP, plan.stats()["applies_per_pass"])),):
            try:
                other_configs[key] = fn()
            except Exception as e:                       # a secondary leg must not take the headline down with it
                other_configs[key] = {"error": "%s: %s" % (type(e).__name__, e)}
            log("config %s done" % key)
        plan.set_model(gates, rhos, effects)
        plan.fill_dprobs_dev(d_out, nP, pidx, None, 1e-7, d_probs, mode)      # leave the headline Jacobian resident
        barrier_sync(plan)

    log("exchange legs done")
    lm_info = legs.lm_step(B)               # one Levenberg-Marquardt iteration's device work end to end
    jtj_info = legs.normal_equations(B)     # --jtj: the normal equations in detail (column-distributed grid or per-kernel timings)
    log("normal-equation leg done")
    host_fill = legs.host_fill(B)           # the reference API end to end: bulk_fill_dprobs into the caller's HOST array
    # the reference's only published end-to-end figure (mpi_2D_scaling: smq2Q L<=64 GST, 3,113 s on one rank): every LM iteration
    # of a recorded run of it replayed on the device, checked against checksums of the reference's own normal equations
    fit_info = None
    if world == 1 and lay_world == 1 and not args.no_fit_replay:
        try:
            _fs = importlib.util.spec_from_file_location("fit_replay2q", os.path.join(ROOT, "tools", "fit_replay2q.py"))
            _fm = importlib.util.module_from_spec(_fs); _fs.loader.exec_module(_fm)
            fit_info = _fm.replay(device=device)
        except Exception as e:                       # a secondary leg must not take the headline down with it
            fit_info = {"error": "%s: %s" % (type(e).__name__, e)}
        log("fit replay leg done")
    def measured_traffic(kernel_prefix, section=None):
        """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes of THIS workload
        (profiles/r0x_hbm_counters.json: FETCH_SIZE and WRITE_SIZE collected in separate passes; KB -> bytes, and
        FETCH_SIZE doubled, the gfx950 correction of MI355X_MICROARCH.md).  None when no profile matches."""
        if world != 1 or lay_world != 1 or args.design != "full" or args.max_len != 1024:
            return None
        for name in ("r06_hbm_counters.json", "r05_hbm_counters.json", "r04_hbm_counters.json", "r03_hbm_counters.json", "r02_hbm_counters.json", "r01_hbm_counters.json"):
            try:
                with open(os.path.join(ROOT, "profiles", name)) as f:
                    prof = json.load(f)[section or args.deriv]
                for k, v in prof.items():
                    if kernel_prefix in k:
                        return (2.0 * v["FETCH_SIZE_KB_per_launch"] + v["WRITE_SIZE_KB_per_launch"]) * 1024.0
            except Exception:
                pass
        return None

    def measured_valu_instructions(kernel_prefix):
        """SQ_INSTS_VALU per launch of the dominant kernel from the newest committed PMC pass of this same workload
        (profiles/r*_bench_pmc_sq_current.json), with its file name -- the cross-check of `executed`."""
        if world != 1 or lay_world != 1 or args.design != "full" or args.max_len != 1024:
            return None
        for name in ("r06_bench_pmc_sq_current.json", "r05_bench_pmc_sq_current.json", "r04_bench_pmc_sq_current.json"):
            try:
                with open(os.path.join(ROOT, "profiles", name)) as f:
                    prof = json.load(f)[args.deriv]
                for k, v in prof.items():
                    if kernel_prefix in k:
                        return float(v["SQ_INSTS_VALU"]), "profiles/" + name
            except Exception:
                pass
        return None

    st = plan.stats()
    if rank == 0:
        D = model.dim
        k_ms = float(np.mean(kernel_ms))
        # algorithmic flops of one FD Jacobian on the schedule actually executed by this rank:
        # (nP perturbed passes) x (2 D^2 per gate application + 2 D per element)
        flops = nP_local * (2.0 * D * D * st["applies_per_pass"] + 2.0 * D * nE_local)
        jac_bytes = 8.0 * nE_local * nP_local
        if args.deriv == "analytic":
            roof = {"bound": "hbm", "kernel": "analytic_mfma_kernel (+ the backward chain pass that feeds it)",
                    "achieved": jac_bytes / (k_ms * 1e-3) / 1e9,
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": jac_bytes / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "note": "algorithmic bytes = the Jacobian write 8*nE*nP; the kernel gathers forward and backward state vectors (20 GB of L2 traffic, of which the PMC `traffic` figure is what reaches HBM)",
                    "kernel_ms": k_ms, "bytes_per_launch": jac_bytes, "traffic": measured_traffic("analytic_mfma_kernel")}
            if roof["traffic"]:
                roof["traffic_over_algorithmic"] = roof["traffic"] / jac_bytes
                roof["traffic_frac_of_peak"] = roof["traffic"] / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
        else:
            roof = None
        fd_exec = None
        if roof is None and D <= 16:
            # what the kernel really executes (gst_get_fd_work: the plan's programs walked with the kernel's own clean/dirty
            # rule), so that the fraction of the roof can be read for the schedule AND for the arithmetic actually issued
            w = plan.fd_work(pidx)
            issued = 64.0 * (2.0 * D * D * w["wave_applies_executed"] + 2.0 * D * w["wave_dots_executed"])
            fd_exec = {"executed_applications": w["col_applies_executed"], "schedule_applications": w["col_applies_schedule"],
                       "executed_over_schedule": w["col_applies_executed"] / max(w["col_applies_schedule"], 1),
                       "wavefront_applications_executed": w["wave_applies_executed"], "wavefront_dots_executed": w["wave_dots_executed"],
                       "wavefronts": w["n_waves"], "issued_flops_per_launch": issued,
                       "frac_executed": issued / (k_ms * 1e-3) / 1e12 / F64_VALU_PEAK_TFLOPS,
                       "frac_of_unfused_ceiling": issued / (k_ms * 1e-3) / 1e12 / (0.5 * F64_VALU_PEAK_TFLOPS),
                       "mul_add_wave_instructions": w["wave_applies_executed"] * 2 * D * D + w["wave_dots_executed"] * 2 * D}
            pmc = measured_valu_instructions("walk_kernel<16, 1")
            if pmc:
                fd_exec["pmc_SQ_INSTS_VALU_per_launch"] = pmc[0]
                fd_exec["pmc_source"] = pmc[1]
                fd_exec["mul_add_share_of_pmc_valu"] = fd_exec["mul_add_wave_instructions"] / pmc[0]
        out = {
            "metric": "dprobs Jacobian-elements/sec, 2Q GST L<=1024 (bulk_fill_dprobs, %s)" % (
                "FD eps=1e-7, bit-identical to the reference Map path" if args.deriv == "fd" else
                "analytic derivatives, MatrixForwardSimulator semantics, <=1e-8"),
            "value": nE_total * (nP if lay_world == world else nP_local) * args.steps / dt,
            "unit": "Jacobian-elements/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "smq2Q_XYICNOT GST L<=%d %s germs: %d circuits, nE=%d, nP=%d, D=%d; "
                                   "target model depolarized 0.01/0.01%s" % (
                                       args.max_len, args.design, len(circuits), layout.global_num_elements, nP, D,
                                       "" if world == 1 else (" -- x%d designs, one per rank (%d circuits in all)" % (world, n_circ_total)
                                                              if args.scaling == "weak" else " -- dealt to %d atoms" % world)),
                       "destination": ("rank 0's assembled HBM-resident Jacobian: every rank's fill writes its rows there over xGMI (gst_comm_map_root_buffer)"
                                       if d_direct else
                                       "device (HBM-resident Jacobian, gst_fill_dprobs_dev); the reference API's host-array fill is the `host_fill` leg (PCIe-bound)"),
                       "derivative": ("forward finite differences, eps=1e-7 (reference MapForwardSimulator semantics)"
                                      if args.deriv == "fd" else "analytic (reference MatrixForwardSimulator semantics)"),
                       "parallelism": (("atoms%d" % world) if not grid else ("grid %dx%d (atom-processors x parameter-processors)" % grid))
                                      if lay_world == world else
                                      "rank %d of %s emulated on one GPU" % (lay_rank, ("atoms%d" % lay_world) if not grid else "grid %dx%d" % grid)},
            "per_rank": {"ms_per_step": per_rank_ms, "dominant_kernel_ms": per_rank_kernel_ms,
                         "note": "each rank's own time for the K steps before the closing barrier; `ms_per_step` is the max over ranks after it"},
            "exchange": exchange,
            # the north-star variant of the N > 1 figure: every step ALSO fans the Jacobian row blocks in to rank 0 over xGMI
            # (`value` leaves them distributed, as the reference's bulk_fill_dprobs does); None at N = 1 or when the leg is off
            "value_incl_jacobian_gather": (exchange or {}).get("jacobian_gather_to_rank0", {}).get("elements_per_s_with_gather"),
            # N > 1 with the fan-in inside `value`: the same K steps with every rank's rows left on its own GPU (the reference's layout)
            "value_rows_distributed": value_rows_distributed,
            "lm_step": lm_info,
            "normal_equations": jtj_info,
            "analytic_dprobs": ana_info,
            "cptplnd_dprobs": cptp_info,
            "other_configs": other_configs,
            "gst_fit_2Q_L64": fit_info,
            "host_fill": host_fill,
            "probs_per_s": nE_total * n_pr / dtp,
            "probs_ms": 1e3 * dtp / n_pr,
            "fast_probs": fast_info,
            "probs_roofline": {"bound": "mfma", "compute_unit": "valu_f64", "unit": "TFLOP/s", "peak": F64_VALU_PEAK_TFLOPS,
                               "achieved": (2.0 * D * D * st["applies_per_pass"] + 2.0 * D * nE_local) / (dtp / n_pr) / 1e12,
                               "frac": (2.0 * D * D * st["applies_per_pass"] + 2.0 * D * nE_local) / (dtp / n_pr) / 1e12 / F64_VALU_PEAK_TFLOPS,
                               "note": "one probability pass is a latency chain -- %d tasks of ~%d dependent mat-vecs on 1,024 SIMDs, 221 ns per step -- not a throughput kernel: the fraction of the compute roof SURVEY 8(d) names is what bit-exact sequential order leaves" % (st["n_tasks"], st["applies_per_pass"] // max(st["n_tasks"], 1))},
            "roofline": roof or fd_roofline(flops, fd_exec, k_ms, jac_bytes, measured_traffic("walk_kernel<16, 1")),
            "plan": {k: st[k] for k in ("n_circuits", "n_elements", "sum_depth", "trie_nodes", "applies_per_pass",
                                        "n_tasks", "prog_words", "max_slots")},
        }
        # flat copies of the figures a reader should not have to open profiles/ for, inside the objects the driver's record keeps
        rf = out["roofline"]
        if ana_info and ana_info.get("roofline"):
            ar = ana_info["roofline"]
            tr = measured_traffic("analytic_mfma_kernel", "analytic")
            rf["analytic_kernel"] = "analytic_mfma_kernel"
            rf["analytic_kernel_ms"] = ana_info["kernel_ms"]
            rf["analytic_frac_hbm"] = ar["frac"]
            rf["analytic_whole_step_frac_hbm"] = ar["whole_step_frac"]
            rf["analytic_bytes_algorithmic"] = ar["bytes_per_launch"]
            rf["analytic_traffic"] = tr
            rf["analytic_traffic_frac_of_peak"] = (tr / (ana_info["kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS) if tr and ana_info["kernel_ms"] else None
            rf["analytic_zeros_resident"] = ana_info["structural_zeros"].startswith("resident")
            # the practical roof of a store stream on this part, measured here (hipMemsetAsync over 4 GiB), next to the 8 TB/s read peak
            ceil = legs.store_only_ceiling(device) if world == 1 else None
            rf["store_only_ceiling_GBps"] = ceil
            rf["analytic_frac_of_store_ceiling"] = (ar["achieved"] / ceil) if (ceil and ar.get("achieved")) else None
            rf["analytic_traffic_frac_of_store_ceiling"] = (tr / (ana_info["kernel_ms"] * 1e-3) / 1e9 / ceil) if (ceil and tr and ana_info["kernel_ms"]) else None
        if host_fill:
            out["config"]["host_fill_elements_per_s"] = host_fill.get("elements_per_s")
            out["config"]["host_fill_ms"] = host_fill.get("ms")
        if lm_info:
            out["config"]["lm_step_ms"] = lm_info.get("ms_per_step")
        if fit_info and "error" not in fit_info:
            out["config"]["fit_2Q_L64_iterations"] = fit_info["iterations"]
            out["config"]["fit_2Q_L64_device_ms_sum"] = fit_info["device_ms_sum"]
            out["config"]["fit_2Q_L64_worst_checksum_deviation"] = fit_info["worst"]
            out["config"]["fit_2Q_L64_reference_s"] = "published 3113 s (1 rank); %.0f s in the build container, %.0f s of it inside these dlsvec calls" % (
                fit_info["reference_run_seconds_build_container"], fit_info["reference_dlsvec_seconds_same_calls"])
        if world == 1 and not args.no_cpu_baseline:
            pctx = None
            if lay_world == 1 and args.deriv == "fd":
                # parity where the number is quoted: the headline Jacobian is (re)filled, then checked against the checker
                plan.set_model(gates, rhos, effects)
                plan.fill_dprobs_dev(d_out, nP, pidx, None, 1e-7, d_probs, mode)
                plan.sync()
                pctx = {"plan": plan, "nE": nE_local, "nP": nP, "d_probs": d_probs, "d_out": d_out,
                        "param_map": layout.param_map(model),
                        "cols": parity_columns(D, len(gates), len(effects), nP)}
            out["cpu_baseline"], out["cpu_baseline_allcores"], out["parity"] = cpu_baseline(pack, model, args.max_len, args.design, pctx)
            cb = out["cpu_baseline"]
            if out["parity"]:
                cb["parity_bitwise"] = out["parity"]["bitwise"]
                cb["parity_rows_x_cols"] = "%d x %d" % (out["parity"]["n_rows"], out["parity"]["n_cols"])
            cb["allcores_value"] = out["cpu_baseline_allcores"]["value"]
            cb["allcores_cores"] = out["cpu_baseline_allcores"]["cores"]
            # the REAL Cython path (BASELINE.md / SURVEY App. C: probed in the build container, 1 Xeon core @2.1 GHz, lite design,
            # per-column model.set_parameter_value Python included) -- context for the C++-reps-under-a-restated-loop figure above
            cb["reference_cython_probe_value"] = 1.66e6
            cb["reference_cython_probe_note"] = "BASELINE.md: real Cython bulk_fill_dprobs, 2Q L<=1024 lite, 1 core, build container"
            cb["driver_loop"] = "restated (oracle/ref_driver.cpp) over the reference's own C++ reps; no per-column Python"
            log("cpu baseline done")
        print(json.dumps(out))
    if world > 1:
        ctx.barrier()               # (nobody frees the mapped array while a peer may still write into it)
    if d_jfull:
        plan.device_free(d_jfull)
    plan.device_free(d_out)
    plan.device_free(d_pfull)
    if world > 1:
        ctx.barrier()
        with gdist._StdoutToStderr():          # (communicator teardown may print as well)
            ctx.shutdown()


if __name__ == "__main__":
    main()
