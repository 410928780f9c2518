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


def build_workload(design, max_len, world, rank, device, target_tasks, max_slots=0, scaling="strong"):
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
        layout = HipCOPALayout(circuits, model, num_atoms=world, devices=[device], rank=rank, size=world,
                               target_tasks=target_tasks, max_slots=max_slots)
    return pack, model, circuits, layout


def cpu_baseline(pack, model, max_len):
    """Time the CPU checker on a bounded sample: the `lite` sub-design (15 of the 88 germs; every one of its
    circuits is in the full design), walking the reference-format prefix table for a fixed number of passes
    (one pass = one finite-difference column)."""
    from oracle import oracle as O, prefix_table as PT
    circuits = pack.create_gst_circuits(max_len, lite=True)
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
    mdl = dict(gates=G, rhos=R, effects=E, pkind=-np.ones(nP, np.int32), pobj=np.zeros(nP, np.int32),
               pelem=np.zeros(nP, np.int32))
    kind = "reference" if os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libgst_ref.so")) else "port"
    orc = O.Oracle(tbl, mdl, kind)
    t1 = orc.time_passes(4)                      # calibrate
    n_pass = int(max(8, min(400, 12.0 / max(t1 / 4, 1e-6))))
    t = orc.time_passes(n_pass)
    nE = tbl["nE"]
    return {"value": nE * n_pass / t, "unit": "Jacobian-elements/s", "cores": 1, "kind": kind,
            "sample": "%d FD columns (probability passes) over the lite sub-design (%d of the workload's circuits, "
                      "nE=%d, %d gate applications per pass on the reference prefix table), %.1f s"
                      % (n_pass, len(circuits), nE, len(tbl["gate_idx"]), t),
            "probs_per_s": nE * n_pass / t}


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
    ap.add_argument("--jtj", action="store_true", help="also time J^T J and J^T f on the resident Jacobian (row f1)")
    ap.add_argument("--gather", action="store_true", help="N>1: also all-gather the probability row blocks (RCCL)")
    ap.add_argument("--scaling", default="strong", choices=["weak", "strong"],
                    help="N>1: weak = one full design per rank (N designs in all); strong = one design dealt to N atoms")
    ap.add_argument("--no-analytic", action="store_true", help="skip the secondary analytic-derivative timing")
    ap.add_argument("--emulate-ranks", type=int, default=0,
                    help="development aid (N=1 only): time rank 0's atom of an N-atom strong-scaling layout on this one "
                         "GPU; the printed value counts only that atom's elements")
    ap.add_argument("--deriv", default="fd", choices=["fd", "analytic"],
                    help="fd: finite differences, bit-identical to the reference Map path (headline); "
                         "analytic: exact derivatives (MatrixForwardSimulator semantics)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
    dist = None
    backend = None
    if world > 1:
        import torch
        import torch.distributed as dist
        ndev = max(torch.cuda.device_count(), 1)
        device = local_rank % ndev                  # (ranks share a GPU only in the 1-GPU plumbing test)
        torch.cuda.set_device(device)
        backend = os.environ.get("GST_BENCH_BACKEND", "nccl")     # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group(backend)
        local_rank = device

    def barrier_sync(plan):
        plan.sync()
        if dist is not None:
            import torch
            torch.cuda.synchronize()
            dist.barrier()

    lay_world, lay_rank = (args.emulate_ranks, 0) if (world == 1 and args.emulate_ranks > 1) else (world, rank)
    pack, model, circuits, layout = build_workload(args.design, args.max_len, lay_world, lay_rank, local_rank,
                                                    args.target_tasks, args.max_slots, args.scaling)
    atom = layout.atoms[0]
    plan = atom.plan()
    gates, rhos, effects = layout.model_arrays(model)
    plan.set_model(gates, rhos, effects)
    plan.set_param_map(*layout.param_map(model))
    nP = model.num_params
    nE_local = atom.num_elements
    nE_total = layout.global_num_elements * (world if args.scaling == "weak" else 1)
    if lay_world != world:
        nE_total = nE_local                       # --emulate-ranks: one atom's share only
    n_circ_total = len(circuits) * (world if args.scaling == "weak" else 1)
    d_out = plan.device_malloc(nE_local * nP * 8)
    d_probs = plan.device_malloc(nE_local * 8)
    pidx = np.arange(nP, dtype=np.int64)

    from pygsti_amd import _lib
    mode = _lib.DERIV_ANALYTIC if args.deriv == "analytic" else _lib.DERIV_FD

    def step():
        plan.set_model(gates, rhos, effects)          # from_vector -> new dense arrays -> H2D
        plan.fill_dprobs_dev(d_out, nP, pidx, None, 1e-7, d_probs, mode)

    for _ in range(args.warmup):
        step()
    barrier_sync(plan)
    t0 = time.perf_counter()
    kernel_ms = []
    for _ in range(args.steps):
        step()
        if world == 1:
            plan.sync()                                # N=1: per-step HIP-event read-out (stream stays ordered)
            kernel_ms.append(plan.stats()["last_kernel_ms"])
    barrier_sync(plan)
    dt = time.perf_counter() - t0
    if not kernel_ms:
        kernel_ms = [plan.stats()["last_kernel_ms"]]
    if dist is not None:
        import torch
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # secondary: probabilities only
    for _ in range(2):
        plan.fill_probs_dev(d_probs)
    barrier_sync(plan)
    tp0 = time.perf_counter()
    n_pr = max(3, args.steps)
    for _ in range(n_pr):
        plan.set_model(gates, rhos, effects)
        plan.fill_probs_dev(d_probs)
    barrier_sync(plan)
    dtp = time.perf_counter() - tp0

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
        barrier_sync(plan)
        dta = time.perf_counter() - ta0
        if dist is not None:
            import torch
            tt = torch.tensor([dta], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dta = float(tt.item())
        ana_info = {"value": nE_total * nP * n_an / dta, "unit": "Jacobian-elements/s", "ms_per_step": 1e3 * dta / n_an,
                    "kernel_ms": plan.stats()["last_kernel_ms"],
                    "note": "analytic derivatives (reference MatrixForwardSimulator semantics); secondary figure, not `value`"}
        plan.fill_dprobs_dev(d_out, nP, pidx, None, 1e-7, d_probs, mode)      # leave the FD Jacobian resident for --jtj
        barrier_sync(plan)

    jtj_info = None
    if args.jtj:
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
        if dist is not None:
            # the path's one real exchange step: every rank holds the partial J^T J of its rows; the optimizer needs
            # the sum -> one all-reduce of nP^2 doubles (RCCL over xGMI under nccl), cf. distlayout.py:1259,1355
            import torch
            host = np.empty((nP, nP)); plan.memcpy_d2h(host, d_jtj)
            tj = torch.from_numpy(host).to("cuda" if backend == "nccl" else "cpu")
            barrier_sync(plan)
            ta = time.perf_counter()
            dist.all_reduce(tj)
            barrier_sync(plan)
            jtj_info["allreduce_ms"] = 1e3 * (time.perf_counter() - ta)
            jtj_info["allreduce_MB"] = nP * nP * 8 / 1e6
        for d in (d_jtj, d_jtf, d_ls, d_w, d_c, d_N):
            plan.device_free(d)

    gather_ms = None
    if dist is not None and args.gather and args.scaling == "strong":
        # the reference's `gather_local_array` equivalent: row blocks of the probabilities travel to every rank
        # (RCCL all-gather over xGMI under nccl); the Jacobian stays distributed, as bulk_fill_dprobs leaves it
        import torch
        from pygsti_amd import dist as gdist
        dev = "cuda" if backend == "nccl" else "cpu"
        loc = torch.zeros(nE_total, dtype=torch.float64, device=dev)
        host = np.empty(nE_local)
        plan.memcpy_d2h(host, d_probs)
        loc[atom.element_slice.start:atom.element_slice.stop] = torch.from_numpy(host).to(dev)
        barrier_sync(plan)
        tg = time.perf_counter()
        full = gdist.gather_elements(loc, layout)
        barrier_sync(plan)
        gather_ms = 1e3 * (time.perf_counter() - tg)
        s = float(full.sum().item())
        assert abs(s - len(circuits)) < 1e-6 * len(circuits), "gathered probabilities must sum to 1 per circuit"

    def measured_traffic(kernel_prefix):
        """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes of THIS workload
        (profiles/r01_hbm_counters.json: FETCH_SIZE and WRITE_SIZE collected in separate passes; KB -> bytes, and
        FETCH_SIZE doubled, the gfx950 correction of MI355X_MICROARCH.md).  None when no profile matches."""
        if world != 1 or lay_world != 1 or args.design != "full" or args.max_len != 1024:
            return None
        try:
            with open(os.path.join(ROOT, "profiles", "r01_hbm_counters.json")) as f:
                prof = json.load(f)[args.deriv]
            for k, v in prof.items():
                if kernel_prefix in k:
                    return (2.0 * v["FETCH_SIZE_KB_per_launch"] + v["WRITE_SIZE_KB_per_launch"]) * 1024.0
        except Exception:
            pass
        return None

    st = plan.stats()
    if rank == 0:
        D = model.dim
        k_ms = float(np.mean(kernel_ms))
        # algorithmic flops of one FD Jacobian on the schedule actually executed by this rank:
        # (nP perturbed passes) x (2 D^2 per gate application + 2 D per element)
        flops = nP * (2.0 * D * D * st["applies_per_pass"] + 2.0 * D * nE_local)
        jac_bytes = 8.0 * nE_local * nP
        if args.deriv == "analytic":
            roof = {"bound": "hbm", "kernel": "analytic_mfma_kernel (+ the backward chain pass that feeds it)",
                    "achieved": jac_bytes / (k_ms * 1e-3) / 1e9,
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": jac_bytes / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "note": "algorithmic bytes = the Jacobian write 8*nE*nP; the kernel gathers forward and backward state vectors (20 GB of L2 traffic, of which the PMC `traffic` figure is what reaches HBM)",
                    "kernel_ms": k_ms, "bytes_per_launch": jac_bytes, "traffic": measured_traffic("analytic_mfma_kernel")}
        else:
            roof = None
        out = {
            "metric": "dprobs Jacobian-elements/sec, 2Q GST L<=1024 (bulk_fill_dprobs, %s)" % (
                "FD eps=1e-7, bit-identical to the reference Map path" if args.deriv == "fd" else
                "analytic derivatives, MatrixForwardSimulator semantics, <=1e-8"),
            "value": nE_total * nP * args.steps / dt,
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
                       "derivative": ("forward finite differences, eps=1e-7 (reference MapForwardSimulator semantics)"
                                      if args.deriv == "fd" else "analytic (reference MatrixForwardSimulator semantics)"),
                       "parallelism": "atoms%d" % world if lay_world == world else
                                      "rank 0 of atoms%d emulated on one GPU" % lay_world},
            "gather_probs_ms": gather_ms,
            "normal_equations": jtj_info,
            "analytic_dprobs": ana_info,
            "probs_per_s": nE_total * n_pr / dtp,
            "probs_ms": 1e3 * dtp / n_pr,
            "roofline": roof or {"bound": "mfma", "compute_unit": "valu_f64",
                         "bound_note": "the compute roof (\"mfma\" in this line's vocabulary): dense fp64 peak 78.6 TFLOP/s, which on MI355X is both the MFMA and the vector-FMA rate. The kernel runs on the fp64 VECTOR ALU -- no matrix instruction can reproduce the reference's un-fused, ordered sums -- so `frac` is against the roof the contract names and `compute_unit` says which pipe does the work",
                         "kernel": "walk_kernel<16,1>", "achieved": flops / (k_ms * 1e-3) / 1e12,
                         "peak": F64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": flops / (k_ms * 1e-3) / 1e12 / F64_VALU_PEAK_TFLOPS,
                         "note": "flops = the reference schedule's nP*(2*D^2*A + 2*D*nE); the kernel executes ~0.56 of them (the rest is provably bit-identical to the base pass) as separate v_mul_f64+v_add_f64 (no FMA: bitwise parity), whose own ceiling is 0.5 of the FMA peak; VALU issue utilisation 93% (profiles/r01_bench_pmc_sq_current.json)",
                         "kernel_ms": k_ms, "flops_per_launch": flops,
                         "hbm_write_GBps": jac_bytes / (k_ms * 1e-3) / 1e9, "hbm_peak_GBps": HBM_PEAK_GBS,
                         "traffic": measured_traffic("walk_kernel<16, 1")},
            "plan": {k: st[k] for k in ("n_circuits", "n_elements", "sum_depth", "trie_nodes", "applies_per_pass",
                                        "n_tasks", "prog_words", "max_slots")},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(pack, model, args.max_len)
        print(json.dumps(out))
    plan.device_free(d_out)
    plan.device_free(d_probs)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
