"""HipMapForwardSimulator: the reference's ForwardSimulator surface on top of libgstfwd.

Same method names, argument meaning and array conventions as the reference so that the callers of
the hot path read the same (pygsti/forwardsims/forwardsim.py:584-609 bulk_fill_probs, :628-662
bulk_fill_dprobs, :701-753 bulk_fill_hprobs, :787-878 iter_hprobs_by_rectangle, :415-582 bulk_probs /
bulk_dprobs / bulk_hprobs; distforwardsim.py:92-340 the per-atom loops and parameter blocking;
mapforwardsim.py:166-172 the constructor, :372-391 the three `_bulk_fill_*_atom` seams).

Derivatives are forward finite differences with `derivative_eps` (1e-7) and FD-of-FD with
`hessian_eps` (1e-5), exactly as MapForwardSimulator computes them -- the device evaluates all
perturbed models concurrently and reproduces the reference's numbers bit for bit.

There is no CPU path: every fill goes through the C ABI and raises `GstDeviceError` without a GPU.
"""
import numpy as np

from . import _lib
from .layout import HipCOPALayout


def _slice_len(s, n):
    if s is None:
        return n
    if isinstance(s, slice):
        return len(range(*s.indices(n)))
    return len(s)


def _to_index_array(s, n):
    if s is None:
        return np.arange(n, dtype=np.int64)
    if isinstance(s, slice):
        return np.arange(*s.indices(n), dtype=np.int64)
    return np.asarray(s, dtype=np.int64)


def _slice_up_range(n, num_slices):
    """mpitools.slice_up_range: `num_slices` nearly equal contiguous slices of range(n)."""
    base, rem = divmod(n, num_slices)
    out, start = [], 0
    for i in range(num_slices):
        ln = base + (1 if i < rem else 0)
        out.append(slice(start, start + ln))
        start += ln
    return out


class HipMapForwardSimulator:
    """Drop-in counterpart of MapForwardSimulator for dense `densitymx` models on MI355X."""

    def __init__(self, model=None, max_cache_size=None, num_atoms=None, processor_grid=None, param_blk_sizes=None,
                 derivative_eps=1e-7, hessian_eps=1e-5, devices=None, target_tasks=0, derivative_mode="fd"):
        self._model = None
        self._max_cache_size = max_cache_size   # accepted for signature parity; the device plan needs no state cache
        self._num_atoms = num_atoms
        self._processor_grid = processor_grid
        self._pblk_sizes = param_blk_sizes
        self.derivative_eps = derivative_eps
        self.hessian_eps = hessian_eps
        self.devices = devices
        self.target_tasks = target_tasks
        # "fd": forward finite differences, bit-identical to MapForwardSimulator (the default simulator);
        # "analytic": exact first derivatives, what MatrixForwardSimulator computes (matrixforwardsim.py:1059-1140)
        assert derivative_mode in ("fd", "analytic")
        self.derivative_mode = derivative_mode
        self.concurrent_fills = None     # None: two-phase (enqueue everywhere, then collect) fills when atoms sit on several GPUs
        if model is not None:
            self.model = model

    # -- model attachment (forwardsim.py:134-150) ---------------------------------------------------------
    @property
    def model(self):
        return self._model

    @model.setter
    def model(self, val):
        self._model = val

    def copy(self, keep_model_attached=False):
        s = HipMapForwardSimulator(None, self._max_cache_size, self._num_atoms, self._processor_grid,
                                   self._pblk_sizes, self.derivative_eps, self.hessian_eps, self.devices,
                                   self.target_tasks, self.derivative_mode)
        if keep_model_attached:
            s._model = self._model
        return s

    def _to_nice_serialization(self):
        # (mapforwardsim.py:174-181 plus this class's own options: a simulator restored from a checkpoint must not
        #  silently change its derivative mode)
        return {"module": type(self).__module__, "class": type(self).__name__,
                "max_cache_size": self._max_cache_size, "derivative_epsilon": self.derivative_eps,
                "hessian_epsilon": self.hessian_eps, "hip_derivative_mode": self.derivative_mode,
                "hip_devices": None if self.devices is None else [int(d) for d in self.devices],
                "hip_target_tasks": int(self.target_tasks)}

    @classmethod
    def _from_nice_serialization(cls, state):
        return cls(None, state.get("max_cache_size"), derivative_eps=state.get("derivative_epsilon", 1e-7),
                   hessian_eps=state.get("hessian_epsilon", 1e-5), devices=state.get("hip_devices"),
                   target_tasks=state.get("hip_target_tasks", 0), derivative_mode=state.get("hip_derivative_mode", "fd"))

    def __getstate__(self):
        st = self.__dict__.copy()
        st["_model"] = None          # the parent model re-attaches itself (forwardsim.py:121-132)
        return st

    # -- layout ---------------------------------------------------------------------------------------------
    def create_layout(self, circuits, dataset=None, resource_alloc=None, array_types=("E",),
                      derivative_dimensions=None, verbosity=0, layout_creation_circuit_cache=None):
        """Build the element index and one device plan per atom (mapforwardsim.py:206-335).  With `dataset` None all
        outcomes are laid out (copalayout.py:161-164); otherwise only the outcomes the data set holds for each circuit,
        in its order (maplayout.py:69): `dataset[circuit]` must give an object with `.outcomes` or an iterable of outcome
        labels."""
        rank = getattr(resource_alloc, "comm_rank", 0) if resource_alloc is not None else 0
        size = getattr(resource_alloc, "comm_size", 1) if resource_alloc is not None else 1
        natoms_default = self._processor_grid[0] if self._processor_grid else max(size, len(self.devices or [0]))
        natoms = self._num_atoms if self._num_atoms is not None else natoms_default
        blk = tuple(self._pblk_sizes) if self._pblk_sizes else (None, None)
        if len(blk) == 1:
            blk = (blk[0], None)
        grid = self._processor_grid      # (None: the layout picks the reference's automatic grid when atoms < ranks)
        return HipCOPALayout(circuits, self.model, natoms, self.devices, rank, size, self.target_tasks, blk, dataset=dataset,
                             mpi_comm=getattr(resource_alloc, "comm", None), processor_grid=grid,
                             partition_cost="depth" if self.derivative_mode == "analytic" else "fd")

    # -- per-atom seams (mapforwardsim.py:372-391) ---------------------------------------------------------------
    def _prepare_atom(self, layout_atom):
        plan = layout_atom.plan()
        L = layout_atom.layout
        if hasattr(self.model, "lindblad_description"):
            # Lindblad-parameterised model (CPTPLND, GLND, H+S: pygsti_amd.lindblad): the DEVICE builds the dense members
            # from the parameter vector, for the base model and for every finite-difference step (gst_set_lindblad)
            desc = self.model.lindblad_description(L.model_gate_labels, L.effect_labels)
            if getattr(plan, "_lb_desc", None) is not desc:
                plan.set_lindblad(desc)
                plan._lb_desc = desc
            plan.set_lindblad_params(self.model.to_vector())
            return plan
        if getattr(plan, "_lb_desc", None) is not None:
            plan.set_lindblad(None)
            plan._lb_desc = None
        plan.set_model(*L.model_arrays(self.model))
        if plan.n_params != self.model.num_params:
            plan.set_param_map(*L.param_map(self.model))
        return plan

    def _bulk_fill_probs_atom(self, array_to_fill, layout_atom, resource_alloc=None):
        plan = self._prepare_atom(layout_atom)
        if array_to_fill.flags.c_contiguous:
            plan.fill_probs(array_to_fill)
        else:
            array_to_fill[:] = plan.fill_probs()

    def _bulk_fill_dprobs_atom(self, array_to_fill, dest_param_slice, layout_atom, param_slice, resource_alloc=None,
                               pr_array_to_fill=None):
        plan = self._prepare_atom(layout_atom)
        nP = self.model.num_params
        pidx = _to_index_array(param_slice, nP)
        didx = None if dest_param_slice is None else _to_index_array(dest_param_slice, array_to_fill.shape[1])
        mode = _lib.DERIV_ANALYTIC if self.derivative_mode == "analytic" else _lib.DERIV_FD
        if mode == _lib.DERIV_FD and getattr(plan, "_lb_desc", None) is not None and plan.stats()["max_depth"] > _lib.LINDBLAD_FD_MAX_DEPTH:
            # Finite differences of a Lindblad-parameterised model at depth: the library offers the device-built route only
            # where it meets the 1e-8 bar (GST_LINDBLAD_FD_MAX_DEPTH).  Beyond, base AND stepped members come from the HOST
            # (this model's own exponential -- the reference's scipy expm), as the reference steps them, and the device walks
            # those dense sets (gst_fill_dprobs_models): both sides of every quotient carry the same member rounding.
            desc, th = plan._lb_desc, self.model.to_vector()
            plan.set_lindblad(None); plan._lb_desc = None
            plan.set_model(*desc.dense(th))
            Gs, Rs, Es = desc.model_sets(th, pidx, self.derivative_eps)
            plan.fill_dprobs_models(Gs, Rs, Es, array_to_fill, didx, self.derivative_eps, pr_array_to_fill)
            return
        plan.fill_dprobs(array_to_fill, pidx, didx, self.derivative_eps, pr_array_to_fill, mode)

    def _bulk_fill_hprobs_atom(self, array_to_fill, dest_param_slice1, dest_param_slice2, layout_atom,
                               param_slice1, param_slice2, resource_alloc=None):
        plan = self._prepare_atom(layout_atom)
        nP = self.model.num_params
        i1, i2 = _to_index_array(param_slice1, nP), _to_index_array(param_slice2, nP)
        d1 = None if dest_param_slice1 is None else _to_index_array(dest_param_slice1, array_to_fill.shape[1])
        d2 = None if dest_param_slice2 is None else _to_index_array(dest_param_slice2, array_to_fill.shape[2])
        # "analytic": exact second derivatives (MatrixForwardSimulator's values) (D = 4, 16, 64);
        # "fd": the Map simulator's FD-of-FD, bit for bit
        mode = _lib.DERIV_ANALYTIC if (self.derivative_mode == "analytic") else _lib.DERIV_FD
        if array_to_fill.flags.c_contiguous:
            plan.fill_hprobs(array_to_fill, i1, i2, d1, d2, self.hessian_eps, mode)
        else:
            tmp = np.ascontiguousarray(array_to_fill)
            plan.fill_hprobs(tmp, i1, i2, d1, d2, self.hessian_eps, mode)
            array_to_fill[...] = tmp

    # -- bulk fills (forwardsim.py:584-753, distforwardsim.py:92-234) ------------------------------------------------
    def _concurrent_devices(self, layout):
        """Several atoms on several GPUs in ONE process (SURVEY 8(e)'s single-process shortcut, `devices=[0..7]`): the
        fills are enqueued on every atom's own stream first and collected afterwards, so the GPUs work side by side."""
        if self.concurrent_fills is not None:
            return bool(self.concurrent_fills) and len(layout.atoms) > 1
        return len(layout.atoms) > 1 and len({at.device for at in layout.atoms}) > 1

    def bulk_fill_probs(self, array_to_fill, layout):
        if self._concurrent_devices(layout):
            jobs = []
            for atom in layout.atoms:
                plan = self._prepare_atom(atom)
                d = plan.workspace("bfp", atom.num_elements * 8)
                plan.fill_probs_dev(d)                               # asynchronous, on the atom's device
                jobs.append((atom, plan, d))
            # drain: every device's copy is enqueued behind its own fill, then all are awaited -- the copies of different
            # GPUs run side by side when the destination is page-locked (allocate_local_array pins large arrays)
            later = []
            for atom, plan, d in jobs:
                out = array_to_fill[atom.element_slice]
                if out.flags.c_contiguous:
                    plan.memcpy_d2h_async(out, d)
                else:
                    later.append((out, plan.memcpy_d2h_async(np.empty(atom.num_elements), d)))
            for atom, plan, d in jobs:
                plan.sync()
            for out, tmp in later:
                out[...] = tmp
            return
        for atom in layout.atoms:
            self._bulk_fill_probs_atom(array_to_fill[atom.element_slice], atom)

    def bulk_fill_dprobs(self, array_to_fill, layout, pr_array_to_fill=None):
        blk = layout.param_dimension_blk_sizes[0]
        gps = layout.global_param_slice
        Np = _slice_len(gps, self.model.num_params)
        if blk is None and self._concurrent_devices(layout) and array_to_fill.flags.c_contiguous and Np > 0 \
                and array_to_fill.shape[1] == Np:
            pidx = _to_index_array(gps, self.model.num_params)
            mode = _lib.DERIV_ANALYTIC if self.derivative_mode == "analytic" else _lib.DERIV_FD
            jobs = []
            for atom in layout.atoms:
                plan = self._prepare_atom(atom)
                d_J = plan.workspace("bfJ", atom.num_elements * Np * 8)
                d_p = plan.workspace("bfp", atom.num_elements * 8)
                plan.fill_dprobs_dev(d_J, Np, pidx, None, self.derivative_eps, d_p, mode)      # asynchronous
                jobs.append((atom, plan, d_J, d_p))
            later = []
            for atom, plan, d_J, d_p in jobs:
                plan.memcpy_d2h_async(array_to_fill[atom.element_slice, :], d_J)
                if pr_array_to_fill is not None:
                    pr = pr_array_to_fill[atom.element_slice]
                    if pr.flags.c_contiguous:
                        plan.memcpy_d2h_async(pr, d_p)
                    else:
                        later.append((pr, plan.memcpy_d2h_async(np.empty(atom.num_elements), d_p)))
            for atom, plan, d_J, d_p in jobs:
                plan.sync()
            for pr, tmp in later:
                pr[...] = tmp
            return
        # columns of the local array this rank fills (distforwardsim.py:399-433 `host_param_slice`): all of them, or its
        # parameter-processor's slice of a full-width array
        hps = layout.host_param_slice
        whole = (hps.start == 0 and array_to_fill.shape[1] == Np)
        for atom in layout.atoms:
            es = atom.element_slice
            pr = None if pr_array_to_fill is None else pr_array_to_fill[es]
            if blk is None:
                self._bulk_fill_dprobs_atom(array_to_fill[es, :], None if whole else hps, atom, gps, None, pr)
            else:
                nblk = int(np.ceil(Np / blk))
                for k, block in enumerate(_slice_up_range(Np, nblk)):
                    shifted = slice(block.start + gps.start, block.stop + gps.start)
                    dest = slice(block.start + hps.start, block.stop + hps.start)
                    self._bulk_fill_dprobs_atom(array_to_fill[es, :], dest, atom, shifted, None, pr if k == 0 else None)
                if Np == 0 and pr is not None:
                    self._bulk_fill_probs_atom(pr, atom)

    def bulk_fill_hprobs(self, array_to_fill, layout, pr_array_to_fill=None, deriv1_array_to_fill=None,
                         deriv2_array_to_fill=None):
        b1, b2 = layout.param_dimension_blk_sizes[0], layout.param_dimension_blk_sizes[1]
        gps1, gps2 = layout.global_param_slice, layout.global_param2_slice
        nP = self.model.num_params
        Np1, Np2 = _slice_len(gps1, nP), _slice_len(gps2, nP)
        # the rank's own slices of full-width arrays (host_param_slice / host_param2_slice); None = the whole width
        h1, h2 = layout.host_param_slice, layout.host_param2_slice
        d1 = None if (h1.start == 0 and array_to_fill.shape[1] == Np1) else h1
        d2 = None if (h2.start == 0 and array_to_fill.shape[2] == Np2) else h2
        for atom in layout.atoms:
            es = atom.element_slice
            if pr_array_to_fill is not None:
                self._bulk_fill_probs_atom(pr_array_to_fill[es], atom)
            if deriv1_array_to_fill is not None:
                self._bulk_fill_dprobs_atom(deriv1_array_to_fill[es, :], d1, atom, gps1)
            if deriv2_array_to_fill is not None:
                if deriv1_array_to_fill is not None and gps1 == gps2:
                    deriv2_array_to_fill[es, h2] = deriv1_array_to_fill[es, h1]
                else:
                    self._bulk_fill_dprobs_atom(deriv2_array_to_fill[es, :], d2, atom, gps2)
            if b1 is None and b2 is None:
                self._bulk_fill_hprobs_atom(array_to_fill[es, :, :], d1, d2, atom, gps1, gps2)
            else:
                assert b1 is not None and b2 is not None, "Both (or neither) of the Hessian block sizes must be specified!"
                for blk1 in _slice_up_range(Np1, int(np.ceil(Np1 / b1))):
                    g1 = slice(blk1.start + gps1.start, blk1.stop + gps1.start)
                    t1 = slice(blk1.start + h1.start, blk1.stop + h1.start)
                    for blk2 in _slice_up_range(Np2, int(np.ceil(Np2 / b2))):
                        g2 = slice(blk2.start + gps2.start, blk2.stop + gps2.start)
                        t2 = slice(blk2.start + h2.start, blk2.stop + h2.start)
                        self._bulk_fill_hprobs_atom(array_to_fill[es, :, :], t1, t2, atom, g1, g2)

    def iter_hprobs_by_rectangle(self, layout, wrt_slices_list, return_dprobs_12=False):
        """Yield (wrtSlice1, wrtSlice2, hprobs[, dprobs12]) per rectangle (forwardsim.py:787-878,
        distforwardsim.py:263-302); the full Hessian is never materialised."""
        nE, nP = layout.num_elements, self.model.num_params
        for s1, s2 in wrt_slices_list:
            n1, n2 = _slice_len(s1, nP), _slice_len(s2, nP)
            h = np.zeros((nE, n1, n2))
            d1 = np.zeros((nE, n1)) if return_dprobs_12 else None
            d2 = np.zeros((nE, n2)) if return_dprobs_12 else None
            for atom in layout.atoms:
                es = atom.element_slice
                if return_dprobs_12:
                    self._bulk_fill_dprobs_atom(d1[es, :], None, atom, s1)
                    if s1 == s2:
                        d2[es, :] = d1[es, :]
                    else:
                        self._bulk_fill_dprobs_atom(d2[es, :], None, atom, s2)
                self._bulk_fill_hprobs_atom(h[es, :, :], None, None, atom, s1, s2)
            if return_dprobs_12:
                yield s1, s2, h, d1[:, :, None] * d2[:, None, :]
            else:
                yield s1, s2, h

    def _iter_atom_hprobs_by_rectangle(self, atom, wrt_slices_list, return_dprobs_12, resource_alloc=None):
        nE, nP = atom.num_elements, self.model.num_params
        for s1, s2 in wrt_slices_list:
            n1, n2 = _slice_len(s1, nP), _slice_len(s2, nP)
            h = np.zeros((nE, n1, n2))
            if return_dprobs_12:
                d1 = np.zeros((nE, n1)); d2 = np.zeros((nE, n2))
                self._bulk_fill_dprobs_atom(d1, None, atom, s1)
                if s1 == s2: d2[:, :] = d1
                else: self._bulk_fill_dprobs_atom(d2, None, atom, s2)
            self._bulk_fill_hprobs_atom(h, None, None, atom, s1, s2)
            if return_dprobs_12:
                yield s1, s2, h, d1[:, :, None] * d2[:, None, :]
            else:
                yield s1, s2, h

    # -- normal equations on the device ("next" row f1: layout.fill_jtj / fill_jtf, distlayout.py:1220-1359) ----------
    def bulk_fill_jtj_jtf(self, jtj, jtf, layout, row_scale=None, f=None, pr_array_to_fill=None):
        """J^T J and J^T f of the (optionally row-scaled) Jacobian without moving the Jacobian off the GPU:
        the Jacobian of each atom is filled in HBM, scaled by `row_scale[k]` (the objective's dlsvec factor),
        contracted there, and only num_params^2 + num_params numbers come back.  jtj: (nP, nP), jtf: (nP,) host
        arrays (summed over this process's atoms; ranks all-reduce them as fill_jtj does)."""
        nP = self.model.num_params
        jtj[...] = 0.0
        if jtf is not None:
            jtf[...] = 0.0
        mode = _lib.DERIV_ANALYTIC if self.derivative_mode == "analytic" else _lib.DERIV_FD
        if layout.processor_grid[1] * layout.processor_grid[2] > 1:
            if getattr(layout, "device_comm", None) is not None:
                return self._bulk_fill_jtj_jtf_grid(jtj, jtf, layout, row_scale, f, pr_array_to_fill, mode)
            # no device communicator attached (dist.init(...).comm -> layout.device_comm): the ranks of an atom-processor
            # hold the SAME atoms, so exactly one of them -- parameter-processor (0, 0) -- contributes the full-width
            # products and the others contribute zeros; the caller's all-reduce over ranks is then right
            if not self._first_param_proc(layout):
                if pr_array_to_fill is not None:
                    for atom in layout.atoms:
                        self._bulk_fill_probs_atom(pr_array_to_fill[atom.element_slice], atom)
                return
        pidx = np.arange(nP, dtype=np.int64)
        for atom in layout.atoms:
            plan = self._prepare_atom(atom)
            nE = atom.num_elements
            es = atom.element_slice
            d_J = plan.device_malloc(nE * nP * 8); d_pr = plan.device_malloc(nE * 8)
            d_jtj = plan.device_malloc(nP * nP * 8); d_jtf = plan.device_malloc(nP * 8)
            d_w = d_f = None
            try:
                plan.fill_dprobs_dev(d_J, nP, pidx, None, self.derivative_eps, d_pr, mode)
                if row_scale is not None:
                    d_w = plan.device_malloc(nE * 8); plan.memcpy_h2d(d_w, np.asarray(row_scale, np.float64)[es])
                plan.fill_jtj_dev(d_J, nE, nP, nP, d_jtj, d_w)
                part = np.empty((nP, nP)); plan.memcpy_d2h(part, d_jtj); jtj += part
                if jtf is not None:
                    d_f = plan.device_malloc(nE * 8); plan.memcpy_h2d(d_f, np.asarray(f, np.float64)[es])
                    plan.fill_jtf_dev(d_J, nE, nP, nP, d_f, d_jtf)
                    pv = np.empty(nP); plan.memcpy_d2h(pv, d_jtf); jtf += pv
                if pr_array_to_fill is not None:
                    plan.memcpy_d2h(pr_array_to_fill[es], d_pr)
            finally:
                for d in (d_J, d_pr, d_jtj, d_jtf, d_w, d_f):
                    if d is not None:
                        plan.device_free(d)

    @staticmethod
    def _first_param_proc(layout):
        """True on the one rank of each atom-processor that contributes whole-atom reductions (sums over elements of
        full-width quantities): under a processor grid with parameter-processors every rank of an atom-processor holds the
        same atoms, and a sum over ranks must count each atom once."""
        return getattr(layout, "param_proc_index", 0) == 0 and getattr(layout, "param2_proc_index", 0) == 0

    def _bulk_fill_jtj_jtf_grid(self, jtj, jtf, layout, row_scale, f, pr_array_to_fill, mode):
        """bulk_fill_jtj_jtf under a processor grid with parameter-processors: a rank fills only ITS column slice of its
        atoms' Jacobian rows, but J^T J needs whole rows.  The ranks of an atom-processor therefore split each atom's ROWS
        among themselves and exchange, between device buffers, the column blocks of each other's row shares
        (gst_comm_exchange_blocks: every pair of ranks its own transfer; the reference broadcasts transposed column
        slices between host arrays, distlayout.py:1306-1346); each then contracts its [rows/G x nP] share on its GPU.  The
        returned jtj / jtf are this rank's partial sums, as without a grid.  Needs `layout.device_comm` (a _lib.Comm)."""
        comm = getattr(layout, "device_comm", None)
        if comm is None:
            raise RuntimeError("a processor grid with parameter-processors needs the device communicator: "
                               "set layout.device_comm = pygsti_amd.dist.init(...).comm")
        nP = self.model.num_params
        na, np1, np2 = layout.processor_grid
        G = np1 * np2
        from .layout import _slice_up_range as _sur
        cs = layout.param_slices
        my_cols = cs[layout.param_proc_index]
        c_mine = my_cols.stop - my_cols.start
        pidx = np.arange(my_cols.start, my_cols.stop, dtype=np.int64)
        qme = layout.param_proc_index * np2 + layout.param2_proc_index
        atoms_of = [layout.atoms_of_processor(g) for g in range(na)]
        for k in range(max(len(x) for x in atoms_of)):
            mine = atoms_of[layout.atom_proc_index][k] if k < len(atoms_of[layout.atom_proc_index]) else None
            blocks = layout.column_exchange_blocks(k)       # the same list on every rank
            plan = self._prepare_atom(mine if mine is not None else layout.atoms[0])
            bufs = []
            try:
                d_J = d_stage = d_T = None
                if mine is not None:
                    nE_a = mine.num_elements
                    es = mine.element_slice
                    share = _sur(nE_a, G)[qme]
                    n_my = share.stop - share.start
                    d_J = plan.device_malloc(max(nE_a * c_mine, 1) * 8); bufs.append(d_J)
                    d_pr = plan.device_malloc(max(nE_a, 1) * 8); bufs.append(d_pr)
                    d_stage = plan.device_malloc(max(n_my * nP, 1) * 8); bufs.append(d_stage)
                    d_T = plan.device_malloc(max(n_my * nP, 1) * 8); bufs.append(d_T)
                    plan.fill_dprobs_dev(d_J, c_mine, pidx, None, self.derivative_eps, d_pr, mode)
                comm.exchange_blocks(d_J, d_stage, blocks, plan)          # stream-ordered behind the fill (RCCL); collective
                if mine is None:
                    continue
                for ip1 in range(np1):                                    # block-column-major staging -> row-major rows
                    c = cs[ip1].stop - cs[ip1].start
                    plan.copy_block_dev(d_T + cs[ip1].start * 8, nP, d_stage + n_my * cs[ip1].start * 8, c, n_my, c)
                rows = slice(es.start + share.start, es.start + share.stop)
                d_jtj = plan.device_malloc(nP * nP * 8); bufs.append(d_jtj)
                d_w = None
                if row_scale is not None and n_my:
                    d_w = plan.device_malloc(n_my * 8); bufs.append(d_w)
                    plan.memcpy_h2d(d_w, np.asarray(row_scale, np.float64)[rows])
                if n_my:
                    plan.fill_jtj_dev(d_T, n_my, nP, nP, d_jtj, d_w)
                    part = np.empty((nP, nP)); plan.memcpy_d2h(part, d_jtj); jtj += part
                    if jtf is not None:
                        d_f = plan.device_malloc(n_my * 8); bufs.append(d_f)
                        d_jtf = plan.device_malloc(nP * 8); bufs.append(d_jtf)
                        plan.memcpy_h2d(d_f, np.asarray(f, np.float64)[rows])
                        plan.fill_jtf_dev(d_T, n_my, nP, nP, d_f, d_jtf)
                        pv = np.empty(nP); plan.memcpy_d2h(pv, d_jtf); jtf += pv
                if pr_array_to_fill is not None:
                    plan.memcpy_d2h(pr_array_to_fill[es], d_pr)
            finally:
                plan.sync()
                for d in bufs:
                    plan.device_free(d)

    def bulk_fill_lsq_step(self, jtj, jtf, layout, counts, total_counts, objective="logl", min_prob_clip=1e-4,
                           radius=1e-4, prob_clip_interval=None, lsvec_to_fill=None, pr_array_to_fill=None):
        """Everything a Levenberg-Marquardt iteration needs from the data, without leaving the GPU: probabilities and
        Jacobian (bulk_fill_dprobs), the objective's element-wise maps (lsvec and the dlsvec row scale of
        TimeIndependentMDCObjectiveFunction, pygsti/objectivefns/objectivefns.py:4573-4665, for `objective` = 'chi2'
        or 'logl'), then J_s^T J_s and J_s^T lsvec (optimize/simplerlm.py:677-678).  `counts` / `total_counts` are
        per-element host arrays in layout order.  Returns the objective value sum(terms) of this process's atoms;
        jtj (nP, nP) and jtf (nP,) are summed over them (ranks all-reduce, as layout.fill_jtj does).  Under a processor
        grid with parameter-processors the ranks of an atom-processor hold the same atoms: parameter-processor (0, 0)
        contributes them, its peers contribute zeros (and fill only `pr_array_to_fill` / `lsvec_to_fill` rows), so the
        all-reduce counts every atom once."""
        nP = self.model.num_params
        jtj[...] = 0.0
        jtf[...] = 0.0
        contributes = self._first_param_proc(layout)
        mode = _lib.DERIV_ANALYTIC if self.derivative_mode == "analytic" else _lib.DERIV_FD
        counts = np.asarray(counts, np.float64)
        total_counts = np.asarray(total_counts, np.float64)
        total = 0.0
        for atom in layout.atoms:
            plan = self._prepare_atom(atom)
            nE = atom.num_elements
            es = atom.element_slice
            if not contributes:                 # (kept for the rows the caller asked for; no share in the sums)
                d_pr, d_c, d_N, d_ls, d_w = [plan.workspace("lsq%d" % k, nE * 8) for k in (1, 2, 3, 4, 5)]
                plan.memcpy_h2d(d_c, counts[es]); plan.memcpy_h2d(d_N, total_counts[es])
                plan.fill_probs_dev(d_pr)
                plan.objective_rows_dev(objective, d_pr, d_c, d_N, nE, d_ls, d_w, None, min_prob_clip, radius, prob_clip_interval)
                if lsvec_to_fill is not None:
                    plan.memcpy_d2h(lsvec_to_fill[es], d_ls)
                if pr_array_to_fill is not None:
                    plan.memcpy_d2h(pr_array_to_fill[es], d_pr)
                continue
            t, part, pv = plan.lsq_step(nP, counts[es], total_counts[es], objective, self.derivative_eps, mode, min_prob_clip,
                                        radius, prob_clip_interval,
                                        None if lsvec_to_fill is None else lsvec_to_fill[es],
                                        None if pr_array_to_fill is None else pr_array_to_fill[es])
            total += t; jtj += part; jtf += pv
        return total

    def bulk_fill_objective_hessian(self, hessian, layout, counts, total_counts, objective="logl", min_prob_clip=1e-4,
                                    radius=1e-4, prob_clip_interval=None, row_block=16, max_block_bytes=64e9):
        """The objective's Hessian (num_params x num_params) the way `_construct_hessian` assembles it from rectangles
        (pygsti/objectivefns/objectivefns.py:1640-1690, 4914-4968): for every atom and every block of `row_block` rows,
        FD-of-FD hprobs and FD dprobs are produced and contracted with the objective's dterms / hterms ON THE DEVICE
        (gst_objective_hessian_block); only the block's row_block x num_params numbers come back.  `hessian` is summed
        over this process's atoms (ranks all-reduce it, as `_gather_hessian` does).  Map-path semantics: both
        derivative levels are finite differences with `hessian_eps` -- or, with derivative_mode="analytic", exact
        derivatives at both levels (Matrix-path semantics).  Under a processor grid with parameter-processors only
        parameter-processor (0, 0) of each atom-processor contributes (its peers hold the same atoms)."""
        nP = self.model.num_params
        hessian[...] = 0.0
        if not self._first_param_proc(layout):
            return hessian
        counts = np.asarray(counts, np.float64)
        total_counts = np.asarray(total_counts, np.float64)
        cols = np.arange(nP, dtype=np.int64)
        mode = _lib.DERIV_ANALYTIC if (self.derivative_mode == "analytic") else _lib.DERIV_FD
        for atom in layout.atoms:
            plan = self._prepare_atom(atom)
            nE = atom.num_elements
            es = atom.element_slice
            rb = int(max(1, min(row_block, max_block_bytes // max(1, nE * nP * 8))))
            d_c = plan.device_malloc(max(nE * 8, 8)); d_N = plan.device_malloc(max(nE * 8, 8))
            try:
                plan.memcpy_h2d(d_c, counts[es]); plan.memcpy_h2d(d_N, total_counts[es])
                for r0 in range(0, nP, rb):
                    rows = np.arange(r0, min(nP, r0 + rb), dtype=np.int64)
                    hessian[rows, :] += plan.objective_hessian_block(objective, d_c, d_N, rows, cols, self.hessian_eps,
                                                                     min_prob_clip, radius, prob_clip_interval, mode)
            finally:
                plan.device_free(d_c); plan.device_free(d_N)
        return hessian

    # -- convenience (forwardsim.py:171-277, 415-582) -------------------------------------------------------------------
    def bulk_probs(self, circuits, clip_to=None, resource_alloc=None, smartc=None):
        layout = self.create_layout(circuits, array_types=("e",))
        vp = np.empty(layout.num_elements)
        self.bulk_fill_probs(vp, layout)
        if clip_to is not None:
            vp = np.clip(vp, clip_to[0], clip_to[1])
        return {c: {o: vp[k] for k, o in zip(range(*inds.indices(len(vp))), outs)}
                for inds, c, outs in layout.iter_unique_circuits()}

    def probs(self, circuit, outcomes=None, time=None, resource_alloc=None):
        return self.bulk_probs([circuit])[tuple(circuit)]

    def bulk_dprobs(self, circuits, resource_alloc=None, smartc=None):
        layout = self.create_layout(circuits, array_types=("ep",))
        vdp = np.empty((layout.num_elements, self.model.num_params))
        self.bulk_fill_dprobs(vdp, layout)
        return {c: {o: vdp[k] for k, o in zip(range(*inds.indices(len(vdp))), outs)}
                for inds, c, outs in layout.iter_unique_circuits()}

    def dprobs(self, circuit, resource_alloc=None):
        return self.bulk_dprobs([circuit])[tuple(circuit)]

    def bulk_hprobs(self, circuits, resource_alloc=None, smartc=None):
        layout = self.create_layout(circuits, array_types=("epp",))
        nP = self.model.num_params
        vhp = np.empty((layout.num_elements, nP, nP))
        self.bulk_fill_hprobs(vhp, layout)
        return {c: {o: vhp[k] for k, o in zip(range(*inds.indices(len(vhp))), outs)}
                for inds, c, outs in layout.iter_unique_circuits()}

    def hprobs(self, circuit, resource_alloc=None):
        return self.bulk_hprobs([circuit])[tuple(circuit)]
