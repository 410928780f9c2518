"""COPA layout for the device path: element index + per-atom evaluation plans.

Host-side mirror of the surface of pyGSTi's `CircuitOutcomeProbabilityArrayLayout` /
`DistributableCOPALayout` / `MapCOPALayout` that the objective functions and the simulator touch
(layouts/copalayout.py:27-801, distlayout.py:110-1405, maplayout.py:171-328):
`num_elements`, `num_circuits`, `atoms` (each with a contiguous `element_slice`),
`indices_for_index`, `outcomes_for_index`, `indices_and_outcomes_for_index`, `iter_unique_circuits`,
`allocate_local_array` for the 'e' / 'ep' / 'epp' families, `global_layout`, `host_element_slice`,
`global_param_slice`, `global_param2_slice`, `param_dimension_blk_sizes`, `resource_alloc`.

Element order follows the reference's 1-atom Map layout: element k <-> (circuit k // nOutcomes,
outcome k % nOutcomes) with circuits in the caller's order (tests/golden fixtures pin this).  With
`num_atoms` > 1 circuits are dealt to atoms as contiguous runs of the prefix-sorted circuit list
(cut where few gate applications are shared), every atom owning one contiguous element slice --
as in the reference, the element order then differs from the caller's circuit order and
`indices_for_index` is the map.  Atoms are the unit of multi-GPU sharding (distlayout.py:326-332).
"""
import itertools

import numpy as np

from . import _lib
from .model import KIND_GATE, KIND_RHO, KIND_EFFECT


def _slice_up_range(n, num_slices):
    """mpitools.slice_up_range (mpitools.py:240-270): `num_slices` nearly equal contiguous slices of range(n)."""
    base, rem = divmod(n, num_slices)
    out, start = [], 0
    for i in range(num_slices):
        ln = base + (1 if i < rem else 0)
        out.append(slice(start, start + ln))
        start += ln
    return out


def _ragged_take(values, ptr, rows, out_ptr):
    """Concatenation of values[ptr[r]:ptr[r+1]] for r in rows (out_ptr: the running lengths, out_ptr[-1] in all)."""
    total = int(out_ptr[-1])
    if total == 0:
        return values[:0].copy()
    rows = np.asarray(rows, np.int64)
    if len(rows) and rows[0] == 0 and len(rows) == len(ptr) - 1 and np.array_equal(rows, np.arange(len(rows))):
        return np.ascontiguousarray(values[:total])              # every row in order: the array itself
    lens = out_ptr[1:] - out_ptr[:-1]
    src = np.repeat(np.asarray(ptr, np.int64)[rows] - out_ptr[:-1], lens) + np.arange(total, dtype=np.int64)
    return values[src]


class _ResourceAlloc:
    """Stand-in for pygsti.baseobjs.ResourceAllocation on the serial/one-process-per-GPU path."""

    def __init__(self, rank=0, size=1):
        self.comm = None
        self.comm_rank, self.comm_size = rank, size
        self.is_host_leader = True
        self.mem_limit = None

    def host_comm_barrier(self):
        pass

    def check_can_allocate_memory(self, nbytes):
        pass


class HipLayoutAtom:
    """One atom: a set of circuits, its contiguous slice of the element dimension, one device plan."""

    def __init__(self, layout, circuit_indices, element_slice, device):
        self.layout = layout
        self.circuit_indices = np.asarray(circuit_indices, dtype=np.int64)
        self.element_slice = element_slice
        self.num_elements = element_slice.stop - element_slice.start
        self.device = device
        self._plan = None

    @property
    def cache_size(self):
        return 0   # the device schedule keeps no per-circuit state cache (save slots live in registers)

    def plan(self):
        """Compile (once) and return the device plan of this atom."""
        if self._plan is None:
            L = self.layout
            nO = L.num_outcomes
            n = len(self.circuit_indices)
            lens = L._circ_len[self.circuit_indices]
            ptr = np.zeros(n + 1, np.int64)
            np.cumsum(lens, out=ptr[1:])
            # (one gather instead of a Python loop over the circuits: 1.4 s of a 1.9 s call for the 2Q L<=1024 design)
            gates = _ragged_take(L._circ_gates, L._circ_ptr, self.circuit_indices, ptr).astype(np.int32, copy=False)
            # effect CSR of this atom's circuits: all outcomes, or only those the data set observed (maplayout.py:69)
            cnt = L._out_ptr[self.circuit_indices + 1] - L._out_ptr[self.circuit_indices]
            eff_ptr = np.zeros(n + 1, np.int64)
            np.cumsum(cnt, out=eff_ptr[1:])
            eff_label = _ragged_take(L._out_idx, L._out_ptr, self.circuit_indices, eff_ptr).astype(np.int32, copy=False)
            eff_dest = np.arange(int(eff_ptr[-1]), dtype=np.int32)
            self._plan = _lib.Plan.from_circuits(L.dim, L.num_gates, L.num_preps, nO, int(eff_ptr[-1]),
                                                 L._circ_rho[self.circuit_indices].astype(np.int32), ptr,
                                                 gates, eff_ptr, eff_label, eff_dest, device=self.device,
                                                 target_tasks=L.target_tasks, max_slots=L.max_slots)
        return self._plan


class HipCOPALayout:
    def __init__(self, circuits, model, num_atoms=1, devices=None, rank=0, size=1, target_tasks=0,
                 param_dimension_blk_sizes=(None, None), max_slots=0, dataset=None, mpi_comm=None, processor_grid=None, partition_cost="fd"):
        self.circuits = [tuple(c) for c in circuits]
        self.num_circuits = len(self.circuits)
        self.model_gate_labels = list(model.operations.keys())
        self.num_gates = len(self.model_gate_labels)
        self.dim = model.dim
        self.effect_labels = model.effect_labels
        self._outcomes = [tuple([lbl.split("_", 1)[1]]) for lbl in self.effect_labels]
        self.num_outcomes = len(self._outcomes)          # effect vectors of the plan (all POVMs)
        # Several state preparations / POVMs (maplayout.py:101-134 handles `rho_labels` and per-circuit effect sets): a
        # circuit then names its preparation first and its POVM last, as the reference's completed circuits do
        # (models/model.py:1439-1631); with one of each they stay implicit.
        self.prep_labels = list(model.preps.keys())
        self.povm_labels = list(model.povms.keys())
        self.num_preps = len(self.prep_labels)
        prep_of = {l: i for i, l in enumerate(self.prep_labels)}
        povm_of = {l: i for i, l in enumerate(self.povm_labels)}
        povm_effects = [[k for k, lbl in enumerate(self.effect_labels) if lbl.split("_", 1)[0] == pl] for pl in self.povm_labels]
        self._circ_rho = np.zeros(self.num_circuits, np.int64)
        self._circ_povm = np.zeros(self.num_circuits, np.int64)
        gate_only = []
        for i, c in enumerate(self.circuits):
            if c and c[0] in prep_of and c[0] not in self.model_gate_labels:
                self._circ_rho[i] = prep_of[c[0]]; c = c[1:]
            elif self.num_preps != 1:
                raise ValueError("circuit %d does not name its state preparation and the model has %d" % (i, self.num_preps))
            if c and c[-1] in povm_of and c[-1] not in self.model_gate_labels:
                self._circ_povm[i] = povm_of[c[-1]]; c = c[:-1]
            elif len(self.povm_labels) != 1:
                raise ValueError("circuit %d does not name its POVM and the model has %d" % (i, len(self.povm_labels)))
            gate_only.append(c)
        self._gate_circuits = gate_only                    # the circuits without their SPAM labels
        self.target_tasks = target_tasks
        self.max_slots = max_slots
        self.pin_arrays = True          # allocate_local_array page-locks large element-dimension arrays
        self.last_array_pinned = False
        self._num_params = model.num_params
        lookup = {l: i for i, l in enumerate(self.model_gate_labels)}
        self._circ_len = np.fromiter((len(c) for c in self._gate_circuits), dtype=np.int64, count=self.num_circuits)
        self._circ_ptr = np.zeros(self.num_circuits + 1, np.int64)
        np.cumsum(self._circ_len, out=self._circ_ptr[1:])
        # (C-level iteration: 32 M labels for the 2Q L<=1024 design)
        self._circ_gates = np.fromiter(map(lookup.__getitem__, itertools.chain.from_iterable(self._gate_circuits)),
                                       dtype=np.int32, count=int(self._circ_ptr[-1]))
        self._rank, self._size = rank, size
        self.partition_cost = partition_cost     # what atoms are balanced on: "fd" (finite-difference work), "depth" (gate applications: analytic mode) or "trie" (new states)
        self._mpi_comm = mpi_comm       # optional mpi4py-style communicator of the caller's ResourceAllocation
        # ---- outcomes laid out per circuit (copalayout.py:155-168) ---------------------------------------------------
        # dataset None: every outcome of the POVM; otherwise only the outcomes the data set holds for the circuit, in
        # the data set's order (`dataset[circuit].outcomes`; outcomes the model does not know are dropped, as
        # bulk_expand_instruments_and_separate_povm does, models/model.py:1764-1768).  `dataset` is anything indexable by
        # the circuit (tuple of gate labels) that returns an object with `.outcomes` or an iterable of outcome labels
        # ('01' or ('01',)).
        nO = self.num_outcomes
        if dataset is None and len(self.povm_labels) == 1:
            self._out_ptr = np.arange(self.num_circuits + 1, dtype=np.int64) * nO
            self._out_idx = np.tile(np.arange(nO, dtype=np.int32), self.num_circuits)
        elif dataset is None:                              # every outcome of the circuit's own POVM
            cnt = np.array([len(e) for e in povm_effects], np.int64)[self._circ_povm]
            self._out_ptr = np.zeros(self.num_circuits + 1, np.int64)
            np.cumsum(cnt, out=self._out_ptr[1:])
            self._out_idx = np.concatenate([np.asarray(povm_effects[k], np.int32) for k in self._circ_povm]) if self.num_circuits else np.zeros(0, np.int32)
        else:
            ptr = np.zeros(self.num_circuits + 1, np.int64); idx = []
            for i, c in enumerate(self.circuits):
                lookup_o = {self._outcomes[k]: k for k in povm_effects[self._circ_povm[i]]}
                row = dataset[c]
                # the reference's Map layout takes `unique_outcomes` (maplayout.py:69, mapforwardsim.py:358): a row of
                # time-stamped data lists an outcome once per time stamp in `.outcomes`
                outs = getattr(row, "unique_outcomes", None)
                if outs is None:
                    outs = getattr(row, "outcomes", row)
                seen = set()
                for o in outs:
                    o = (o,) if isinstance(o, str) else tuple(o)
                    if o in lookup_o and o not in seen:
                        seen.add(o)
                        idx.append(lookup_o[o])
                ptr[i + 1] = len(idx)
            self._out_ptr, self._out_idx = ptr, np.asarray(idx, np.int32)
        self._has_dataset = dataset is not None or len(self.povm_labels) != 1      # (outcome sets differ per circuit)

        # ---- processor grid, then deal circuits to atoms --------------------------------------------
        num_atoms = max(1, int(num_atoms or 1))
        if not processor_grid and size > 1 and num_atoms < size and size % num_atoms == 0:
            # fewer atoms than ranks: the ranks left over split the parameter columns (the reference's automatic grid,
            # distforwardsim.py:469-481: na = gcd(nprocs, natoms), the rest of the processors on the first parameter dimension)
            processor_grid = (num_atoms, size // num_atoms)
        na_req = int(processor_grid[0]) if processor_grid else size
        num_atoms = min(max(num_atoms, na_req), max(self.num_circuits, 1))      # natoms = max(na, num_atoms), distforwardsim.py:461
        groups = self._partition(num_atoms)
        self.global_num_elements = int(self._out_ptr[-1])
        self._circuit_offset = np.empty(self.num_circuits, np.int64)   # first element of each circuit
        atoms, off = [], 0
        devices = list(devices) if devices else [-1]
        n_out = self._out_ptr[1:] - self._out_ptr[:-1]
        for a, idx in enumerate(groups):
            cnt = n_out[idx]
            n_el = int(cnt.sum())
            self._circuit_offset[idx] = off + np.concatenate([[0], np.cumsum(cnt)[:-1]]) if len(idx) else off
            atoms.append(HipLayoutAtom(self, idx, slice(off, off + n_el), devices[a % len(devices)]))
            off += n_el
        self.all_atoms = atoms
        # ---- processor grid (distforwardsim.py:445-485 `_compute_processor_distribution`, distlayout.py:424-660) ----------
        # (na, np1[, np2]): the ranks form `na` atom-processors of np1 * np2 ranks each; the ranks of one atom-processor
        # hold the same atoms and split the parameter columns (np1 slices; np2 slices of the second Hessian dimension)
        # among themselves, hierarchically and in rank order as the reference's nested sub-communicators do.  Default:
        # one rank per atom-processor -- atoms are the only axis.
        grid = tuple(int(x) for x in processor_grid) if processor_grid else (size,)
        na = grid[0]; np1 = grid[1] if len(grid) > 1 else 1; np2 = grid[2] if len(grid) > 2 else 1
        if len(grid) > 3 or min(na, np1, np2) < 1 or na * np1 * np2 != size:
            raise ValueError("processor_grid %r must be (na, np1[, np2]) with na * np1 * np2 == %d ranks" % (processor_grid, size))
        if np1 > max(self._num_params, 1) or np2 > max(self._num_params, 1):
            raise ValueError("more parameter-processors than parameters")
        if na > len(atoms):
            raise ValueError("%d atom-processors but only %d atoms (%d circuits)" % (na, len(atoms), self.num_circuits))
        self.processor_grid = (na, np1, np2)
        q = rank % (np1 * np2)
        self.atom_proc_index, self.param_proc_index, self.param2_proc_index = rank // (np1 * np2), q // np2, q % np2
        self.param_slices = _slice_up_range(self._num_params, np1)          # matches the param-processor indices
        self.param2_slices = _slice_up_range(self._num_params, np2)
        self.max_param_slice_length = max(s.stop - s.start for s in self.param_slices)
        # one process per GPU: atom-processor k owns a SEQUENTIAL block of atoms (distribute_indices_base + _assert_sequential,
        # distlayout.py:327-329 / mpitools.py:191-215: the first `natoms mod na` processors get one atom more), so that its rows
        # are one contiguous range [off_k, off_k + nE_k) of the global element dimension
        self._atom_proc = np.empty(len(atoms), np.int64)
        for g, sl in enumerate(_slice_up_range(len(atoms), na)):
            self._atom_proc[sl] = g
        self.atoms = [at for a, at in enumerate(atoms) if self._atom_proc[a] == self.atom_proc_index]
        mine = [a for a in range(len(atoms)) if self._atom_proc[a] == self.atom_proc_index]
        assert mine == list(range(mine[0], mine[0] + len(mine))), "an atom-processor's atoms must be sequential"
        assert all(self.atoms[i].element_slice.stop == self.atoms[i + 1].element_slice.start for i in range(len(self.atoms) - 1))
        self.local_element_slice = slice(self.atoms[0].element_slice.start, self.atoms[-1].element_slice.stop)   # this rank's rows
        self.num_elements = self.global_num_elements   # arrays are allocated full-size; see allocate_local_array
        self.host_element_slice = slice(0, self.global_num_elements)
        self.global_param_slice = self.param_slices[self.param_proc_index]
        self.global_param2_slice = self.param2_slices[self.param2_proc_index]
        # full-size arrays on every rank (the reference's one-host case, distlayout.py:519-521: a host holds all
        # parameters and every rank writes its own slice): positions in the array ARE the parameter indices
        self.host_param_slice = self.global_param_slice
        self.host_param2_slice = self.global_param2_slice
        self.num_params = self.global_param_slice.stop - self.global_param_slice.start
        self.param_dimension_blk_sizes = tuple(param_dimension_blk_sizes)
        self.global_num_params = self._num_params
        self.host_num_elements = self.global_num_elements
        self.max_atom_elements = max(at.num_elements for at in atoms)

    # ---- partition ------------------------------------------------------------------------------------
    def _partition(self, num_atoms):
        if num_atoms == 1:
            return [np.arange(self.num_circuits)]
        # prefix order of (preparation, gate labels ...), labels compared as strings, and each circuit's common prefix with
        # its predecessor: gst_sort_circuits on integer ranks of the labels (0.6 s for the 2Q L<=1024 design; the Python
        # tuple sort and compare loop this replaces took 8.5 s of a 13 s layout)
        rank = np.empty(max(self.num_gates, 1), np.int32)
        rank[np.argsort(np.array([str(l) for l in self.model_gate_labels], dtype=object), kind="stable")] = np.arange(self.num_gates, dtype=np.int32)
        rho_strs = sorted({str(int(r)) for r in range(max(self.num_preps, 1))})
        head = np.array([rho_strs.index(str(int(r))) for r in self._circ_rho], np.int32)
        order, lcp = _lib.sort_circuits(self._circ_ptr, rank[self._circ_gates] if len(self._circ_gates) else np.zeros(0, np.int32), head)
        cost = (self._circ_len[order] + 1 - lcp + 1).astype(np.int64)          # new trie states (+1)
        if self.partition_cost == "fd":
            cost = self._fd_cost(order, lcp)
        elif self.partition_cost == "depth":      # analytic derivatives: one block product per gate application of every circuit
            cost = (self._circ_len[order] + 1).astype(np.int64)
        cum = np.cumsum(cost)
        total = int(cum[-1])
        cuts = [0]
        for a in range(1, num_atoms):
            target = total * a // num_atoms
            k0 = int(np.searchsorted(cum, target))
            win = max(8, min(200, self.num_circuits // (50 * num_atoms)))     # (2 % of an atom: the cheapest restart near the balanced cut)
            lo, hi = max(cuts[-1] + 1, k0 - win), min(self.num_circuits - 1, k0 + win)
            if lo > hi:
                k = min(max(cuts[-1] + 1, k0), self.num_circuits - 1)
            else:
                k = lo + int(np.argmin(lcp[lo:hi + 1]))   # cheapest restart near the balanced cut
            cuts.append(k)
        cuts.append(self.num_circuits)
        order = np.asarray(order, np.int64)
        return [np.sort(order[cuts[a]:cuts[a + 1]]) for a in range(num_atoms) if cuts[a + 1] > cuts[a]]

    def _fd_cost(self, order, lcp):
        """What the new states of each circuit (in prefix-sorted order) cost a FINITE-DIFFERENCE Jacobian: a state is
        re-propagated by the parameter wavefronts of every gate that occurs on its path (the others find it in the base
        pass's cache) and by the preparation's wavefront.  Atoms of equal trie work differ by up to 17 % in this measure --
        germs made of several distinct gates dirty more wavefronts -- and so did the measured steps of the eight ranks of
        the 2Q design (4.11 ... 4.62 ms); the N-GPU step is the slowest rank's.  First occurrence of every gate in every
        circuit (gst_circuit_first_use), then, per gate, the number of new states behind it."""
        n, nG = self.num_circuits, self.num_gates
        ptr, g = self._circ_ptr, self._circ_gates
        ln = ptr[1:] - ptr[:-1]
        first = _lib.circuit_first_use(ptr, g, nG)
        first[first < 0] = np.iinfo(np.int64).max // 4                    # never applied: behind every state
        waves = max(1, (self.dim * self.dim) // 64)                      # wavefronts of 64 parameters per gate
        Lk = ln[order] + 1                                                # path length in states (the preparation first)
        fo = first[order] + 1                                             # state index behind the first occurrence
        cost = (Lk - lcp).astype(np.int64)                                # the preparation's wavefront: every new state
        for gg in range(nG):
            cost += waves * np.maximum(0, Lk - np.maximum(lcp, fo[:, gg]))
        return np.maximum(cost, 1)

    # ---- element index (copalayout.py:683-763) ----------------------------------------------------------------
    def __len__(self):
        return self.num_elements

    @property
    def global_layout(self):
        return self

    def indices_for_index(self, index):
        o = int(self._circuit_offset[index])
        return slice(o, o + int(self._out_ptr[index + 1] - self._out_ptr[index]))

    def outcomes_for_index(self, index):
        if not self._has_dataset:
            return tuple(self._outcomes)
        return tuple(self._outcomes[k] for k in self._out_idx[self._out_ptr[index]:self._out_ptr[index + 1]])

    def indices_and_outcomes_for_index(self, index):
        return self.indices_for_index(index), self.outcomes_for_index(index)

    def indices(self, circuit):
        return self.indices_for_index(self.circuits.index(tuple(circuit)))

    def outcomes(self, circuit):
        return self.outcomes_for_index(self.circuits.index(tuple(circuit)))

    def iter_unique_circuits(self):
        for i, c in enumerate(self.circuits):
            yield self.indices_for_index(i), c, self.outcomes_for_index(i)

    # ---- arrays (copalayout.py:284-361) ----------------------------------------------------------------------------
    PIN_MIN_BYTES = 1 << 18          # (256 KB: the 1Q Jacobian is 1 MB)

    @staticmethod
    def _own_pages_array(shape, dtype):
        """A zero-filled array on pages of its own (anonymous mmap: page-aligned, shares no page with the heap) -- what may be
        handed to gst_host_register.  numpy itself serves arrays below 32 MB from the brk heap, where a registered range
        shares its first and last page with other allocations (the GPU fault on a host-heap address)."""
        import mmap
        dt = np.dtype(dtype)
        count = int(np.prod(shape, dtype=np.int64))
        mm = mmap.mmap(-1, max(count * dt.itemsize, mmap.PAGESIZE))
        return np.frombuffer(mm, dtype=dt, count=count).reshape(shape)

    def allocate_local_array(self, array_type, dtype="d", zero_out=False, memory_tracker=None, extra_elements=0):
        nE, nP = self.num_elements + extra_elements, self._num_params
        shape = {"e": (nE,), "ep": (nE, nP), "ep2": (nE, nP), "epp": (nE, nP, nP), "p": (nP,), "jtj": (nP, nP),
                 "jtf": (nP,), "c": (self.num_circuits,), "cp": (self.num_circuits, nP),
                 "cp2": (self.num_circuits, nP), "cpp": (self.num_circuits, nP, nP)}[array_type]
        # element-dimension arrays are what the fills copy into: page-lock the large ones once, here, so that every
        # later bulk_fill_* runs at PCIe rate (the reference allocates these once per objective and reuses them) -- on
        # pages of their own
        self.last_array_pinned = False
        nbytes = int(np.prod(shape, dtype=np.int64)) * np.dtype(dtype).itemsize
        if array_type in ("e", "ep", "ep2", "epp") and nbytes >= self.PIN_MIN_BYTES and self.pin_arrays:
            arr = self._own_pages_array(shape, dtype)
            self.last_array_pinned = _lib.pin_host_array(arr)
            return arr
        return np.zeros(shape, dtype) if zero_out else np.empty(shape, dtype)

    def free_local_array(self, local_array):
        if isinstance(local_array, np.ndarray):
            _lib.unpin_host_array(local_array)

    # ---- who holds what (processor grid) ---------------------------------------------------------------------------
    def rank_of(self, atom_proc, param_proc=0, param2_proc=0):
        na, np1, np2 = self.processor_grid
        return (atom_proc * np1 + param_proc) * np2 + param2_proc

    def atom_owner_rank(self, atom_index):
        """The first rank of the atom-processor that holds atom `atom_index` (the one that contributes its rows to
        element-only gathers; its peers hold the same rows)."""
        return self.rank_of(int(self._atom_proc[atom_index]))

    def owned_blocks(self, array_type, rank):
        """[(row start, row stop, col slice or None, col2 slice or None)]: the blocks of a full-size local array of
        `array_type` that `rank` is the one to contribute to a gather (every entry of the global array has exactly one
        contributor: rows by atom-processor, columns by parameter-processor; ranks that hold copies contribute nothing)."""
        na, np1, np2 = self.processor_grid
        q = rank % (np1 * np2)
        ia, ip1, ip2 = rank // (np1 * np2), q // np2, q % np2
        dims = {"e": 0, "ep": 1, "ep2": 1, "epp": 2}[array_type]
        if (dims < 2 and ip2 != 0 and array_type != "ep2") or (dims < 1 and ip1 != 0) or (array_type == "ep2" and ip1 != 0):
            return []
        c1 = None if dims == 0 else (self.param2_slices[ip2] if array_type == "ep2" else self.param_slices[ip1])
        c2 = self.param2_slices[ip2] if dims == 2 else None
        return [(at.element_slice.start, at.element_slice.stop, c1, c2) for a, at in enumerate(self.all_atoms) if self._atom_proc[a] == ia]

    def atoms_of_processor(self, atom_proc):
        return [at for a, at in enumerate(self.all_atoms) if self._atom_proc[a] == atom_proc]

    def column_exchange_blocks(self, k):
        """Block list of gst_comm_exchange_blocks for round k of the column-distributed normal equations: every
        atom-processor works on its k-th atom; the rank holding column slice ip1 of that atom's rows ([nE_a x c] row-major)
        sends to each rank q of the atom-processor the rows of q's share, which land block-column-major in q's staging
        array (slice ip1's block at |share| * slice start).  [(src rank, dst rank, src offset, dst offset, count)] in
        doubles, the same on every rank."""
        na, np1, np2 = self.processor_grid
        G = np1 * np2
        blocks = []
        for g in range(na):
            mine = self.atoms_of_processor(g)
            if k >= len(mine):
                continue
            shares = _slice_up_range(mine[k].num_elements, G)
            for ip1, cs in enumerate(self.param_slices):
                c = cs.stop - cs.start
                for q in range(G):
                    n = shares[q].stop - shares[q].start
                    blocks.append((self.rank_of(g, ip1, 0), self.rank_of(g) + q, shares[q].start * c, n * cs.start, n * c))
        return blocks

    def gather_local_array(self, array_type, array_portion, extra_elements=0, all_gather=False, return_shared=False):
        """Assemble the global array from the ranks' portions (distlayout.py:1010-1156): element-dimension arrays
        ('e', 'ep', 'ep2', 'epp') are full-size on every rank with only the rank's own atoms (rows) and parameter slices
        (columns) filled, so the blocks travel to rank 0 (Gatherv) or to every rank (`all_gather`); the other ranks get
        None, as in the reference.  Host arrays go through the control group (pygsti_amd.dist); device arrays have
        `dist.gather_elements_dev`."""
        if self._size == 1 or array_type not in ("e", "ep", "ep2", "epp"):
            return array_portion
        from . import dist as _dist
        nE = self.global_num_elements
        body = _dist.gather_blocks(np.ascontiguousarray(array_portion[:nE]), self, array_type, dst=None if all_gather else 0)
        if body is None:
            return None
        if extra_elements:
            body = np.concatenate([body, array_portion[nE:nE + extra_elements]], axis=0)
        return body

    def allgather_local_array(self, array_type, array_portion, extra_elements=0, return_shared=False):
        """copalayout.py:479-518"""
        return self.gather_local_array(array_type, array_portion, extra_elements, True, return_shared)

    # ---- normal equations (distlayout.py:1220-1359; copalayout.py:549-598) --------------------------------------------
    def _owned_rows_to_device(self, atom, arr, name):
        plan = atom.plan()
        rows = np.ascontiguousarray(arr[atom.element_slice], dtype=np.float64)
        d = plan.workspace(name, max(rows.nbytes, 8))
        plan.memcpy_h2d(d, rows)
        return plan, d

    def _complete_columns(self, j):
        """With parameter-processors (np1 > 1) a rank's 'ep' array holds only its own column slice of its atoms' rows;
        the products below need whole rows.  The reference broadcasts every slice's transpose within the atom-processor
        (fill_jtj, distlayout.py:1306-1346); here the column blocks are exchanged through the control group."""
        if self.processor_grid[1] == 1:
            return j
        from . import dist as _dist
        return _dist.gather_blocks(np.ascontiguousarray(j[:self.global_num_elements]), self, "ep", dst=None, within_atom_proc=True)

    def _row_share(self, atom):
        """The rows of `atom` whose products THIS rank computes: the ranks of an atom-processor hold the same rows (after
        _complete_columns) and split them evenly, so that every GPU does 1/size of the contraction."""
        na, np1, np2 = self.processor_grid
        es = atom.element_slice
        share = _slice_up_range(es.stop - es.start, np1 * np2)[self.param_proc_index * np2 + self.param2_proc_index]
        return slice(es.start + share.start, es.start + share.stop)

    def fill_jtj(self, j, jtj):
        """jtj[:] = j.T @ j for a host 'ep' array: each owned atom's rows are contracted by the split-K MFMA kernel on
        that atom's GPU, the partial products are added and all-reduced over the ranks (fill_jtj, distlayout.py:1259)."""
        nP = j.shape[1]
        acc = np.zeros((nP, nP))
        j = self._complete_columns(j)
        for atom in self.atoms:
            rows = self._row_share(atom)
            if rows.stop <= rows.start:
                continue
            plan = atom.plan()
            blk = np.ascontiguousarray(j[rows], dtype=np.float64)
            d_j = plan.workspace("njJ", max(blk.nbytes, 8)); plan.memcpy_h2d(d_j, blk)
            d_out = plan.workspace("njC", nP * nP * 8)
            plan.fill_jtj_dev(d_j, rows.stop - rows.start, nP, nP, d_out)
            part = np.empty((nP, nP)); plan.memcpy_d2h(part, d_out); acc += part
        if self._size > 1:
            from . import dist as _dist
            _dist.allreduce_sum_host(acc, expect_size=self._size, comm=self._mpi_comm)
        jtj[...] = acc

    def fill_jtf(self, j, f, jtf):
        """jtf[:] = j.T @ f (distlayout.py:1220-1257), same arrangement as fill_jtj."""
        nP = j.shape[1]
        acc = np.zeros(nP)
        j = self._complete_columns(j)
        for atom in self.atoms:
            rows = self._row_share(atom)
            if rows.stop <= rows.start:
                continue
            plan = atom.plan()
            blk = np.ascontiguousarray(j[rows], dtype=np.float64); fv = np.ascontiguousarray(f[rows], dtype=np.float64)
            d_j = plan.workspace("njJ", max(blk.nbytes, 8)); plan.memcpy_h2d(d_j, blk)
            d_f = plan.workspace("njF", max(fv.nbytes, 8)); plan.memcpy_h2d(d_f, fv)
            d_out = plan.workspace("njV", nP * 8)
            plan.fill_jtf_dev(d_j, rows.stop - rows.start, nP, nP, d_f, d_out)
            part = np.empty(nP); plan.memcpy_d2h(part, d_out); acc += part
        if self._size > 1:
            from . import dist as _dist
            _dist.allreduce_sum_host(acc, expect_size=self._size, comm=self._mpi_comm)
        jtf[...] = acc

    def resource_alloc(self, sub_alloc_name=None, empty_if_missing=True):
        return _ResourceAlloc(self._rank, self._size)

    # ---- model arrays in plan order -------------------------------------------------------------------------------
    def model_arrays(self, model):
        gates = np.array([model.operations[l] for l in self.model_gate_labels], dtype=np.float64)
        rhos = np.array([model.preps[l] for l in self.prep_labels], dtype=np.float64)
        effects = np.array([model.effect_vector(l) for l in self.effect_labels], dtype=np.float64)
        return gates, rhos, effects

    def param_map(self, model):
        """(kind, obj, elem) of every model parameter, object indices in plan order."""
        nP = model.num_params
        kind = -np.ones(nP, np.int32); obj = np.zeros(nP, np.int32); elem = np.zeros(nP, np.int32)
        for i, rho_label in enumerate(self.prep_labels):
            s = model.gpindices(KIND_RHO, rho_label)
            kind[s] = KIND_RHO; obj[s] = i; elem[s] = np.arange(s.stop - s.start)
        for i, l in enumerate(self.effect_labels):
            s = model.gpindices(KIND_EFFECT, l)
            kind[s] = KIND_EFFECT; obj[s] = i; elem[s] = np.arange(s.stop - s.start)
        for i, l in enumerate(self.model_gate_labels):
            s = model.gpindices(KIND_GATE, l)
            kind[s] = KIND_GATE; obj[s] = i; elem[s] = np.arange(s.stop - s.start)
        return kind, obj, elem
