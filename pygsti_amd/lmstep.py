"""The binding that takes one Levenberg-Marquardt evaluation of the reference's OWN optimizer to the device
(SURVEY 8(f) row f1 made reachable from `GateSetTomography.run`).

What the reference does per outer iteration (optimize/simplerlm.py:663-678):

    Jac = objective.dlsvec(x)                  # fills the host 'ep' array: bulk_fill_dprobs, then numpy row scalings
    Jnorm = sqrt(ari.norm2_jac(Jac))           # arraysinterface.py:1147-1168   np.linalg.norm(j)**2
    ari.fill_jtj(Jac, JTJ, buf)                # -> layout.fill_jtj   (layouts/distlayout.py:1279-1359)  j.T @ j on the host
    ari.fill_jtf(Jac, f, minus_JTf)            # -> layout.fill_jtf   (layouts/distlayout.py:1220-1263)  j.T @ f

Nothing else ever looks at `Jac`.  So `dlsvec` may return a HANDLE instead of the array:

  * `DeviceJacobian`   -- what the device-side `dlsvec` returns: the scaled Jacobian J_s stays in HBM (one block per layout
                           atom), J_s^T J_s is already formed there; `np.linalg.norm(handle)` answers from its trace
                           (numpy's __array_function__ protocol), `np.asarray(handle)` still materialises the host array
                           for any caller that really wants one (counted: tests assert it never happens in a fit).
  * `LayoutNormalEquations` -- mixed into the layout the simulator returns: `fill_jtj` / `fill_jtf` recognise the handle
                           and copy nP^2 + nP numbers instead of multiplying a (nE, nP) host array; anything else goes to
                           the reference's own implementation.  Also page-locks the reference's 'ep' array -- lazily, on
                           the first fill that really targets it (a fit on the device path never touches it).
  * `DeviceLMStepLogic` -- mixed into the reference's objective classes (`Chi2Function`, `PoissonPicDeltaLogLFunction`;
                           selected through the reference's own `ObjectiveFunctionBuilder(cls_to_build)` hook,
                           objectivefns.py:278-370): `dlsvec` = TimeIndependentMDCObjectiveFunction.dlsvec
                           (objectivefns.py:4633-4665) with the (nE, nP) part on the device and the penalty rows
                           (a handful, :4680-4740) on the host as before.

Everything here is written against duck types so that the GPU box -- where pyGSTi does not exist -- drives the same
code with stand-ins (tests/test_fit_replay.py); the classes that inherit from pyGSTi's are assembled in
pygsti_adapter.py.  No torch, no oracle, no CPU fallback for the arithmetic: when the device step does not apply
(sparse data sets with omitted outcomes, MPI grids, unknown raw objectives) `dlsvec` falls back to the REFERENCE's
host-array algebra over a Jacobian that the device still fills.
"""
import numpy as np

from . import _lib


class DeviceJacobian:
    """The scaled Jacobian of one `dlsvec` call, resident on the device.

    parts: [(plan, d_J, element_slice, d_w)] -- per layout atom the plan whose stream owns the block, the device pointer of
    its row-major [nE_atom][nP] block of dprobs, the rows it covers, and the device vector of the objective's dlsvec row
    factors: the scaled Jacobian J_s = diag(w) J is never stored, the products apply w on the fly
    (gst_fill_normal_eqs_dev).  A part without d_w (3-tuple, or None) holds rows that are already scaled.
    jtj: the (nP, nP) sum over atoms of J_s^T J_s (host copy of the device product).
    penalty_rows: (ex, nP) host array of the objective's extra rows (regularisation / CPTP / SPAM penalties), scaled, or None.

    Valid until the objective's next `dlsvec` (the blocks are plan-lifetime workspaces, re-used every iteration -- the
    reference's `self.jac` is re-used the same way, simplerlm.py:370 "doesn't actually allocate any more mem")."""

    __array_priority__ = 1000.0
    materialisations = 0            # class-wide count of host materialisations (tests: a fit on the device path keeps 0)

    def __init__(self, parts, jtj, n_elements, n_params, penalty_rows=None):
        self.parts = [tuple(p) + (None,) * (4 - len(p)) for p in parts]
        self._jtj = jtj
        self.penalty_rows = penalty_rows
        ex = 0 if penalty_rows is None else penalty_rows.shape[0]
        self.n_elements = int(n_elements)
        self.shape = (int(n_elements) + ex, int(n_params))
        self.ndim = 2
        self.dtype = np.dtype(np.float64)
        self.size = self.shape[0] * self.shape[1]
        self.nbytes = self.size * 8             # (device bytes; simplerlm.py:666 only prints it)

    # -- what the optimizer asks of a Jacobian ---------------------------------------------------------------------------
    def jtj(self):
        out = self._jtj
        if self.penalty_rows is not None and self.penalty_rows.shape[0]:
            out = out + self.penalty_rows.T @ self.penalty_rows
        return out

    def jtf(self, f):
        """J_s^T f for a host vector f of length nE (+ ex): f's element rows go to the device (nE doubles per call; the
        optimizer passes the lsvec it got from `objective.lsvec`, simplerlm.py:678), the penalty rows are a host product."""
        f = np.ascontiguousarray(f, np.float64)
        nP = self.shape[1]
        if f.shape[0] != self.shape[0]:
            raise ValueError("fill_jtf: f has %d rows, the Jacobian %d" % (f.shape[0], self.shape[0]))
        out = np.zeros(nP)
        part = np.empty(nP)
        for plan, d_J, es, d_w in self.parts:
            n = es.stop - es.start
            d_f = plan.workspace("lm_f", n * 8)
            d_o = plan.workspace("lm_jtf", nP * 8)
            plan.memcpy_h2d(d_f, f[es])
            if d_w is None:
                plan.fill_jtf_dev(d_J, n, nP, nP, d_f, d_o)
            else:
                plan.fill_normal_eqs_dev(d_J, n, nP, nP, d_w, d_f=d_f, d_jtf=d_o)
            plan.memcpy_d2h(part, d_o)
            out += part
        if self.penalty_rows is not None and self.penalty_rows.shape[0]:
            out += self.penalty_rows.T @ f[self.n_elements:]
        return out

    def norm2(self):
        """||J_s||_F^2 = trace(J_s^T J_s)"""
        return float(np.trace(self.jtj()))

    # -- numpy interoperability ------------------------------------------------------------------------------------------
    def to_host(self, out=None):
        """The (nE + ex, nP) host array the reference's dlsvec would have returned (7 GB over PCIe for the 2Q design: the
        thing this class exists to avoid; here for callers outside the LM loop)."""
        DeviceJacobian.materialisations += 1
        nP = self.shape[1]
        if out is None:
            out = np.empty(self.shape)
        for plan, d_J, es, d_w in self.parts:
            view = out[es]
            blk = view if view.flags.c_contiguous else np.empty((es.stop - es.start, nP))
            plan.memcpy_d2h(blk, d_J)
            if d_w is not None:                   # the row factors, applied here as the products apply them on the device
                w = np.empty(es.stop - es.start); plan.memcpy_d2h(w, d_w)
                blk *= w[:, None]
            if blk is not view:
                out[es] = blk
        if self.penalty_rows is not None and self.penalty_rows.shape[0]:
            out[self.n_elements:] = self.penalty_rows
        return out

    def __array__(self, dtype=None, copy=None):
        a = self.to_host()
        return a if dtype is None else a.astype(dtype, copy=False)

    def __array_function__(self, func, types, args, kwargs):
        # np.linalg.norm(j) (Frobenius) is what ArraysInterface.norm2_jac asks (optimize/arraysinterface.py:479-492,
        # 1147-1168): answered from the trace of the product the device already holds
        if func is np.linalg.norm and len(args) >= 1 and args[0] is self and all(a is None for a in args[1:]) \
                and all(v is None or (k == "keepdims" and not v) for k, v in kwargs.items()):
            return np.sqrt(max(self.norm2(), 0.0))
        args = tuple(np.asarray(a) if isinstance(a, DeviceJacobian) else a for a in args)
        return func(*args, **kwargs)

    @property
    def T(self):
        return np.asarray(self).T

    def __len__(self):
        return self.shape[0]

    def __getitem__(self, idx):
        return np.asarray(self)[idx]


# ---- layout side ---------------------------------------------------------------------------------------------------------
_PIN_CANDIDATES = {}        # address of an 'ep'-type array allocated by one of our layouts -> nbytes (pinned on first use)


def pin_destination(array_to_fill):
    """Called by the simulator's host-array fills: if the array is (a view of) an 'ep' array one of our layouts allocated,
    page-lock that whole array once (gst_host_register) so that this and every later fill runs at PCIe rate and the
    kernels write straight into it (the reference allocates the array once per objective and re-uses it:
    objectivefns.py:4418-4426).  Lazy, because a fit on the device path never fills it."""
    if not _PIN_CANDIDATES:
        return False
    base = array_to_fill
    while isinstance(getattr(base, "base", None), np.ndarray):
        base = base.base
    addr = base.ctypes.data
    nbytes = _PIN_CANDIDATES.get(addr)
    if nbytes is None or nbytes != base.nbytes:
        return False
    del _PIN_CANDIDATES[addr]
    return _lib.pin_host_array(base)


class LayoutNormalEquations:
    """Mixin for the COPA layout the simulator hands to the objective (class to the left of pyGSTi's MapCOPALayout in the
    MRO; on the GPU box: to the left of a stand-in).  Overrides exactly the three layout methods on the LM path."""

    PIN_MIN_BYTES = 1 << 26          # mmap-backed arrays only (see HipCOPALayout.allocate_local_array)

    def fill_jtj(self, j, jtj, shared_mem_buf=None):
        if isinstance(j, DeviceJacobian):
            sub = getattr(self, "fine_param_subslice", slice(None))
            jtj[:, :] = j.jtj()[sub, :]
            return
        return super().fill_jtj(j, jtj, shared_mem_buf)

    def fill_jtf(self, j, f, jtf):
        if isinstance(j, DeviceJacobian):
            sub = getattr(self, "fine_param_subslice", slice(None))
            jtf[:] = j.jtf(f)[sub]
            return
        return super().fill_jtf(j, f, jtf)

    def allocate_local_array(self, array_type, dtype, zero_out=False, memory_tracker=None, extra_elements=0):
        arr = super().allocate_local_array(array_type, dtype, zero_out, memory_tracker, extra_elements)
        if array_type in ("ep", "ep2", "epp") and isinstance(arr, np.ndarray) and self._is_private(arr):
            base = arr
            while isinstance(getattr(base, "base", None), np.ndarray):
                base = base.base
            if base.flags.c_contiguous and base.nbytes >= self.PIN_MIN_BYTES:
                _PIN_CANDIDATES[base.ctypes.data] = base.nbytes
        return arr

    @staticmethod
    def _is_private(arr):
        """not a shared-memory array (single process: a LocalNumpyArray's handles are all None, sharedmemtools.py)"""
        shm = getattr(arr, "shared_memory_handle", None)
        return shm is None or (isinstance(shm, dict) and all(v is None for v in shm.values()))

    def free_local_array(self, local_array):
        try:
            if isinstance(local_array, np.ndarray):
                base = local_array
                while isinstance(getattr(base, "base", None), np.ndarray):
                    base = base.base
                _PIN_CANDIDATES.pop(base.ctypes.data, None)
                _lib.unpin_host_array(base)
        except Exception:        # (objectives are also torn down at interpreter exit, objectivefns.py:4436-4441)
            pass
        return super().free_local_array(local_array)


# ---- objective side ------------------------------------------------------------------------------------------------------
def raw_objective_descriptor(raw_objfn):
    """('chi2' | 'logl', min_prob_clip, radius) of a raw objective the device's element-wise maps implement
    (gst_objective_rows_dev: RawChi2Function; RawPoissonPicDeltaLogLFunction with the 'minp' regularisation and a
    zero-frequency radius -- the two the stock GST protocol builds, protocols/gst.py:790-834), or None."""
    name = type(raw_objfn).__name__
    if name == "RawChi2Function":
        return "chi2", float(raw_objfn.min_prob_clip_for_weighting), 1e-4
    if name == "RawPoissonPicDeltaLogLFunction" and getattr(raw_objfn, "regtype", None) == "minp" \
            and getattr(raw_objfn, "radius", None) is not None:
        return "logl", float(raw_objfn.min_p), float(raw_objfn.radius)
    return None


class DeviceLMStepLogic:
    """`dlsvec` of TimeIndependentMDCObjectiveFunction with the element rows on the device.  Expects on `self` what the
    reference's class has: model (with .sim, .from_vector, .to_vector, .num_params), layout, raw_objfn, counts,
    total_counts, probs, obj, nelements, ex, local_ex, firsts, prob_clip_interval, resource_alloc,
    _process_penalties, _dterms_fill_penalty, lsvec."""

    device_lm_step = True           # set False on an instance to force the reference's host algebra
    last_dlsvec_route = None        # 'device' | 'host' (for tests and logs)

    def _device_step_blocker(self):
        """None when the device step applies, else the reason it does not (then the reference's own dlsvec runs, over a
        Jacobian the device fills into the host array)."""
        if not self.device_lm_step:
            return "disabled on this objective"
        sim = getattr(self.model, "sim", None)
        if not hasattr(sim, "device_scaled_jacobian"):
            return "the model's simulator is not a device simulator"
        if not isinstance(self.layout, LayoutNormalEquations):
            return "the layout was not created by the device simulator"
        if raw_objective_descriptor(self.raw_objfn) is None:
            return "raw objective %s has no device maps" % type(self.raw_objfn).__name__
        if getattr(self, "firsts", None) is not None:
            return "data set with omitted outcomes (zero-frequency corrections couple rows)"
        ralloc = getattr(self, "resource_alloc", None)
        if getattr(ralloc, "comm", None) is not None:
            return "MPI run: the reference's own distributed products are used"
        for name in ("_reweight_terms", "_reweight_jac"):
            mine = getattr(type(self), name, None)
            base = getattr(DeviceLMStepLogic._reference_base(type(self)), name, None)
            if mine is not None and base is not None and mine is not base:
                return "objective re-weights its rows"
        return None

    @staticmethod
    def _reference_base(cls):
        """the first class after the mixin in the MRO that is not ours (pyGSTi's objective class)"""
        for c in cls.__mro__:
            if c.__name__ == "TimeIndependentMDCObjectiveFunction":
                return c
        return object

    def dlsvec(self, paramvec=None):
        why = self._device_step_blocker()
        if why is not None:
            self.last_dlsvec_route = "host"
            self.last_dlsvec_blocker = why
            return super().dlsvec(paramvec)
        self.last_dlsvec_route = "device"
        if paramvec is None:
            paramvec = self.model.to_vector()
        else:
            self.model.from_vector(paramvec)
        kind, mpc, radius = raw_objective_descriptor(self.raw_objfn)
        nE, nP = int(self.nelements), int(self.model.num_params)
        lsvec = self.obj                        # (nE + ex): the array `lsvec()` returns a view of (objectivefns.py:4548)
        parts, jtj = self.model.sim.device_scaled_jacobian(
            self.layout, self.counts, self.total_counts, kind, mpc, radius, self.prob_clip_interval,
            lsvec_to_fill=lsvec[:nE], pr_array_to_fill=self.probs)
        pen = None
        ex = int(getattr(self, "local_ex", 0))
        if ex > 0:
            # the penalty rows exactly as dterms / dlsvec form them (objectivefns.py:4626-4627, 4657-4662), on their own
            # small array instead of the tail of the (nE + ex, nP) one
            pen = np.zeros((ex, nP))
            self._dterms_fill_penalty(paramvec, pen)
            terms_pen = self._terms_penalty(paramvec)
            lsvec[nE:] = np.sqrt(terms_pen)
            ls_pen = lsvec[nE:]
            with np.errstate(divide="ignore", invalid="ignore"):
                p5 = 0.5 / ls_pen
            p5[np.abs(ls_pen) < 1e-100] = 0.0
            pen *= p5[:, None]
        return DeviceJacobian(parts, jtj, nE, nP, pen)
