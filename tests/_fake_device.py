"""A RECORDING STAND-IN for `_lib.Plan` -- TEST INFRASTRUCTURE for the CPU container only.

tests/test_pygsti_lmstep.py runs the real pyGSTi's `GateSetTomography.run(simulator=HipMapForwardSimulator())` where no
GPU exists.  What is under test there is the BINDING (objective class -> DeviceJacobian -> layout.fill_jtj / fill_jtf ->
the reference's own optimizer), not the kernels: this class answers the subset of the `Plan` API that binding uses with
the CPU checker (oracle/, numpy) behind fake device pointers, and records every call so that the test can assert which
route a fit took.  The product never imports it.
"""
import numpy as np

from oracle import oracle as O
from oracle import objective_oracle as OO


class FakePlan:
    log = []                     # (method name, detail) of every call, class-wide
    host_jacobian_fills = 0      # fill_dprobs calls whose destination is a host (nE, nP) array

    def __init__(self, D, n_gates, n_rhos, n_effects, n_elements, circ_rho, circ_ptr, circ_gates, eff_ptr, eff_label, eff_dest):
        self.D, self.n_gates, self.n_rhos, self.n_effects = int(D), int(n_gates), int(n_rhos), int(n_effects)
        self.n_elements = int(n_elements)
        self.n_params = 0
        R = len(circ_rho)
        self._tbl = dict(D=D, nE=n_elements, cache_size=0, t_dest=np.arange(R, dtype=np.int32), t_start=-np.ones(R, np.int32),
                         t_cache=-np.ones(R, np.int32), t_rho=np.asarray(circ_rho, np.int32), row_ptr=np.asarray(circ_ptr, np.int64),
                         gate_idx=np.asarray(circ_gates, np.int32), eff_ptr=np.asarray(eff_ptr, np.int64),
                         eff_label=np.asarray(eff_label, np.int32), eff_dest=np.asarray(eff_dest, np.int32))
        self._model = None
        self._pmap = None
        self._oracle = None
        self._mem = {}
        self._next = 0x10000000
        self._workspaces = {}

    # ---- construction hooks (monkeypatched over _lib.Plan.from_circuits) ------------------------------------------------
    @classmethod
    def from_circuits(cls, D, n_gates, n_rhos, n_effects, n_elements, circ_rho, circ_ptr, circ_gates, eff_ptr, eff_label,
                      eff_dest, device=-1, target_tasks=0, **kw):
        cls.log.append(("from_circuits", len(circ_rho)))
        return cls(D, n_gates, n_rhos, n_effects, n_elements, circ_rho, circ_ptr, circ_gates, eff_ptr, eff_label, eff_dest)

    # ---- model ----------------------------------------------------------------------------------------------------------
    def set_model(self, gates, rhos, effects):
        self._model = (np.array(gates, np.float64).reshape(self.n_gates, self.D, self.D),
                       np.array(rhos, np.float64).reshape(self.n_rhos, self.D),
                       np.array(effects, np.float64).reshape(self.n_effects, self.D))
        if self._oracle is not None:
            self._oracle.set_model(*self._model)

    def set_param_map(self, kind, obj, elem):
        self._pmap = (np.asarray(kind, np.int32), np.asarray(obj, np.int32), np.asarray(elem, np.int32))
        self.n_params = len(kind)
        self._oracle = None

    def set_derivs(self, n_params, objs):
        assert not list(objs), "the stand-in implements element maps only"

    def set_complement_effect(self, comp_index, identity=None, others=()):
        assert comp_index is None or comp_index < 0, "the stand-in implements element maps only"

    def _orc(self):
        if self._oracle is None:
            k, o, e = self._pmap if self._pmap is not None else (np.zeros(0, np.int32),) * 3
            G, R, E = self._model
            self._oracle = O.Oracle(self._tbl, dict(gates=G, rhos=R, effects=E, pkind=k, pobj=o, pelem=e), "port")
        return self._oracle

    # ---- host fills -----------------------------------------------------------------------------------------------------
    def fill_probs(self, out=None):
        FakePlan.log.append(("fill_probs", self.n_elements))
        p = self._orc().probs()
        if out is None:
            return p
        out[...] = p
        return out

    def fill_dprobs(self, out=None, param_idx=None, dest_idx=None, eps=1e-7, probs_out=None, mode=0):
        pidx = np.arange(self.n_params) if param_idx is None else np.asarray(param_idx, np.int64)
        FakePlan.log.append(("fill_dprobs", (self.n_elements, len(pidx))))
        FakePlan.host_jacobian_fills += 1
        assert mode == 0, "the stand-in computes finite differences"
        J, pr = self._orc().dprobs(pidx, eps=eps, return_probs=True)
        if probs_out is not None:
            probs_out[...] = pr
        if out is None:
            return J
        out[:, np.arange(len(pidx)) if dest_idx is None else np.asarray(dest_idx)] = J
        return out

    # ---- "device" memory ------------------------------------------------------------------------------------------------
    def workspace(self, name, nbytes):
        have = self._workspaces.get(name)
        if have is not None and have[1] >= nbytes:
            return have[0]
        ptr = self.device_malloc(max(int(nbytes), 8))
        self._workspaces[name] = (ptr, int(nbytes))
        return ptr

    def device_malloc(self, nbytes):
        ptr = self._next
        self._next += (int(nbytes) + 255) // 256 * 256 + 256
        self._mem[ptr] = np.full(int(nbytes) // 8, np.nan)
        return ptr

    def _dev(self, ptr, n):
        return self._mem[int(ptr)][:n]

    def memcpy_h2d(self, d_ptr, arr, offset_bytes=0):
        a = np.ascontiguousarray(arr, np.float64).ravel()
        FakePlan.log.append(("memcpy_h2d", a.nbytes))
        self._dev(d_ptr, a.size)[...] = a

    def memcpy_d2h(self, out, d_ptr, offset_bytes=0):
        FakePlan.log.append(("memcpy_d2h", out.nbytes))
        out[...] = self._dev(d_ptr, out.size).reshape(out.shape)
        return out

    # ---- device fills ---------------------------------------------------------------------------------------------------
    def fill_dprobs_dev(self, d_out_ptr, ld, param_idx, dest_idx=None, eps=1e-7, d_probs_ptr=None, mode=0):
        pidx = np.asarray(param_idx, np.int64)
        FakePlan.log.append(("fill_dprobs_dev", (self.n_elements, len(pidx))))
        assert mode == 0 and dest_idx is None and ld == len(pidx)
        J, pr = self._orc().dprobs(pidx, eps=eps, return_probs=True)
        self._dev(d_out_ptr, J.size)[...] = J.ravel()
        if d_probs_ptr is not None:
            self._dev(d_probs_ptr, pr.size)[...] = pr

    def objective_rows_dev(self, kind, d_probs, d_counts, d_totals, n, d_lsvec, d_rowscale, d_terms=None,
                           min_prob_clip=1e-4, radius=1e-4, prob_clip_interval=None, want_sum=True):
        FakePlan.log.append(("objective_rows_dev", kind))
        p = self._dev(d_probs, n)
        if prob_clip_interval is not None and prob_clip_interval[0] < prob_clip_interval[1]:
            np.clip(p, prob_clip_interval[0], prob_clip_interval[1], out=p)
        k = {"chi2": OO.CHI2, "logl": OO.DLOGL}[kind]
        t, ls, _, rs = OO.objective_rows(k, p, self._dev(d_counts, n), self._dev(d_totals, n), min_prob_clip, radius)
        self._dev(d_lsvec, n)[...] = ls
        self._dev(d_rowscale, n)[...] = rs
        if d_terms is not None:
            self._dev(d_terms, n)[...] = t
        return float(t.sum()) if want_sum else None

    def fill_jtj_dev(self, d_J, n_rows, n_cols, ld, d_jtj, d_row_scale=None):
        FakePlan.log.append(("fill_jtj_dev", (n_rows, n_cols)))
        J = self._dev(d_J, n_rows * ld).reshape(n_rows, ld)
        if d_row_scale is not None:
            J *= self._dev(d_row_scale, n_rows)[:, None]
        self._dev(d_jtj, n_cols * n_cols)[...] = (J[:, :n_cols].T @ J[:, :n_cols]).ravel()

    def fill_jtf_dev(self, d_J, n_rows, n_cols, ld, d_f, d_jtf):
        FakePlan.log.append(("fill_jtf_dev", (n_rows, n_cols)))
        J = self._dev(d_J, n_rows * ld).reshape(n_rows, ld)
        self._dev(d_jtf, n_cols)[...] = J[:, :n_cols].T @ self._dev(d_f, n_rows)

    def fill_normal_eqs_dev(self, d_J, n_rows, n_cols, ld, d_row_scale=None, d_f=None, d_jtj=None, d_jtf=None):
        J = self._dev(d_J, n_rows * ld).reshape(n_rows, ld)[:, :n_cols]
        Js = J if d_row_scale is None else J * self._dev(d_row_scale, n_rows)[:, None]      # (J itself stays as it is)
        if d_jtj is not None:
            FakePlan.log.append(("fill_jtj_dev", (n_rows, n_cols)))
            self._dev(d_jtj, n_cols * n_cols)[...] = (Js.T @ Js).ravel()
        if d_jtf is not None:
            FakePlan.log.append(("fill_jtf_dev", (n_rows, n_cols)))
            self._dev(d_jtf, n_cols)[...] = Js.T @ self._dev(d_f, n_rows)

    def sync(self):
        pass
