"""oracle/oracle.py -- ctypes face of the CPU checker.  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module
(the product path, pygsti_amd/, must never do so; tests/test_layout_rules.py greps for it).

Two back ends with one signature:
  * "port"      -> oracle/liboracle.so       (our C restatement, mapfill_oracle.c)
  * "reference" -> oracle/_ref/libgst_ref.so (the reference's own C++ reps + ref_driver.cpp)

Plus `analytic_dprobs`: a small numpy forward/backward analytic Jacobian, the independent check
SURVEY.md section 8(c) asks for; it is validated against MatrixForwardSimulator golden vectors
(pygsti/forwardsims/matrixforwardsim.py:1059-1140 computes the same sums over an eval tree).
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


class _Plan(C.Structure):
    _fields_ = [("D", C.c_int32), ("n_rows", C.c_int32), ("cache_size", C.c_int32),
                ("t_dest", C.c_void_p), ("t_start", C.c_void_p), ("t_cache", C.c_void_p), ("t_rho", C.c_void_p),
                ("row_ptr", C.c_void_p), ("gate_idx", C.c_void_p),
                ("eff_ptr", C.c_void_p), ("eff_label", C.c_void_p), ("eff_dest", C.c_void_p)]


class _Model(C.Structure):
    _fields_ = [("nG", C.c_int32), ("nR", C.c_int32), ("nEl", C.c_int32),
                ("gates", C.c_void_p), ("rhos", C.c_void_p), ("effects", C.c_void_p),
                ("nP", C.c_int32), ("pkind", C.c_void_p), ("pobj", C.c_void_p), ("pelem", C.c_void_p)]


REF_SO = os.path.join(HERE, "_ref", "libgst_ref.so")
REF_HASH = os.path.join(HERE, "ref_build.sha256")      # tracked: the hash of the library `make _ref` produced where /root/reference exists


def _sha256(path):
    import hashlib
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for chunk in iter(lambda: f.read(1 << 20), b""):
            h.update(chunk)
    return h.hexdigest()


def build(ref=True):
    """Compile liboracle.so (always) and _ref/libgst_ref.so (when /root/reference is present).  Where the reference build
    happens its hash is recorded in oracle/ref_build.sha256 (committed); everywhere else -- the GPU box, where the library
    arrives prebuilt and cannot be rebuilt -- `verify_ref()` checks the shipped binary against that record."""
    subprocess.check_call(["make", "-s", "-C", HERE])
    if ref and os.path.isdir("/root/reference/pygsti/evotypes/densitymx"):
        subprocess.check_call(["make", "-s", "-C", HERE, "_ref"])
        digest = _sha256(REF_SO)
        have = open(REF_HASH).read().split()[0] if os.path.exists(REF_HASH) else None
        if have != digest:
            with open(REF_HASH, "w") as f:
                f.write(digest + "  oracle/_ref/libgst_ref.so  (g++ -O3 -std=c++11 -ffp-contract=off, oracle/Makefile)\n")


_ref_verified = None


def verify_ref():
    """True when oracle/_ref/libgst_ref.so is the binary oracle/ref_build.sha256 records (checked once per process); raises
    when a library is present that is NOT that binary -- a "reference"-kind check must never run on something else."""
    global _ref_verified
    if _ref_verified is None:
        if not os.path.exists(REF_SO):
            _ref_verified = False
        elif not os.path.exists(REF_HASH):
            raise RuntimeError("oracle/_ref/libgst_ref.so exists but oracle/ref_build.sha256 does not: rebuild with oracle.build()")
        else:
            want = open(REF_HASH).read().split()[0]
            got = _sha256(REF_SO)
            if want != got:
                raise RuntimeError("oracle/_ref/libgst_ref.so (sha256 %s) is not the build recorded in oracle/ref_build.sha256 (%s)" % (got[:16], want[:16]))
            _ref_verified = True
    return _ref_verified


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class Oracle:
    """Holds one table-format plan + model (the arrays of a golden fixture or of a host layout)."""

    def __init__(self, tbl, model, kind="port"):
        """tbl: dict with D, cache_size, t_dest, t_start, t_cache, t_rho, row_ptr, gate_idx,
        eff_ptr, eff_label, eff_dest, nE.  model: dict with gates, rhos, effects, pkind, pobj, pelem."""
        path = os.path.join(HERE, "liboracle.so") if kind == "port" else os.path.join(HERE, "_ref", "libgst_ref.so")
        if not os.path.exists(path):
            raise RuntimeError("oracle library %s is not built (run oracle.build())" % path)
        if kind != "port":
            verify_ref()
        self.kind = kind
        self.lib = C.CDLL(path)
        self.prefix = "oracle_" if kind == "port" else "ref_"
        i32 = lambda x: np.ascontiguousarray(x, dtype=np.int32)
        i64 = lambda x: np.ascontiguousarray(x, dtype=np.int64)
        f64 = lambda x: np.ascontiguousarray(x, dtype=np.float64)
        self.D = int(tbl["D"])
        self.nE = int(tbl["nE"])
        self._keep = dict(
            t_dest=i32(tbl["t_dest"]), t_start=i32(tbl["t_start"]), t_cache=i32(tbl["t_cache"]),
            t_rho=i32(tbl["t_rho"]), row_ptr=i64(tbl["row_ptr"]), gate_idx=i32(tbl["gate_idx"]),
            eff_ptr=i64(tbl["eff_ptr"]), eff_label=i32(tbl["eff_label"]), eff_dest=i32(tbl["eff_dest"]),
            gates=f64(model["gates"]), rhos=f64(model["rhos"]), effects=f64(model["effects"]),
            pkind=i32(model["pkind"]), pobj=i32(model["pobj"]), pelem=i32(model["pelem"]))
        k = self._keep
        if len(k["gate_idx"]) == 0:
            k["gate_idx"] = np.zeros(1, np.int32)
        self.plan = _Plan(self.D, len(k["t_dest"]), int(tbl["cache_size"]),
                          _ptr(k["t_dest"]), _ptr(k["t_start"]), _ptr(k["t_cache"]), _ptr(k["t_rho"]),
                          _ptr(k["row_ptr"]), _ptr(k["gate_idx"]),
                          _ptr(k["eff_ptr"]), _ptr(k["eff_label"]), _ptr(k["eff_dest"]))
        self.nP = len(k["pkind"])
        self.model = _Model(k["gates"].shape[0], k["rhos"].shape[0], k["effects"].shape[0],
                            _ptr(k["gates"]), _ptr(k["rhos"]), _ptr(k["effects"]),
                            self.nP, _ptr(k["pkind"]), _ptr(k["pobj"]), _ptr(k["pelem"]))

    def set_model(self, gates, rhos, effects):
        k = self._keep
        k["gates"][...] = gates; k["rhos"][...] = rhos; k["effects"][...] = effects

    def _fn(self, name):
        return getattr(self.lib, self.prefix + name)

    def probs(self):
        out = np.empty(self.nE, np.float64)
        rc = self._fn("fill_probs")(C.byref(self.plan), C.byref(self.model), _ptr(out))
        assert rc == 0
        return out

    def dprobs(self, param_idx=None, eps=1e-7, return_probs=False):
        if param_idx is None:
            param_idx = np.arange(self.nP)
        param_idx = np.ascontiguousarray(param_idx, np.int64)
        n = len(param_idx)
        out = np.empty((self.nE, n), np.float64)
        pr = np.empty(self.nE, np.float64)
        rc = self._fn("fill_dprobs")(C.byref(self.plan), C.byref(self.model), C.c_int64(self.nE),
                                     _ptr(param_idx), None, C.c_int64(n), C.c_double(eps),
                                     _ptr(out), C.c_int64(n), _ptr(pr))
        assert rc == 0
        return (out, pr) if return_probs else out

    def hprobs(self, param_idx1, param_idx2, eps=1e-5):
        assert self.kind == "port"
        p1 = np.ascontiguousarray(param_idx1, np.int64)
        p2 = np.ascontiguousarray(param_idx2, np.int64)
        out = np.empty((self.nE, len(p1), len(p2)), np.float64)
        rc = self.lib.oracle_fill_hprobs(C.byref(self.plan), C.byref(self.model), C.c_int64(self.nE),
                                         _ptr(p1), None, C.c_int64(len(p1)), _ptr(p2), None, C.c_int64(len(p2)),
                                         C.c_double(eps), _ptr(out), C.c_int64(len(p1)), C.c_int64(len(p2)))
        assert rc == 0
        return out

    def time_passes(self, n_pass):
        """Run n_pass full probability passes (the cost of n_pass FD columns); returns seconds."""
        import time
        out = np.empty(self.nE, np.float64)
        t0 = time.perf_counter()
        rc = self._fn("time_passes")(C.byref(self.plan), C.byref(self.model), C.c_int64(self.nE),
                                     C.c_int(n_pass), _ptr(out))
        assert rc == 0
        return time.perf_counter() - t0


def from_fixture(fx, kind="port"):
    """Build an Oracle from a tests/golden/*.npz fixture (np.load result or dict)."""
    tbl = {k: fx[k] for k in ("D", "cache_size", "t_dest", "t_start", "t_cache", "t_rho", "row_ptr",
                              "gate_idx", "eff_ptr", "eff_label", "eff_dest", "nE")}
    mdl = {k: fx[k] for k in ("gates", "rhos", "effects", "pkind", "pobj", "pelem")}
    return Oracle(tbl, mdl, kind)


# ------------------------------------------------------------------------------------------------
# numpy analytic Jacobian / Hessian (independent check for the analytic device mode)
# ------------------------------------------------------------------------------------------------

def expand_table_circuits(fx):
    """Full gate string + rho index of every expanded circuit, rebuilt from the prefix table."""
    R = len(fx["t_dest"])
    full = [None] * R
    rho = np.zeros(R, np.int32)
    cache = {}
    rp = fx["row_ptr"]
    for k in range(R):
        seg = list(fx["gate_idx"][rp[k]:rp[k + 1]])
        if fx["t_start"][k] == -1:
            pre, r = [], int(fx["t_rho"][k])
        else:
            pre, r = cache[int(fx["t_start"][k])]
        s = pre + seg
        i = int(fx["t_dest"][k])
        full[i] = s
        rho[i] = r
        if fx["t_cache"][k] != -1:
            cache[int(fx["t_cache"][k])] = (s, r)
    return full, rho


def analytic_dprobs(fx, cols=None):
    """dp/dtheta by forward states and backward effect vectors (exact derivative, `full` params)."""
    D = int(fx["D"]); nE = int(fx["nE"]); nP = len(fx["pkind"])
    gates, rhos, effects = fx["gates"], fx["rhos"], fx["effects"]
    full, rho = expand_table_circuits(fx)
    # parameter index of each dense element
    gmap = -np.ones((gates.shape[0], D * D), np.int64)
    rmap = -np.ones((rhos.shape[0], D), np.int64)
    emap = -np.ones((effects.shape[0], D), np.int64)
    for p in range(nP):
        k, o, e = int(fx["pkind"][p]), int(fx["pobj"][p]), int(fx["pelem"][p])
        if k == 0: gmap[o, e] = p
        elif k == 1: rmap[o, e] = p
        elif k == 2: emap[o, e] = p
    J = np.zeros((nE, nP))
    P = np.zeros(nE)
    for i, s in enumerate(full):
        fwd = [rhos[rho[i]]]
        for g in s:
            fwd.append(gates[g] @ fwd[-1])
        for x in range(fx["eff_ptr"][i], fx["eff_ptr"][i + 1]):
            el, dest = int(fx["eff_label"][x]), int(fx["eff_dest"][x])
            P[dest] = effects[el] @ fwd[-1]
            J[dest, emap[el][emap[el] >= 0]] += fwd[-1][emap[el] >= 0]
            b = effects[el].copy()
            for pos in range(len(s) - 1, -1, -1):
                g = s[pos]
                outer = np.outer(b, fwd[pos]).ravel()
                m = gmap[g] >= 0
                np.add.at(J[dest], gmap[g][m], outer[m])
                b = gates[g].T @ b
            m = rmap[rho[i]] >= 0
            J[dest, rmap[rho[i]][m]] += b[m]
    if cols is not None:
        J = J[:, cols]
    return J, P


def element_param_map(fx):
    """(kind, obj, elem) of the `full` element layout [rhos | effects | gates] -- the columns of the element Jacobian
    that the chain rule of general parameterisations starts from."""
    D = int(fx["D"]); nG, nR, nEl = len(fx["gates"]), len(fx["rhos"]), len(fx["effects"])
    kind = np.concatenate([np.full(nR * D, 1), np.full(nEl * D, 2), np.full(nG * D * D, 0)]).astype(np.int32)
    obj = np.concatenate([np.repeat(np.arange(nR), D), np.repeat(np.arange(nEl), D), np.repeat(np.arange(nG), D * D)]).astype(np.int32)
    elem = np.concatenate([np.tile(np.arange(D), nR), np.tile(np.arange(D), nEl), np.tile(np.arange(D * D), nG)]).astype(np.int32)
    return kind, obj, elem


def derivs_from_fixture(fx):
    """[(kind, obj, param_idx[n], deriv[n_elem, n])] from the flat dv_* arrays of a general-parameterisation fixture."""
    out, p0, d0 = [], 0, 0
    D = int(fx["D"])
    for k, o, n in zip(fx["dv_kind"], fx["dv_obj"], fx["dv_ncols"]):
        ne = D * D if k == 0 else D
        out.append((int(k), int(o), np.asarray(fx["dv_param_idx"][p0:p0 + n], np.int64),
                    np.asarray(fx["dv_deriv"][d0:d0 + ne * n], np.float64).reshape(ne, n)))
        p0 += n; d0 += ne * n
    return out


def tp_param_map(fx):
    """(kind, obj, elem) per parameter of a "full TP" fixture, read off the members' deriv_wrt_params: TPState, FullTPOp
    and the non-complement effects of a TPPOVM are one parameter <-> one dense element (a single 1.0 per derivative
    column); the complement effect (columns of -1.0, fx["comp_index"]) carries no parameter of its own."""
    nP = int(fx["nP"])
    if "tp_kind" in fx:       # fixtures too large for the derivative tensors (D = 64) carry the map itself
        return np.asarray(fx["tp_kind"], np.int32), np.asarray(fx["tp_obj"], np.int32), np.asarray(fx["tp_elem"], np.int32)
    pk = np.full(nP, -1, np.int32); po = np.zeros(nP, np.int32); pe = np.zeros(nP, np.int32)
    comp = int(fx["comp_index"]) if "comp_index" in fx else -1
    for k, o, pidx, dm in derivs_from_fixture(fx):
        if k == 2 and o == comp:
            assert set(np.unique(dm)) <= {0.0, -1.0}
            continue
        for j, pi in enumerate(pidx):
            rows = np.nonzero(dm[:, j])[0]
            assert len(rows) == 1 and dm[rows[0], j] == 1.0, "not a one-parameter-per-element member"
            assert pk[pi] == -1
            pk[pi], po[pi], pe[pi] = k, o, rows[0]
    return pk, po, pe


def second_derivs_from_fixture(fx):
    """Per object of derivs_from_fixture: hessian_wrt_params [n_elem, n, n], or None for a linear member."""
    out, h0 = [], 0
    D = int(fx["D"])
    for k, n, nz in zip(fx["dv_kind"], fx["dv_ncols"], fx["dv2_nonzero"]):
        ne = D * D if k == 0 else D
        if nz:
            out.append(np.asarray(fx["dv2_hess"][h0:h0 + ne * n * n], np.float64).reshape(ne, n, n))
            h0 += ne * n * n
        else:
            out.append(None)
    return out


def analytic_dprobs_general(fx, cols=None):
    """Jacobian w.r.t. model parameters of a general (TP, CPTP, ...) parameterisation: element Jacobian x the members'
    deriv_wrt_params -- what MatrixForwardSimulator._dprobs_from_rho_e assembles from `_doperation`
    (pygsti/forwardsims/matrixforwardsim.py:126-190, 1100-1180)."""
    D = int(fx["D"]); nG, nR, nEl = len(fx["gates"]), len(fx["rhos"]), len(fx["effects"])
    fe = dict(fx)
    fe["pkind"], fe["pobj"], fe["pelem"] = element_param_map(fx)
    Je, P = analytic_dprobs(fe)
    nP = int(fx["nP"])
    J = np.zeros((Je.shape[0], nP))
    base = {1: 0, 2: nR * D, 0: nR * D + nEl * D}
    for k, o, pidx, dm in derivs_from_fixture(fx):
        ne = D * D if k == 0 else D
        a = base[k] + o * ne
        np.add.at(J, (slice(None), pidx), Je[:, a:a + ne] @ dm)
    if cols is not None:
        J = J[:, cols]
    return J, P


def analytic_hprobs_general(fx, idx1, idx2):
    """Exact Hessian block of a LINEAR general parameterisation (TP, ...): the element Hessian contracted with the
    members' deriv_wrt_params on both sides (their hessian_wrt_params vanish) -- what MatrixForwardSimulator's
    _hprobs_from_rho_e returns for such models (matrixforwardsim.py:1190-1287)."""
    D = int(fx["D"]); nR, nEl = len(fx["rhos"]), len(fx["effects"])
    fe = dict(fx)
    fe["pkind"], fe["pobj"], fe["pelem"] = element_param_map(fx)
    base = {1: 0, 2: nR * D, 0: nR * D + nEl * D}
    idx1 = np.asarray(idx1); idx2 = np.asarray(idx2)

    def weights(idx):
        pos = {int(p): k for k, p in enumerate(idx)}
        trip = []
        for k, o, pidx, dm in derivs_from_fixture(fx):
            ne = D * D if k == 0 else D
            a = base[k] + o * ne
            for c, pi in enumerate(pidx):
                if int(pi) in pos:
                    for r in np.nonzero(dm[:, c])[0]:
                        trip.append((a + int(r), pos[int(pi)], dm[r, c]))
        elems = sorted(set(t[0] for t in trip))
        W = np.zeros((len(elems), len(idx)))
        for a, j, w in trip:
            W[elems.index(a), j] += w
        return np.array(elems, np.int64), W
    e1, W1 = weights(idx1)
    e2, W2 = weights(idx2)
    He = analytic_hprobs(fe, e1, e2)
    H = np.einsum("eab,ai,bj->eij", He, W1, W2)
    # members that are not linear in their parameters (CPTPLND, ...): + sum_a (d p / d elem_a) d^2 elem_a / d p1 d p2,
    # with the members' hessian_wrt_params (fixture arrays dv2_*), as _hoperation feeds it into the same assembly
    if "dv2_nonzero" in fx and np.any(fx["dv2_nonzero"]):
        Je, _ = analytic_dprobs(fe)
        pos1 = {int(p): k for k, p in enumerate(idx1)}
        pos2 = {int(p): k for k, p in enumerate(idx2)}
        for (k, o, pidx, dm), S in zip(derivs_from_fixture(fx), second_derivs_from_fixture(fx)):
            if S is None:
                continue
            ne = D * D if k == 0 else D
            a = base[k] + o * ne
            c1 = [(c, pos1[int(pi)]) for c, pi in enumerate(pidx) if int(pi) in pos1]
            c2 = [(c, pos2[int(pi)]) for c, pi in enumerate(pidx) if int(pi) in pos2]
            for ca, i in c1:
                for cb, j in c2:
                    H[:, i, j] += Je[:, a:a + ne] @ S[:, ca, cb]
    return H


def analytic_hprobs(fx, idx1, idx2):
    """d^2 p / d theta_1 d theta_2 for the `full` parameterisation, exactly (what MatrixForwardSimulator's
    _hprobs_from_rho_e assembles from hProdCache, matrixforwardsim.py:1190-1287), by derivative states:
        dF_k = G_k dF_{k-1} + [g_k = g1] e_a1 F_{k-1}[b1]          (theta_1 earlier in the chain than theta_2)
        dB_{k-1} = G_k^T dB_k + [g_k = g1] e_b1 B_k[a1]            (theta_1 later)
        H = sum_{k: g_k = g2} B_k[a2] dF_{k-1}[b2] + dB_k[a2] F_{k-1}[b2]    (+ the SPAM cases)
    Pinned against the reference's Matrix-simulator vectors (tests/test_oracle.py)."""
    D = int(fx["D"]); nE = int(fx["nE"])
    gates, rhos, effects = fx["gates"], fx["rhos"], fx["effects"]
    full, rho = expand_table_circuits(fx)
    pk, po, pe = fx["pkind"], fx["pobj"], fx["pelem"]
    idx1 = np.asarray(idx1); idx2 = np.asarray(idx2)
    H = np.zeros((nE, len(idx1), len(idx2)))
    for i, s in enumerate(full):
        n = len(s)
        F = [rhos[rho[i]]]
        for g in s:
            F.append(gates[g] @ F[-1])
        for x in range(fx["eff_ptr"][i], fx["eff_ptr"][i + 1]):
            el, dest = int(fx["eff_label"][x]), int(fx["eff_dest"][x])
            B = [None] * (n + 1)
            B[n] = effects[el]
            for k in range(n, 0, -1):
                B[k - 1] = gates[s[k - 1]].T @ B[k]
            for ii, p1 in enumerate(idx1):
                k1, o1, e1 = int(pk[p1]), int(po[p1]), int(pe[p1])
                dF = [np.zeros(D) for _ in range(n + 1)]
                dB = [np.zeros(D) for _ in range(n + 1)]
                if k1 == 1 and o1 == rho[i]:
                    dF[0][e1] = 1.0
                if k1 == 2 and o1 == el:
                    dB[n][e1] = 1.0
                a1, b1 = divmod(e1, D)
                for k in range(1, n + 1):
                    dF[k] = gates[s[k - 1]] @ dF[k - 1]
                    if k1 == 0 and s[k - 1] == o1:
                        dF[k][a1] += F[k - 1][b1]
                for k in range(n, 0, -1):
                    dB[k - 1] = gates[s[k - 1]].T @ dB[k]
                    if k1 == 0 and s[k - 1] == o1:
                        dB[k - 1][b1] += B[k][a1]
                for jj, p2 in enumerate(idx2):
                    k2, o2, e2 = int(pk[p2]), int(po[p2]), int(pe[p2])
                    if k2 == 0:
                        a2, b2 = divmod(e2, D)
                        H[dest, ii, jj] = sum(B[k][a2] * dF[k - 1][b2] + dB[k][a2] * F[k - 1][b2]
                                              for k in range(1, n + 1) if s[k - 1] == o2)
                    elif k2 == 1:
                        H[dest, ii, jj] = dB[0][e2] if o2 == rho[i] else 0.0
                    elif k2 == 2:
                        H[dest, ii, jj] = dF[n][e2] if o2 == el else 0.0
    return H

