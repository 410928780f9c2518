"""Description of an implicit model's layer operations for gst_set_composite, and its host restatement.

pyGSTi's implicit models (LocalNoiseModel / CloudNoiseModel, `create_crosstalk_free_model` ...) build a circuit layer as a
ComposedOp of EmbeddedOps over a handful of small "leaf" operations (pygsti/modelmembers/operations/composedop.py,
embeddedop.py; reps evotypes/densitymx/opcreps.cpp:93-158, 242-276); with `independent_gates=False` one leaf stands behind
several layers, so its parameters are shared.  `CompositeModel` holds exactly that structure -- leaves, factors = (leaf,
target qubits), layers = ordered factor lists -- in the arrays the C ABI takes, plus a plain numpy restatement of what the
device builds from it (dense layers, the dense model after every finite-difference step, the layers' derivative matrices):
the checker of tests/test_composite.py and tests/test_gpu_composite.py, and the reading aid for
csrc/gst_kernels_composite.hip.  No pyGSTi import here (pygsti_adapter.atom_composite extracts it from a real model).

Index convention: the register's Pauli-product basis is the Kronecker product in qubit order, so a state index is a
base-4 number with qubit 0 as its most significant digit; Emb(f)[i][j] = f[digits_T(i)][digits_T(j)] when i and j agree
outside the target qubits T, else 0.
"""
import numpy as np


class CompositeModel:
    def __init__(self, D, n_params, leaf_dims, leaf_params, leaf_static, gate_factors, leaf_general=None):
        """leaf_dims[l] in (4, 16, 64); leaf_params[l]: int64 [dl*dl] parameter index of each element (row-major) or -1;
        leaf_static[l]: float64 [dl*dl] values of the elements that are no parameter (others ignored);
        gate_factors[g] = [(leaf, (target qubits...)), ...] in order of application (first factor acts first);
        leaf_general[l]: None for element leaves, or the ascending int64 array of the model parameters of a GENERAL leaf
        (its elements are functions of those parameters: the caller supplies values / derivatives / stepped values)."""
        self.D, self.num_params = int(D), int(n_params)
        self.nq = {4: 1, 16: 2, 64: 3}[self.D]
        self.leaf_dims = [int(d) for d in leaf_dims]
        self.leaf_params = [np.asarray(p, np.int64).ravel() for p in leaf_params]
        self.leaf_static = [np.asarray(s, np.float64).ravel() for s in leaf_static]
        self.gate_factors = [[(int(l), tuple(int(t) for t in tg)) for l, tg in fs] for fs in gate_factors]
        for d, p, s in zip(self.leaf_dims, self.leaf_params, self.leaf_static):
            assert d in (4, 16, 64) and d <= self.D and len(p) == d * d == len(s)
        for fs in self.gate_factors:
            for l, tg in fs:
                assert len(tg) == {4: 1, 16: 2, 64: 3}[self.leaf_dims[l]] and len(set(tg)) == len(tg) and all(0 <= t < self.nq for t in tg)
        self.leaf_off = np.concatenate([[0], np.cumsum([d * d for d in self.leaf_dims])]).astype(np.int64)
        self.leaf_general = [None] * len(self.leaf_dims) if leaf_general is None else \
            [None if g is None else np.asarray(g, np.int64) for g in leaf_general]
        for g, p in zip(self.leaf_general, self.leaf_params):
            assert g is None or ((p == -1).all() and len(g) > 0 and (np.diff(g) > 0).all())

    # ---- the arrays of gst_composite_desc ---------------------------------------------------------------------------------
    def arrays(self):
        leaf_dim = np.asarray(self.leaf_dims, np.int32)
        leaf_param = np.concatenate(self.leaf_params).astype(np.int64) if self.leaf_params else np.zeros(0, np.int64)
        fptr = np.zeros(len(self.gate_factors) + 1, np.int32)
        fl, ft = [], []
        for g, fs in enumerate(self.gate_factors):
            for l, tg in fs:
                fl.append(l); ft.append(list(tg) + [-1] * (3 - len(tg)))
            fptr[g + 1] = len(fl)
        return leaf_dim, leaf_param, fptr, np.asarray(fl, np.int32), np.asarray(ft, np.int32).reshape(-1, 3)

    def general_arrays(self):
        """(leaf_n_params [n_leaves], concatenated parameter lists) of gst_composite_desc"""
        n = np.asarray([0 if g is None else len(g) for g in self.leaf_general], np.int32)
        lst = [g for g in self.leaf_general if g is not None]
        return n, (np.concatenate(lst).astype(np.int64) if lst else np.zeros(0, np.int64))

    @property
    def general_leaves(self):
        return [l for l, g in enumerate(self.leaf_general) if g is not None]

    def values(self, theta, general_values=None):
        """The leaves' elements (concatenated, row-major) for the model's parameter vector; general_values[l] = the dense
        elements of general leaf l at theta (their owner computes them)."""
        theta = np.asarray(theta, np.float64)
        out = []
        for l, (p, s) in enumerate(zip(self.leaf_params, self.leaf_static)):
            if self.leaf_general[l] is not None:
                out.append(np.asarray(general_values[l], np.float64).ravel())
                continue
            v = s.copy()
            v[p >= 0] = theta[p[p >= 0]]
            out.append(v)
        return np.concatenate(out) if out else np.zeros(0)

    def pack_general(self, derivs=None, fd_values=None):
        """Concatenate per-leaf dicts {l: [d*d, np]} / {l: [np, d*d]} in leaf order for gst_set_composite_general."""
        dv = None if derivs is None else np.concatenate([np.ascontiguousarray(derivs[l], np.float64).reshape(self.leaf_dims[l] ** 2, -1).ravel() for l in self.general_leaves])
        fd = None if fd_values is None else np.concatenate([np.ascontiguousarray(fd_values[l], np.float64).reshape(-1, self.leaf_dims[l] ** 2).ravel() for l in self.general_leaves])
        return dv, fd

    # ---- host restatement -------------------------------------------------------------------------------------------------
    def _digit_maps(self, targets):
        """(leaf index of every state index, state index with the target digits zeroed)"""
        i = np.arange(self.D)
        li = np.zeros(self.D, np.int64); rest = i.copy()
        for t in targets:
            sh = 2 * (self.nq - 1 - t)
            li = li * 4 + ((i >> sh) & 3)
            rest &= ~(3 << sh)
        return li, rest

    def embed(self, f, targets):
        f = np.asarray(f, np.float64).reshape(int(round(np.sqrt(np.size(f)))), -1)
        li, rest = self._digit_maps(targets)
        return np.where(rest[:, None] == rest[None, :], f[li[:, None], li[None, :]], 0.0)

    def _leaf(self, values, l):
        d = self.leaf_dims[l]
        return values[self.leaf_off[l]:self.leaf_off[l + 1]].reshape(d, d)

    def dense_gates(self, values):
        G = np.empty((len(self.gate_factors), self.D, self.D))
        for g, fs in enumerate(self.gate_factors):
            X = np.eye(self.D)
            for l, tg in fs:
                X = self.embed(self._leaf(values, l), tg) @ X
            G[g] = X
        return G

    def model_sets(self, values, rhos, effects, param_map, param_idx, eps, general_fd=None):
        """The dense model after each finite-difference step: (gates [n, nG, D, D], rhos [n, nR, D], effects [n, nEl, D]);
        param_map = (kind, obj, elem) of the SPAM parameters (GST_KIND_NONE = -1 elsewhere) or None; general_fd[l] =
        [np, d*d] the elements of general leaf l after each of its parameters' steps."""
        values = np.asarray(values, np.float64)
        leaf_param = np.concatenate(self.leaf_params)
        base = self.dense_gates(values)
        n = len(param_idx)
        G = np.repeat(base[None], n, axis=0)
        R = np.repeat(np.asarray(rhos, np.float64)[None], n, axis=0); E = np.repeat(np.asarray(effects, np.float64)[None], n, axis=0)
        for c, q in enumerate(int(x) for x in param_idx):
            hit = leaf_param == q
            v = None
            if hit.any():
                v = values.copy(); v[hit] = v[hit] + eps
            for l in self.general_leaves:
                pos = np.nonzero(self.leaf_general[l] == q)[0]
                if len(pos):
                    v = values.copy() if v is None else v
                    v[self.leaf_off[l]:self.leaf_off[l + 1]] = np.asarray(general_fd[l], np.float64).reshape(len(self.leaf_general[l]), -1)[pos[0]]
            if v is not None:
                G[c] = self.dense_gates(v)
            if param_map is not None:
                k, o, e = (int(a[q]) for a in param_map)
                if k == 1: R[c, o, e] = R[c, o, e] + eps
                elif k == 2: E[c, o, e] = E[c, o, e] + eps
        return G, R, E

    def gate_params(self, g):
        qs = {int(q) for l, _ in self.gate_factors[g] for q in self.leaf_params[l] if q >= 0}
        qs |= {int(q) for l, _ in self.gate_factors[g] if self.leaf_general[l] is not None for q in self.leaf_general[l]}
        return np.asarray(sorted(qs), np.int64)

    def _leaf_holds(self, l, q):
        return (self.leaf_params[l] == q).any() or (self.leaf_general[l] is not None and (self.leaf_general[l] == q).any())

    def _leaf_direction(self, l, q, general_derivs):
        d = self.leaf_dims[l]
        if self.leaf_general[l] is None:
            return (self.leaf_params[l] == q).astype(float).reshape(d, d)
        pos = np.nonzero(self.leaf_general[l] == q)[0][0]
        return np.asarray(general_derivs[l], np.float64).reshape(d * d, -1)[:, pos].reshape(d, d)

    def gate_derivs(self, values, general_derivs=None):
        """[(parameter indices [n], d(dense layer)/d(parameter) [D*D, n])] per layer (product rule over its factors);
        general_derivs[l] = [d*d, np] deriv_wrt_params of general leaf l."""
        out = []
        for g, fs in enumerate(self.gate_factors):
            qs = self.gate_params(g)
            dm = np.zeros((self.D * self.D, len(qs)))
            for c, q in enumerate(qs):
                acc = np.zeros((self.D, self.D))
                for k, (lk, _) in enumerate(fs):
                    if not self._leaf_holds(lk, q):
                        continue
                    X = np.eye(self.D)
                    for f, (l, tg) in enumerate(fs):
                        F = self._leaf_direction(l, q, general_derivs) if f == k else self._leaf(values, l)
                        X = self.embed(F, tg) @ X
                    acc += X
                dm[:, c] = acc.ravel()
            out.append((qs, dm))
        return out
