"""oracle/prefix_table.py -- TEST INFRASTRUCTURE (CPU baseline / checker input), NOT PRODUCT CODE.

Restates how the reference turns a circuit list into the prefix table its Cython loop walks:
PrefixTable.__init__ / _cache_hits / _build_table (pygsti/layouts/prefixtable.py:26-101, 680-741)
with max_cache_size=None -- circuits are evaluated in order of length; a circuit starts from the
longest OTHER circuit that is a prefix of it (if any); a circuit's final state is cached iff some
later circuit starts from it.  The reference finds prefixes by scanning the cache backwards
(O(rows x cache) tuple compares, 16 s for 2Q L<=1024); here the same relation is found from the
lexicographic order (a circuit's prefixes are exactly the open ancestors on a sorted sweep).

Pinned by tests/test_oracle.py: for every golden fixture the table built here has the same number
of gate applications and the same cache size as the reference's own table, and walking it with the
oracle gives bit-identical probabilities.
"""
import numpy as np


def build_table(circ_ptr, circ_gates, n_outcomes, circ_rho=None):
    """circuits (CSR of gate indices) -> dict in the fixture / oracle table format.
    Elements are laid out circuit-major: element k = (circuit k // n_outcomes, outcome k % n_outcomes)."""
    circ_ptr = np.asarray(circ_ptr, np.int64)
    circ_gates = np.asarray(circ_gates, np.int32)
    nC = len(circ_ptr) - 1
    lens = np.diff(circ_ptr)
    rho = np.zeros(nC, np.int32) if circ_rho is None else np.asarray(circ_rho, np.int32)
    circs = [circ_gates[circ_ptr[i]:circ_ptr[i + 1]] for i in range(nC)]
    keys = [(int(rho[i]),) + tuple(int(g) for g in circs[i]) for i in range(nC)]
    order = sorted(range(nC), key=lambda i: keys[i])
    # parent[i] = longest other circuit that is a (non-strict, for duplicates) prefix of circuit i
    parent = -np.ones(nC, np.int64)
    stack = []   # indices of circuits that are prefixes of the current one, shortest first
    for i in order:
        ki = keys[i]
        while stack:
            kt = keys[stack[-1]]
            if len(kt) <= len(ki) and ki[:len(kt)] == kt:
                break
            stack.pop()
        if stack:    # the reference's circuits include the prep label, so even 'rho only' (Lc = 1 > 0) is a cache hit
            parent[i] = stack[-1]
        stack.append(i)
    has_child = np.zeros(nC, bool)
    has_child[parent[parent >= 0]] = True
    # evaluation order: by length (stable), as sorted(..., key=len) in PrefixTable.__init__
    eval_order = sorted(range(nC), key=lambda i: lens[i])
    t_dest = np.empty(nC, np.int32); t_start = np.empty(nC, np.int32)
    t_cache = np.empty(nC, np.int32); t_rho = -np.ones(nC, np.int32)
    row_ptr = np.zeros(nC + 1, np.int64)
    cache_slot = -np.ones(nC, np.int64)
    gidx = []
    n_cache = 0
    for k, i in enumerate(eval_order):
        t_dest[k] = i
        p = parent[i]
        if p >= 0:
            assert cache_slot[p] >= 0, "parent must have been evaluated (it is shorter or equal and sorted first)"
            t_start[k] = cache_slot[p]
            gidx.append(circs[i][lens[p]:])
        else:
            t_start[k] = -1
            t_rho[k] = rho[i]
            gidx.append(circs[i])
        if has_child[i]:
            cache_slot[i] = n_cache; t_cache[k] = n_cache; n_cache += 1
        else:
            t_cache[k] = -1
        row_ptr[k + 1] = row_ptr[k] + len(gidx[-1])
    gate_idx = np.concatenate(gidx).astype(np.int32) if gidx else np.zeros(0, np.int32)
    nO = int(n_outcomes)
    return dict(n_rows=nC, cache_size=n_cache, nE=nC * nO, t_dest=t_dest, t_start=t_start, t_cache=t_cache,
                t_rho=t_rho, row_ptr=row_ptr, gate_idx=gate_idx,
                eff_ptr=np.arange(nC + 1, dtype=np.int64) * nO,
                eff_label=np.tile(np.arange(nO, dtype=np.int32), nC),
                eff_dest=np.arange(nC * nO, dtype=np.int32))
