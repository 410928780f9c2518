"""Test-only numpy interpreter of libgstfwd walk programs (include/gstfwd.h, GST_OP_*).

Used by the CPU test-suite to check the plan compiler without a GPU: executes the programs with
the reference's arithmetic order (ascending j, separate multiply and add, start from 0.0), so the
result must equal the oracle bit for bit.
"""
import numpy as np

OP_END, OP_RHO, OP_APPLY, OP_SAVE, OP_LOAD, OP_EMIT, OP_NODE = 0, 1, 2, 3, 4, 5, 6


def matvec(M, v):
    acc = np.zeros(len(v))
    for j in range(len(v)):
        acc = acc + M[:, j] * v[j]
    return acc


def dot(e, v):
    acc = 0.0
    for j in range(len(v)):
        acc = acc + e[j] * v[j]
    return acc


def run_programs(words, task_off, gates, rhos, effects, eff_ptr, eff_label, eff_dest, n_elements):
    out = np.full(n_elements, np.nan)
    written = np.zeros(n_elements, np.int32)
    stats = dict(applies=0, max_slot=0)
    for t in range(len(task_off) - 1):
        pc = int(task_off[t])
        v = None
        slots = {}
        while True:
            w = int(words[pc]); pc += 1
            op, arg = w >> 28, w & 0x0FFFFFFF
            if op == OP_END:
                break
            if op == OP_RHO:
                v = rhos[arg].copy()
            elif op == OP_APPLY:
                v = matvec(gates[arg], v); stats['applies'] += 1
            elif op == OP_SAVE:
                slots[arg] = v.copy(); stats['max_slot'] = max(stats['max_slot'], arg + 1)
            elif op == OP_LOAD:
                v = slots[arg].copy()
            elif op == OP_EMIT:
                for x in range(eff_ptr[arg], eff_ptr[arg + 1]):
                    out[eff_dest[x]] = dot(effects[eff_label[x]], v)
                    written[eff_dest[x]] += 1
            elif op == OP_NODE:
                node_states = stats.setdefault('node_states', {})
                if arg in node_states:      # a replayed state must be bit-identical to its first computation
                    assert np.array_equal(node_states[arg].view(np.uint64), v.view(np.uint64))
                node_states[arg] = v.copy()
            else:
                raise AssertionError("bad opcode %d" % op)
        assert pc == task_off[t + 1]
    return out, written, stats


OP_CACHE = 7


def run_dirty_program(words, gates, rhos, effects, eff_ptr, eff_label, eff_dest, cache, out):
    """One "dirty program" (gst_get_dirty_programs): the part of a task's walk that a perturbation of one object changes.
    `gates` / `rhos` are the PERTURBED model's arrays, `cache` the base pass's states by id (OP_CACHE starts from one of
    them); probabilities of the emitted circuits go into `out`.  Returns the list of emitted circuits."""
    v, slots, emitted = None, {}, []
    for w in words:
        w = int(w)
        op, arg = w >> 28, w & 0x0FFFFFFF
        if op == OP_END:
            break
        if op == OP_RHO:
            v = rhos[arg].copy()
        elif op == OP_CACHE:
            v = cache[arg].copy()
        elif op == OP_APPLY:
            v = matvec(gates[arg], v)
        elif op == OP_SAVE:
            slots[arg] = v.copy()
        elif op == OP_LOAD:
            v = slots[arg].copy()
        elif op == OP_EMIT:
            emitted.append(arg)
            for x in range(eff_ptr[arg], eff_ptr[arg + 1]):
                out[eff_dest[x]] = dot(effects[eff_label[x]], v)
        else:
            raise AssertionError("opcode %d does not belong in a dirty program" % op)
    return emitted
