"""Development aid: what does a (task, 64-column wavefront) pair of the FD walk cost?  Runs the 1/8 atom of the bench design one
64-column block at a time (so that every pair has a SIMD of its own), traces the pairs (GST_FD_TRACE) and regresses
their durations on features read off the walk programs: perturbed (dirty) applications, dirty / clean EMITs, words,
SAVE / LOAD.  The fitted weights are what gst::task_gate_costs should estimate (per-SIMD queues of small atoms)."""
import os, sys, json, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pygsti_amd import _lib

OP_END, OP_RHO, OP_APPLY, OP_SAVE, OP_LOAD, OP_EMIT, OP_NODE = 0, 1, 2, 3, 4, 5, 6
ranks = int(os.environ.get("RANKS", "8"))
pack, model, circuits, layout = bench.build_workload("full", 1024, ranks, 0, 0, 0, 0, "strong")
plan = layout.atoms[0].plan()
plan.set_model(*layout.model_arrays(model)); plan.set_param_map(*layout.param_map(model))
nE, nP = layout.num_elements, model.num_params
words, off = plan.program()
nT = len(off) - 1
nG = 6


def features(g):
    """per task: [dirty applies, dirty emits, clean emits, words, dirty saves + loads] for a wavefront that perturbs gate g
    (g = -1: the preparation: dirty from the start)"""
    F = np.zeros((nT, 5))
    for t in range(nT):
        dirty = False; slot = {}
        for w in words[off[t]:off[t + 1]]:
            op, arg = int(w) >> 28, int(w) & 0x0FFFFFFF
            F[t, 3] += 1
            if op == OP_APPLY:
                if arg == g or g < 0: dirty = True
                if dirty: F[t, 0] += 1
            elif op == OP_RHO: dirty = g < 0
            elif op == OP_SAVE:
                slot[arg] = dirty
                if dirty: F[t, 4] += 1
            elif op == OP_LOAD:
                dirty = slot.get(arg, False)
                if dirty: F[t, 4] += 1
            elif op == OP_EMIT:
                F[t, 1 if dirty else 2] += 1
    return F


pk, po, pe = layout.param_map(model)
d_J = plan.device_malloc(nE * 64 * 8); d_p = plan.device_malloc(nE * 8)
X, y, tag = [], [], []
tmp = os.path.join(tempfile.gettempdir(), "calib_trace.bin")
for g in [-1] + list(range(nG)):
    if g < 0:
        cols = np.where(pk == 1)[0][:16]                    # the preparation's 16 parameters: one wavefront
    else:
        cols = np.where((pk == 0) & (po == g))[0][:64]      # 64 of the gate's 256 parameters: one wavefront
    os.environ["GST_FD_TRACE"] = tmp
    plan.fill_dprobs_dev(d_J, 64, cols.astype(np.int64), None, 1e-7, d_p, _lib.DERIV_FD); plan.sync()
    plan.fill_dprobs_dev(d_J, 64, cols.astype(np.int64), None, 1e-7, d_p, _lib.DERIV_FD); plan.sync()
    del os.environ["GST_FD_TRACE"]
    raw = np.fromfile(tmp, dtype=np.uint64)
    n = int(raw[0]); rec = raw[1:1 + 4 * n].reshape(n, 4)
    rec = rec[(rec[:, 0] & np.uint64(0x40000000)) == 0]
    pair = (rec[:, 0] & np.uint64(0x3fffffff)).astype(np.int64)
    dur = (rec[:, 2].astype(np.int64) - rec[:, 1].astype(np.int64)) / 100.0
    F = features(g)
    form = plan.stats()["last_fd_form"]
    n_units = max(1, len(pair) // nT)
    for p_, d_ in zip(pair, dur):
        t = int(p_) // n_units
        if t < nT:
            X.append(F[t]); y.append(d_); tag.append(g)
    print("gate %2d: %d pairs traced, form %d, duration us 50/100: %.1f %.1f" % (g, len(pair), form, np.median(dur), dur.max()), flush=True)
X = np.array(X); y = np.array(y); tag = np.array(tag)
A = np.column_stack([X, np.ones(len(X))])
coef, res, *_ = np.linalg.lstsq(A, y, rcond=None)
pred = A @ coef
print("fit: us per dirty apply %.4f, dirty emit %.4f, clean emit %.4f, word %.5f, dirty save/load %.4f, const %.2f" % tuple(coef))
print("relative to a dirty apply: dirty emit %.3f, clean emit %.3f, word %.4f, save/load %.3f, const %.1f applies" % tuple(coef[1:] / coef[0]))
print("residual std / mean: %.3f ; corr(pred, y) %.4f" % (np.std(y - pred) / y.mean(), np.corrcoef(pred, y)[0, 1]))
old = X[:, 0] + (X[:, 1] + X[:, 2]) / 4.0
k = np.polyfit(old, y, 1)
print("current estimate (applies + emits / 4): corr %.4f, residual std / mean %.3f" % (np.corrcoef(old, y)[0, 1], np.std(y - np.polyval(k, old)) / y.mean()))
np.savez(os.path.join("gpurun_out", "calib_pairs.npz"), X=X, y=y, tag=tag, coef=coef)
