"""development aid: latency of the base (probabilities) pass on synthetic plans -- isolates the per-step cost of an
uninterrupted chain from the cost of branches/EMITs.   python tools/base_latency.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygsti_amd import _lib

def run(name, circuits, D=16, nG=6, nEff=4, cache=True):
    ptr = np.zeros(len(circuits) + 1, np.int64); ptr[1:] = np.cumsum([len(c) for c in circuits])
    gates = np.concatenate(circuits).astype(np.int32)
    nC = len(circuits)
    eff_ptr = np.arange(nC + 1, dtype=np.int64) * nEff
    eff_label = np.tile(np.arange(nEff, dtype=np.int32), nC)
    eff_dest = np.arange(nC * nEff, dtype=np.int32)
    plan = _lib.Plan.from_circuits(D, nG, 1, nEff, nC * nEff, np.zeros(nC, np.int32), ptr, gates, eff_ptr, eff_label, eff_dest)
    rng = np.random.default_rng(0)
    G = rng.standard_normal((nG, D, D)) * 0.2
    plan.set_model(G, rng.standard_normal((1, D)), rng.standard_normal((nEff, D)))
    plan.set_param_map(np.zeros(1, np.int32), np.zeros(1, np.int32), np.zeros(1, np.int32))
    d_p = plan.device_malloc(nC * nEff * 8)
    d_J = plan.device_malloc(nC * nEff * 8)
    for _ in range(3):
        plan.fill_probs_dev(d_p)
    plan.sync()
    n = 20
    t0 = time.perf_counter()
    for _ in range(n):
        plan.fill_probs_dev(d_p)
    plan.sync()
    t = (time.perf_counter() - t0) / n
    st = plan.stats()
    prog, off = plan.program()
    ap = np.add.reduceat(((prog >> 28) == 2).astype(np.int64), off[:-1])
    print("%-40s tasks %5d applies/task max %6d  %8.1f us/pass  -> %6.1f ns per apply of the longest task" % (
        name, st["n_tasks"], ap.max(), t * 1e6, t * 1e9 / ap.max()))

rng = np.random.default_rng(1)
run("64 random chains depth 4000", [rng.integers(0, 6, 4000) for _ in range(64)])
run("1024 random chains depth 4000", [rng.integers(0, 6, 4000) for _ in range(1024)])
run("64 random chains depth 200", [rng.integers(0, 6, 200) for _ in range(64)])
# germ-power family: chain of depth 1024 with 11 branch points x 11 short tails
fam = []
base = rng.integers(0, 6, 1024)
for p in [1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024]:
    for t in range(11):
        fam.append(np.concatenate([base[:p], rng.integers(0, 6, 3)]))
run("1 germ-power family (121 circuits)", fam)
fams = []
for f in range(173):
    base = rng.integers(0, 6, 1024)
    for p in [1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024]:
        for t in range(11):
            fams.append(np.concatenate([base[:p], rng.integers(0, 6, 3)]))
run("173 germ-power families", fams)
