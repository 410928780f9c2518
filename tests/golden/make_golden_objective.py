"""Generate tests/golden/objective_vectors.npz from the reference's raw objective functions (build container only).

    PYTHONPATH=<built copy of the reference> OMP_NUM_THREADS=1 python tests/golden/make_golden_objective.py

Inputs are synthetic (seeded): probabilities around and outside [0, 1], counts with exact zeros, probabilities below
min_prob_clip / radius and equal to the frequency, so every branch of the element-wise maps is exercised.  Outputs
are what RawChi2Function / RawPoissonPicDeltaLogLFunction (pygsti/objectivefns/objectivefns.py:1750-2110, 2829-3230)
return for terms, lsvec, dterms, dlsvec, and the objective-level row scale used by
TimeIndependentMDCObjectiveFunction.dlsvec (:4633-4665): 0.5/lsvec * dterms."""
import os
import numpy as np
from pygsti.objectivefns import objectivefns as O

rng = np.random.default_rng(20260927)
n = 4096
N = rng.choice([50.0, 100.0, 1000.0, 1e5], size=n)
p_true = rng.dirichlet(np.ones(4), size=n // 4).reshape(-1)
counts = rng.binomial(N.astype(np.int64), np.clip(p_true, 0, 1)).astype(np.float64)
counts[rng.random(n) < 0.15] = 0.0                                 # zero-frequency elements
probs = p_true + rng.normal(0, 0.01, n)
probs[:64] = np.linspace(-2e-4, 3e-4, 64)                          # straddle min_prob_clip = radius = 1e-4 (and 0)
probs[64:96] = counts[64:96] / N[64:96]                            # p == f exactly
probs[96:128] = 1.0 + np.linspace(-1e-3, 1e-3, 32)                 # around 1
counts[128:160] = N[128:160]                                       # f == 1
freqs = counts / N
out = dict(probs=probs, counts=counts, total_counts=N)
with np.errstate(all="ignore"):
    chi2 = O.RawChi2Function({"min_prob_clip_for_weighting": 1e-4})
    out["chi2_terms"] = chi2.terms(probs, counts, N, freqs)
    out["chi2_lsvec"] = chi2.lsvec(probs, counts, N, freqs)
    out["chi2_dterms"] = chi2.dterms(probs, counts, N, freqs)
    out["chi2_dlsvec"] = chi2.dlsvec(probs, counts, N, freqs)
    out["chi2_hterms"] = chi2.hterms(probs, counts, N, freqs)
    pl = O.RawPoissonPicDeltaLogLFunction({"min_prob_clip": 1e-4, "radius": 1e-4})
    out["logl_terms"] = pl.terms(probs, counts, N, freqs)
    out["logl_lsvec"] = pl.lsvec(probs, counts, N, freqs)
    out["logl_dterms"] = pl.dterms(probs, counts, N, freqs)
    out["logl_hterms"] = pl.hterms(probs, counts, N, freqs)
    ls = out["logl_lsvec"]
    p5 = 0.5 / ls
    p5[np.abs(ls) < 1e-100] = 0.0
    out["logl_rowscale"] = p5 * out["logl_dterms"]                 # objectivefns.py:4655-4660
    ls = out["chi2_lsvec"]
    p5 = 0.5 / ls
    p5[np.abs(ls) < 1e-100] = 0.0
    out["chi2_rowscale"] = p5 * out["chi2_dterms"]
out["min_prob_clip"] = 1e-4
out["radius"] = 1e-4
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "objective_vectors.npz"), **out)
for k, v in out.items():
    if isinstance(v, np.ndarray):
        print(k, v.shape, float(np.nanmin(v)), float(np.nanmax(v)), int(np.isnan(v).sum()))
