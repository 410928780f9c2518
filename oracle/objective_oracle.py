"""CPU restatement (numpy) of the element-wise objective maps that turn probabilities into the least-squares vector
and the row scale of its Jacobian -- TEST INFRASTRUCTURE, never imported by the product.

Follows pygsti/objectivefns/objectivefns.py of the reference:
  chi2   : RawChi2Function.lsvec :1814-1848, .dlsvec :1850-1885, _weights :2040-2060, _dweights :2062-2084;
           terms = lsvec**2 :559-590, dterms = 2*lsvec*dlsvec :631-669
  dlogl  : RawPoissonPicDeltaLogLFunction ('minp' regularisation, "harsh" zero-frequency radius):
           _intermediates :2944-2978, terms :2980-3053, lsvec :3055-3096, dterms :3098-3160,
           _zero_freq_terms_harsh / _zero_freq_dterms_harsh :3185-3195
  row scale of the objective-level Jacobian (what multiplies each row of dprobs):
           TimeIndependentMDCObjectiveFunction.dterms :4595-4631 and .dlsvec :4633-4665  ->  (0.5 / lsvec) * dterms,
           0 where |lsvec| < 1e-100; lsvec carries the sign of the raw lsvec (:4573-4593).
Pinned against tests/golden/objective_vectors.npz (generated from the reference by tests/golden/make_golden_objective.py).
"""
import numpy as np

CHI2, DLOGL = 0, 1


def objective_rows(kind, probs, counts, total_counts, min_prob_clip=1e-4, radius=1e-4):
    """-> (terms, lsvec, dterms, rowscale), each shaped like probs."""
    p = np.asarray(probs, np.float64)
    c = np.asarray(counts, np.float64)
    N = np.asarray(total_counts, np.float64)
    f = c / N
    with np.errstate(all="ignore"):
        if kind == CHI2:
            cp = np.maximum(p, min_prob_clip)
            w = np.sqrt(N / cp)
            ls = (p - f) * w
            dw = np.where(p < min_prob_clip, 0.0, -0.5 * w / cp)
            dls = w + (p - f) * dw
            terms = ls * ls
            dterms = 2 * ls * dls
        elif kind == DLOGL:
            fnz = np.where(c == 0, 1.0, f)
            pos = np.where(p < min_prob_clip, min_prob_clip, p)
            c0 = N - c / min_prob_clip
            c1 = 0.5 * c / (min_prob_clip * min_prob_clip)
            t = c * (np.log(fnz) - 1.0) - c * np.log(pos) + N * pos
            t = np.maximum(t, 0.0)
            dpm = p - min_prob_clip
            t = np.where(p < min_prob_clip, t + c0 * dpm + c1 * (dpm * dpm), t)
            a = radius
            zf = N * np.where(p >= a, p, (-1.0 / (3 * a * a)) * (p * p * p) + (p * p) / a + a / 3.0)
            terms = np.where(c == 0, zf, t)
            ls = np.sqrt(terms)
            d = np.where(p < min_prob_clip, c0 + 2 * c1 * dpm, N - c / pos)
            dzf = N * np.where(p >= a, 1.0, (-1.0 / (a * a)) * (p * p) + 2 * p / a)
            dterms = np.where(c == 0, dzf, d)
        else:
            raise ValueError("unknown objective kind")
        p5 = np.where(np.abs(ls) < 1e-100, 0.0, 0.5 / ls)
        rowscale = p5 * dterms
    return terms, ls, dterms, rowscale


def objective_coeffs(kind, probs, counts, total_counts, min_prob_clip=1e-4, radius=1e-4):
    """-> (dterms, hterms): first and second derivative of every term in its probability -- the coefficients of
    _hessian_from_block (objectivefns.py:4914-4968).  chi2: base-class dterms / hterms over lsvec, dlsvec, hlsvec
    (:631-669, 755-794, 1886-1921, 2086-2108); dlogl: RawPoissonPicDeltaLogLFunction.dterms / .hterms (:3098-3183)."""
    p = np.asarray(probs, np.float64)
    c = np.asarray(counts, np.float64)
    N = np.asarray(total_counts, np.float64)
    f = c / N
    with np.errstate(all="ignore"):
        if kind == CHI2:
            cp = np.maximum(p, min_prob_clip)
            w = np.sqrt(N / cp)
            clipped = p < min_prob_clip
            dw = np.where(clipped, 0.0, -0.5 * w / cp)
            hw = np.where(clipped, 0.0, 0.75 * w / cp**2)
            ls = (p - f) * w
            dls = w + (p - f) * dw
            hls = 2 * dw + (p - f) * hw
            return 2 * ls * dls, 2 * (dls**2 + ls * hls)
        if kind == DLOGL:
            _, _, dterms, _ = objective_rows(kind, p, c, N, min_prob_clip, radius)
            pos = np.where(p < min_prob_clip, min_prob_clip, p)
            c1 = 0.5 * c / (min_prob_clip * min_prob_clip)
            h = np.where(p < min_prob_clip, 2 * c1, c / pos**2)
            a = radius
            hzf = np.where(p >= a, 0.0, N * ((-2.0 / a**2) * p + 2.0 / a))
            return dterms, np.where(c == 0, hzf, h)
    raise ValueError("unknown objective kind")

