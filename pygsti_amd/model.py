"""Host-side dense explicit model: the 6-method interface the forward simulator consumes.

Mirror of the slice of pyGSTi's `ExplicitOpModel` that sits below the hot path:
dense `full`-parameterised members (modelmembers/operations/fullarbitraryop.py:98-164: one
parameter per dense element, `deriv_wrt_params` = identity), `to_vector` / `from_vector`
(models/model.py:1151-1196), `num_params`, `dim`, `depolarize` (models/explicitmodel.py:1101-1150;
observed behaviour: operations and state preparations are scaled, POVM effects are left as they
are), `kick`.  Parameter order follows the reference: preps, then POVM effects, then operations
(smq2Q_XYICNOT: rho 0:16, Mdefault 16:80, gates from 80 in 256-slices).
"""
from collections import OrderedDict

import numpy as np

KIND_GATE, KIND_RHO, KIND_EFFECT = 0, 1, 2


class ExplicitDenseModel:
    def __init__(self, operations, preps, povms, sim=None):
        self.operations = OrderedDict((k, np.array(v, dtype=np.float64)) for k, v in operations.items())
        self.preps = OrderedDict((k, np.array(v, dtype=np.float64).ravel()) for k, v in preps.items())
        self.povms = OrderedDict((k, OrderedDict((o, np.array(e, dtype=np.float64).ravel()) for o, e in p.items()))
                                 for k, p in povms.items())
        self.dim = next(iter(self.preps.values())).size
        for g in self.operations.values():
            assert g.shape == (self.dim, self.dim)
        self._sim = None
        if sim is not None:
            self.sim = sim

    # -- simulator attachment (models/model.py:486-501 keeps sim.model <-> model.sim in sync) --------
    @property
    def sim(self):
        if self._sim is None:
            from .forwardsim import HipMapForwardSimulator
            self.sim = HipMapForwardSimulator()
        return self._sim

    @sim.setter
    def sim(self, simulator):
        self._sim = simulator
        if simulator is not None:
            simulator.model = self

    # -- parameters -------------------------------------------------------------------------------
    def _members(self):
        """(kind, label, array) in parameter order."""
        for k, v in self.preps.items():
            yield KIND_RHO, k, v
        for pk, povm in self.povms.items():
            for ok, e in povm.items():
                yield KIND_EFFECT, pk + "_" + ok, e
        for k, g in self.operations.items():
            yield KIND_GATE, k, g

    @property
    def num_params(self):
        return sum(a.size for _, _, a in self._members())

    def to_vector(self):
        return np.concatenate([a.ravel() for _, _, a in self._members()])

    def from_vector(self, v, close=False):
        v = np.asarray(v, dtype=np.float64)
        assert v.size == self.num_params
        off = 0
        for _, _, a in self._members():
            a.ravel()[...] = v[off:off + a.size]   # members own C-contiguous buffers: ravel() is a view
            off += a.size

    def gpindices(self, kind, label):
        off = 0
        for k, l, a in self._members():
            if k == kind and l == label:
                return slice(off, off + a.size)
            off += a.size
        raise KeyError(label)

    # -- labels ---------------------------------------------------------------------------------------
    @property
    def effect_labels(self):
        return [pk + "_" + ok for pk, povm in self.povms.items() for ok in povm]

    def effect_vector(self, full_label):
        pk, ok = full_label.split("_", 1)
        return self.povms[pk][ok]

    # -- noise helpers ----------------------------------------------------------------------------------
    def copy(self):
        m = ExplicitDenseModel(self.operations, self.preps, self.povms)
        if self._sim is not None:
            m.sim = self._sim.copy()
        return m

    def depolarize(self, op_noise=None, spam_noise=None):
        m = self.copy()
        D = self.dim
        if op_noise is not None:
            s = np.array([1.0] + [1.0 - op_noise] * (D - 1))
            for k in m.operations:
                m.operations[k] = m.operations[k] * s[:, None]      # diag(s) @ G
        if spam_noise is not None:
            s = np.array([1.0] + [1.0 - spam_noise] * (D - 1))
            for k in m.preps:
                m.preps[k] = m.preps[k] * s
        return m

    def kick(self, absmag=1.0, seed=None):
        """Random additive perturbation of every operation element (not the reference's RNG stream)."""
        m = self.copy()
        rng = np.random.default_rng(seed)
        for k in m.operations:
            m.operations[k] = m.operations[k] + absmag * (2 * rng.random(m.operations[k].shape) - 1)
        return m
