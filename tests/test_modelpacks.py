"""The build's own model packs / GST circuit generator against pins captured from the reference
(tests/golden/designs.npz): circuit COUNT, total depth and sha256 of the integerised list in the
reference's own circuit order, for every max length 1..1024, lite and full germ sets."""
import hashlib

import numpy as np
import pytest

from conftest import load_fixture
from pygsti_amd import modelpacks as MP


def _hash(ptr, g):
    h = hashlib.sha256(); h.update(np.asarray(ptr, np.int64).tobytes()); h.update(np.asarray(g, np.int32).tobytes())
    return h.hexdigest()


@pytest.mark.parametrize("tag,pack", [("1Q", MP.smq1Q_XYI), ("2Q", MP.smq2Q_XYICNOT)])
def test_target_models_match_reference(tag, pack):
    d = load_fixture("designs")
    assert list(d[tag + "_op_labels"]) == list(pack.gate_labels)
    tm = pack.target_model()
    G = np.array([tm.operations[l] for l in pack.gate_labels])
    assert np.abs(G - d[tag + "_gates"]).max() < 4e-16        # the reference carries ~1e-17 conversion noise
    assert np.abs(tm.preps["rho0"] - d[tag + "_rho"].ravel()).max() < 4e-16
    E = np.array(list(tm.povms["Mdefault"].values()))
    assert np.abs(E - d[tag + "_effects"].reshape(E.shape)).max() < 4e-16
    assert list(tm.povms["Mdefault"].keys()) == list(d[tag + "_effect_labels"])
    dm = tm.depolarize(op_noise=0.01, spam_noise=0.01)
    assert np.abs(dm.to_vector() - d[tag + "_paramvec_depol"]).max() < 4e-16   # same parameter ORDER, too
    v = dm.to_vector(); dm2 = tm.copy(); dm2.from_vector(v)
    assert np.array_equal(dm2.to_vector(), v)


@pytest.mark.parametrize("tag,pack", [("1Q", MP.smq1Q_XYI), ("2Q", MP.smq2Q_XYICNOT)])
@pytest.mark.parametrize("lite", [True, False])
def test_experiment_designs_match_reference(tag, pack, lite):
    d = load_fixture("designs")
    key = "%s_%s" % (tag, "lite" if lite else "full")
    lookup = {l: i for i, l in enumerate(pack.gate_labels)}
    checked = 0
    for L, cnt, dep, sha in zip(d[key + "_L"], d[key + "_counts"], d[key + "_depth"], d[key + "_sha256"]):
        if cnt < 0 or (tag == "2Q" and not lite and L == 1024):
            continue          # the 136,275-circuit design is checked (count + depth) in the slower test below
        cl = pack.create_gst_circuits(int(L), lite=lite)
        ptr = np.zeros(len(cl) + 1, np.int64); ptr[1:] = np.cumsum([len(c) for c in cl])
        g = np.fromiter((lookup[x] for c in cl for x in c), np.int32, count=int(ptr[-1]))
        assert (len(cl), len(g), _hash(ptr, g)) == (cnt, dep, str(sha)), (key, L)
        checked += 1
    assert checked >= 2


def test_full_2q_design_count_depth_hash():
    d = load_fixture("designs")
    pack = MP.smq2Q_XYICNOT
    cl = pack.create_gst_circuits(1024, lite=False)
    lookup = {l: i for i, l in enumerate(pack.gate_labels)}
    ptr = np.zeros(len(cl) + 1, np.int64); ptr[1:] = np.cumsum([len(c) for c in cl])
    g = np.fromiter((lookup[x] for c in cl for x in c), np.int32, count=int(ptr[-1]))
    i = list(d["2Q_full_L"]).index(1024)
    assert len(cl) == 136275 == d["2Q_full_counts"][i]
    assert len(g) == 31903477 == d["2Q_full_depth"][i]
    assert _hash(ptr, g) == str(d["2Q_full_sha256"][i])
