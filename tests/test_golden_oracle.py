"""CPU: the oracle against the committed golden vectors (tests/golden, made by make_golden.py)."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, fnv1a64, middlebury2003

GOLD = json.load(open(os.path.join(GOLDEN, "golden.json")))


def test_teddy_crop_all_taps(orc):
    z = np.load(os.path.join(GOLDEN, "teddy_crop_64x48x16.npz"))
    disp, taps = orc.match(z["left"], z["right"], orc.default_option(max_disparity=16), want="all")
    assert np.array_equal(disp, z["disp_final"])
    for k, v in taps.items():
        assert np.array_equal(v, z[k]), k


@pytest.mark.parametrize("name", ["teddy", "cones"])
def test_middlebury_hashes(orc, name):
    L, R, gt = middlebury2003(name)
    disp, taps = orc.match(L, R, orc.default_option(max_disparity=64), want="all")
    g = GOLD[name]
    assert fnv1a64(disp) == g["hash"]["disp_final"]
    for k, v in taps.items():
        assert fnv1a64(v) == g["hash"][k], k
    z = np.load(os.path.join(GOLDEN, "%s_disp.npz" % name))
    assert np.array_equal(disp, z["disp_final"])
    bad, _ = orc.bad_fraction(disp, gt, 4.0, 2.0)
    assert abs(bad - g["bad2"]) < 1e-12


def test_synth_generator_frozen(synth):
    for key in ("synth_7", "synth_11", "synth_13"):
        g = GOLD[key]
        L, R, _ = synth.pair(int(key.split("_")[1]), g["W"], g["H"], g["dmax"] - g["dmin"])
        assert fnv1a64(L) + fnv1a64(R) == g["input_hash"]
