"""include/colate_coal_EM.hpp: the C++ class with the reference's `coal_EM` interface, used the way
include/coal/coal.cpp:3698-3721 uses it (csrc/tools/coal_EM_shim_check.cpp), against the oracle's per-bin outputs."""
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cxx_coal_EM_shim_matches_oracle():
    ep, _ = ol.epochs_from_bins("3,7,0.2")
    rng = np.random.default_rng(4)
    rates = np.exp(rng.uniform(np.log(1e-6), np.log(1e-3), ep.size))
    grid = ol.age_grid()
    ages = grid[[1, 30, 41, 64, 65, 90, 120, 150, 170, 184]]
    text = " ".join(["%d" % ep.size] + ["%.17g" % x for x in ep] + ["%.17g" % x for x in rates] + ["%d" % ages.size]
                    + ["%.17g" % a for a in ages])
    exe = os.path.join(ROOT, "colate_amd", "bin", "coal_EM_shim_check")
    r = subprocess.run([exe], input=text, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    rows = [[float.fromhex(t) for t in line.split()] for line in r.stdout.strip().split("\n")]
    assert len(rows) == 2 * ages.size
    i = 0
    for a in ages:
        for kind in (0, 1):
            ll0, n0, d0 = ol.em_call(kind, ep, rates, a)
            row = rows[i]
            i += 1
            num, den = np.array(row[1::2]), np.array(row[2::2])
            assert abs(row[0] - ll0) <= 1e-12 * max(1.0, abs(ll0))
            assert (np.abs(num - n0) <= 1e-8 * np.abs(n0) + 1e-300).all()
            dt = np.append(np.diff(ep), 0.0)
            assert (np.abs(den - d0) <= 1e-6 * np.abs(d0) + 1e-13 * dt + 1e-300).all()
