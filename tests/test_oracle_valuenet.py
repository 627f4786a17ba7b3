"""Pin oracle/valuenet_oracle.c against the reference's own Net (model/model_vv.py:13-52).

tests/golden/ref_valuenet.npz: weights of Net() under torch.manual_seed(0) (+ a scaled 'trained-like' set),
64 boards in the style of tools/test.py:23-28 and the reference's CPU fp32 outputs.  Tolerance: 1e-4 absolute
on the post-affine outputs (BASELINE.json north_star).
"""
import math
import os

import numpy as np

TOL = 1e-4


def test_valuenet_oracle_within_tolerance_of_reference(oracle, golden_dir):
    z = np.load(os.path.join(golden_dir, "ref_valuenet.npz"))
    L = oracle.lib()
    states = np.ascontiguousarray(z["states"].reshape(-1, 200))
    for pk, ok in (("params", "out"), ("params2", "out2")):
        p = np.ascontiguousarray(z[pk])
        v = np.zeros(len(states), np.float32)
        var = np.zeros(len(states), np.float32)
        L.orc_valuenet_forward(oracle.ptr(p), oracle.ptr(states), len(states), oracle.ptr(v), oracle.ptr(var))
        ref = z[ok]
        scale = np.maximum(1.0, np.abs(ref).max(0) / 1000.0)  # params2 widens the output range to 4000
        assert np.abs(v - ref[:, 0]).max() <= TOL * scale[0], np.abs(v - ref[:, 0]).max()
        assert np.abs(var - ref[:, 1]).max() <= TOL * max(1.0, p[478339] / 1000.0), np.abs(var - ref[:, 1]).max()


def test_orc_exp_accuracy(oracle):
    L = oracle.lib()
    xs = np.concatenate([np.linspace(-40, 40, 4001), np.random.default_rng(0).standard_normal(2000) * 5])
    for x in xs:
        e = L.orc_exp(float(x))
        assert abs(e / math.exp(x) - 1) < 4e-16
