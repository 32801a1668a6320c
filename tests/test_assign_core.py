"""CPU: the assignment solver the device kernel is built from (memotr_amd/csrc/assign_core.h, compiled for the host
by this test with g++) against scipy.optimize.linear_sum_assignment -- what the reference's matcher calls
(models/matcher.py:122-124) -- on random cost matrices with and without ties: identical pairs, not just equal cost."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
from scipy.optimize import linear_sum_assignment

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def host_lib(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("assign") / "libassign_host.so")
    src = os.path.join(ROOT, "tests", "native", "assign_host.cpp")
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-shared", "-fPIC", src, "-o", out], check=True)
    lib = ctypes.CDLL(out)
    lib.assign_host.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                ctypes.c_void_p]
    lib.assign_host.restype = ctypes.c_int
    return lib


def solve(lib, cost, lanes):
    cost = np.ascontiguousarray(cost, dtype=np.float32)
    n = min(cost.shape)
    r, c = np.full(n, -7, np.int32), np.full(n, -7, np.int32)
    rc = lib.assign_host(cost.ctypes.data, cost.shape[0], cost.shape[1], lanes, r.ctypes.data, c.ctypes.data)
    return rc, r, c


def cases(n=1000, seed=0):
    """Matcher-shaped problems (hundreds of queries x a few ground truths, and the transpose), square ones, and
    matrices full of ties: small integer costs, duplicated rows / columns, constants."""
    rng = np.random.default_rng(seed)
    for k in range(n):
        kind = k % 8
        if kind in (0, 1):
            shape = (int(rng.integers(40, 340)), int(rng.integers(1, 21)))
        elif kind == 2:
            shape = (int(rng.integers(1, 21)), int(rng.integers(40, 340)))
        elif kind == 3:
            m = int(rng.integers(1, 40))
            shape = (m, m)
        else:
            shape = (int(rng.integers(1, 60)), int(rng.integers(1, 60)))
        if kind <= 3:
            c = rng.standard_normal(shape).astype(np.float32) * 3
        elif kind == 4:
            c = rng.integers(0, 3, shape).astype(np.float32)                 # heavy ties
        elif kind == 5:
            c = rng.integers(-2, 2, shape).astype(np.float32)
            c[:, ::2] = c[:, :1]                                               # duplicated columns
        elif kind == 6:
            c = np.full(shape, float(rng.integers(-3, 3)), np.float32)        # constant: scipy returns the identity
        else:
            c = np.round(rng.standard_normal(shape) * 2).astype(np.float32) / 2
            c[rng.integers(0, shape[0])] = c[0]                                # duplicated rows
        yield c


@pytest.mark.parametrize("lanes", [1, 7, 64])
def test_assignments_are_identical_to_scipy_on_1000_random_matrices(host_lib, lanes):
    n_ties = 0
    for c in cases():
        want_r, want_c = linear_sum_assignment(c)
        rc, r, col = solve(host_lib, c, lanes)
        assert rc == min(c.shape)
        assert np.array_equal(r, want_r) and np.array_equal(col, want_c), (c.shape, r, want_r, col, want_c)
        n_ties += len(np.unique(c)) < c.size
    assert n_ties > 300


def test_degenerate_shapes_and_infeasible_costs(host_lib):
    rc, r, c = solve(host_lib, np.zeros((0, 5), np.float32), 64)
    assert rc == 0
    one = np.array([[3.0, 1.0, 2.0]], np.float32)
    rc, r, c = solve(host_lib, one, 64)
    assert rc == 1 and r.tolist() == [0] and c.tolist() == [1]
    rc, r, c = solve(host_lib, one.T.copy(), 64)
    assert rc == 1 and r.tolist() == [1] and c.tolist() == [0]
    bad = np.array([[np.inf, np.inf], [1.0, 2.0]], np.float32)                 # scipy: "cost matrix is infeasible"
    with pytest.raises(ValueError):
        linear_sum_assignment(bad)
    assert solve(host_lib, bad, 64)[0] == -1
