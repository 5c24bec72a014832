"""Summation-order model of the checker (oracle/linalg.hpp::assoc_mode, round 6): unit tests of the two Eigen 3.4 / SSE2 orders it restates
and of what switching it on does to whole solves (CPU only).  The GPU-side flip rates against the shipped library are
tests/test_cross_arithmetic.py::test_bench_batch_against_eigen_order_checker."""
import ctypes

import numpy as np
import pytest


def _redux(p):      # Redux.h, LinearVectorizedTraversal / NoUnrolling, Packet2d, aligned start 0
    n = len(p)
    a2 = (n // 4) * 4; a1 = (n // 2) * 2
    if a1 == 0:
        r = p[0] if n else 0.0
        for v in p[1:]:
            r = r + v
        return r
    l0, l1 = p[0], p[1]
    if a1 > 2:
        m0, m1 = p[2], p[3]
        for i in range(4, a2, 4):
            l0 = l0 + p[i]; l1 = l1 + p[i + 1]; m0 = m0 + p[i + 2]; m1 = m1 + p[i + 3]
        l0 = l0 + m0; l1 = l1 + m1
        if a1 > a2:
            l0 = l0 + p[a2]; l1 = l1 + p[a2 + 1]
    r = l0 + l1
    for v in p[a1:]:
        r = r + v
    return r


def _gemv_row(p):   # GeneralMatrixVector.h, RowMajor: one Packet2d accumulator from zero, predux, scalar tail
    n = len(p)
    c0 = c1 = np.float64(0.0)
    j = 0
    while j + 2 <= n:
        c0 = p[j] + c0; c1 = p[j + 1] + c1; j += 2
    cc = c0 + c1
    for v in p[j:]:
        cc = cc + v
    return cc


@pytest.fixture(scope="module")
def olib(api, oracle_built):
    lib = api.load_oracle()
    lib.cddp_oracle_sum_order.restype = ctypes.c_double
    lib.cddp_oracle_sum_order.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_double), ctypes.c_int]
    return lib


def test_sum_orders_against_python_emulation(olib):
    rng = np.random.default_rng(3)
    differs = {1: 0, 2: 0}
    for n in range(1, 17):
        for _ in range(40):
            p = (rng.standard_normal(n) * 10.0 ** rng.integers(-8, 8, n)).astype(np.float64)
            ptr = p.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
            serial = np.float64(0.0)
            for v in p:
                serial = serial + v
            assert olib.cddp_oracle_sum_order(0, ptr, n) == serial
            assert olib.cddp_oracle_sum_order(1, ptr, n) == _redux(list(p)), n
            assert olib.cddp_oracle_sum_order(2, ptr, n) == _gemv_row(list(p)), n
            if n <= 3:   # up to three terms every order is the serial one: plants with nx, nu, m <= 3 cannot be affected
                assert olib.cddp_oracle_sum_order(1, ptr, n) == serial and olib.cddp_oracle_sum_order(2, ptr, n) == serial
            else:
                differs[1] += olib.cddp_oracle_sum_order(1, ptr, n) != serial
                differs[2] += olib.cddp_oracle_sum_order(2, ptr, n) != serial
    assert differs[1] > 50 and differs[2] > 50      # the orders are not aliases of the serial one


@pytest.mark.parametrize("plant", ["pendulum", "unicycle", "cartpole", "cartpole_clddp"])
def test_whole_solves_under_the_eigen_order(api, oracle_built, plant):
    if plant == "pendulum":
        p = api.pendulum_problem(api.SOLVER_IPDDP, True); spread = [0.1, 0.1]
    elif plant == "unicycle":
        p = api.unicycle_problem(api.SOLVER_IPDDP, 100, True); spread = [0.05, 0.05, 0.05]
    else:
        p = api.cartpole_problem(api.SOLVER_CLDDP if plant.endswith("clddp") else api.SOLVER_IPDDP, True); spread = [0.1, 0.3, 0.1, 0.1]
    p.options.max_iterations = 30
    B = 8
    x0 = api.batch_x0(p, B, 20261110, spread)
    r0 = api.oracle_solve_batch(p, x0, n_threads=4, want_traj=False)[0]
    api.set_assoc_mode(1)
    try:
        r1 = api.oracle_solve_batch(p, x0, n_threads=4, want_traj=False)[0]
    finally:
        api.set_assoc_mode(0)
    assert np.array_equal(r0["iterations"], r1["iterations"]) and np.array_equal(r0["status"], r1["status"])
    rel = np.abs(r0["final_objective"] - r1["final_objective"]) / np.maximum(1.0, np.abs(r0["final_objective"]))
    assert rel.max() < 1e-8
    if p.nx <= 3:
        assert np.array_equal(r0["final_objective"], r1["final_objective"])       # no sum of four or more non-zero terms anywhere
    else:
        assert not np.array_equal(r0["final_objective"], r1["final_objective"])   # the order reaches the arithmetic (last bits move)
