"""BASELINE.json's full-size configurations through size-independent properties: a trajectory's solve does not
depend on which batch it is part of (trajectories are independent problems: the solve of trajectory b inside the
full batch is, bit for bit, its solve inside a small batch), a sample of the full batch agrees with the oracle, and
the result record gather covers every trajectory."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _solve(api, p, x0, U0):
    hs = api.HipBatchSolver(p, x0.shape[0])
    hs.set_initial(x0, U0)
    hs.solve()
    r = hs.results(); X, U = hs.trajectory(); K, k = hs.gains()
    hs.close()
    return r, X, U, K, k


@pytest.mark.parametrize("config", ["C2_cartpole_4096", "C3_unicycle_8192", "C5_manip7_share"])
def test_full_batch_equals_sub_batches_and_oracle(api, config, oracle_built):
    if config == "C2_cartpole_4096":
        p, B, spread, n_oracle = api.cartpole_problem(api.SOLVER_IPDDP, True), 4096, np.array([0.3, 0.3, 0.1, 0.1]), 3
    elif config == "C3_unicycle_8192":
        p, B, spread, n_oracle = api.unicycle_problem(api.SOLVER_IPDDP, 200, True), 8192, 0.05 * np.ones(3), 2
    else:   # the per-GPU share of config 5 is 4096; 1024 keeps the test in seconds and still spans many wavefront tiles
        p, B, spread, n_oracle = api.manipulator7_problem(api.SOLVER_IPDDP, 150, terminal_equality=True, n_alphas=16), 1024, 0.02 * np.ones(14), 0
    x0 = api.batch_x0(p, B, 20260928 + 1, spread)
    U0 = api.batch_U0(p, B)
    r, X, U, K, k = _solve(api, p, x0, U0)
    assert np.all(np.isfinite(r["final_objective"])) and r["iterations"].min() >= 1
    # a scattered sample re-solved as its own small batch (other tile position, other ladder statistics)
    idx = np.array([0, 1, 63, 64, 65, B // 2 - 1, B // 2, B - 65, B - 2, B - 1] + list(range(100, 100 + 54)))
    rs, Xs, Us, Ks, ks = _solve(api, p, np.ascontiguousarray(x0[idx]), None if U0 is None else np.ascontiguousarray(U0[idx]))
    for key in ("final_objective", "iterations", "status"):
        assert np.array_equal(r[key][idx], rs[key]), key
    assert np.array_equal(X[idx], Xs) and np.array_equal(U[idx], Us) and np.array_equal(K[idx], Ks) and np.array_equal(k[idx], ks)
    # a few trajectories against the oracle (counts and status identical; values to the solve-level tolerance)
    for b in idx[:n_oracle]:
        o = api.Oracle(p); o.set_initial(x0[b], None if U0 is None else U0[b]); ro = o.solve()
        assert ro["iterations"] == r["iterations"][b] and ro["status"] == r["status"][b]
        if ro["status"] in (api.STATUS_OPTIMAL, api.STATUS_ACCEPTABLE):
            assert abs(ro["final_objective"] - r["final_objective"][b]) <= 1e-6 * max(1.0, abs(ro["final_objective"]))
