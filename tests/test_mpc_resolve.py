"""MPC re-solves on a re-used handle (SURVEY.md 8(f1) caller side): K rounds of
"start from the state the previous plan predicted one step ahead, keep controls / duals / gains (warm start = existing solver
state, ipddp_solver.cpp:675-731), solve again" -- the loop of examples/ipddp_mpcc_rc.py:649-705 -- against the oracle object driven
the same way (pins of the branch: tests/cddp_core/test_ipddp_solver.cpp:1306-1380).  Every round must agree in iteration count,
status, sweep / rollout counts and objective; the K-th plan in its controls."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def rel(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b)))) if a.size else 0.0


@pytest.mark.parametrize("case", ["pendulum_ipddp_box", "cartpole_ipddp_box", "unicycle_ipddp_box_ball", "pendulum_clddp_box"])
def test_k_th_mpc_resolve_matches_the_oracle(api, oracle_built, case):
    import test_gpu_parity as T
    p = T.make(api, case)
    B, K = 4, 4
    x0 = api.batch_x0(p, B, 20261102, T.spread_for(p))
    U0 = api.batch_U0(p, B)
    hs = api.HipBatchSolver(p, B); hs.set_initial(x0, U0); hs.solve()
    r0 = hs.results()
    oracles = []
    for b in range(B):
        o = api.Oracle(p); o.set_initial(x0[b], None if U0 is None else U0[b]); q = o.solve(); o.set_warm_start(True)
        oracles.append(o)
        assert q["iterations"] == r0["iterations"][b] and q["status"] == r0["status"][b]
    hs.set_warm_start(True)      # (after the oracles were made: the flag lives in the shared problem object)
    for k in range(K):
        X, U = hs.trajectory()
        u0h, x1h = hs.plan_head()
        assert np.array_equal(u0h, U[:, 0, :]) and np.array_equal(x1h, X[:, 1, :])     # cddp_hip_get_plan_head == the full getter's rows
        hs.set_initial_state(x1h)
        hs.solve()
        r = hs.results(); Xn, Un = hs.trajectory()
        for b in range(B):
            o = oracles[b]
            Xo, _ = o.trajectory()
            assert rel(X[b, 1], Xo[1]) < 1e-9                      # both sides shift from (numerically) the same predicted state
            o.update_initial(X[b, 1]); q = o.solve()
            assert q["iterations"] == r["iterations"][b] and q["status"] == r["status"][b], (case, k, b, q, r[b])
            assert q["n_backward"] == r["n_backward"][b] and q["n_forward"] == r["n_forward"][b], (case, k, b)
            if np.isfinite(q["final_objective"]):
                assert rel(r["final_objective"][b], q["final_objective"]) < 1e-7, (case, k, b)
            if k == K - 1 and q["status"] in (api.STATUS_OPTIMAL, api.STATUS_ACCEPTABLE):
                _, Uo = o.trajectory()
                assert rel(Un[b], Uo) < 1e-5, (case, b)
    hs.close()


@pytest.mark.parametrize("case", ["pendulum_ipddp_box", "unicycle_ipddp_box_ball", "pendulum_clddp_box"])
def test_forgotten_solver_state_is_a_new_solver_object(api, oracle_built, case):
    """cddp_hip_forget_solver_state (round 6): the MPC caller that builds a fresh problem per step and seeds it with the shifted previous
    plan takes the reference's "warm start with provided trajectory" branch (ipddp_solver.cpp:733-816).  On ONE long-lived handle,
    forget + set_initial(shifted plan) + solve must give the bits a NEW handle gives for the same seed, and what a new oracle object gives."""
    import test_gpu_parity as T
    p = T.make(api, case)
    B = 6
    x0 = api.batch_x0(p, B, 20261103, T.spread_for(p))
    U0 = api.batch_U0(p, B)
    hs = api.HipBatchSolver(p, B); hs.set_initial(x0, U0); hs.solve()
    hs.set_warm_start(True)
    for k in range(3):
        X, U = hs.trajectory()
        Xs = np.ascontiguousarray(np.concatenate([X[:, 1:], X[:, -1:]], axis=1)); Us = np.ascontiguousarray(np.concatenate([U[:, 1:], U[:, -1:]], axis=1))
        x1 = np.ascontiguousarray(Xs[:, 0])
        hs.forget_solver_state(); hs.set_initial(x1, Us, Xs); hs.solve()
        r = hs.results(); Xa, Ua = hs.trajectory()
        fresh = api.HipBatchSolver(p, B); fresh.set_initial(x1, Us, Xs); fresh.solve()
        rf = fresh.results(); Xf, Uf = fresh.trajectory(); fresh.close()
        for name in ("iterations", "status", "n_backward", "n_forward", "final_objective"):
            assert np.array_equal(r[name], rf[name]), (k, name)
        assert np.array_equal(Xa, Xf) and np.array_equal(Ua, Uf)
        for b in range(B):
            o = api.Oracle(p); o.set_warm_start(True); o.set_initial(x1[b], Us[b], Xs[b]); q = o.solve()
            assert q["iterations"] == r["iterations"][b] and q["status"] == r["status"][b], (k, b, q["iterations"], r["iterations"][b])
            assert rel(q["final_objective"], r["final_objective"][b]) < 1e-7
    hs.close()
