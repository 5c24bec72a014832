"""Warm start: options.warm_start on the oracle (CPU pins) and on the HIP path (GPU parity).

Reference behaviour restated: a fresh solver object with warm_start takes the "provided trajectory" branch
(ipddp_solver.cpp:733-816) -- what CDDP::solve() always does, because it creates a new solver per call
(cddp_core.cpp:235-270); a re-initialised solver object takes the "existing solver state" branch (:675-731), which
keeps slack / dual / terminal variables (pins: tests/cddp_core/test_ipddp_solver.cpp:1306-1380).  On the GPU the
handle is the solver object."""
import numpy as np
import pytest


def rel(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    if a.size == 0:
        return 0.0
    return float(np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b))))


# ------------------------------------------------------------------------------------------------ CPU pins
def test_oracle_warm_start_preserves_path_dual_slack_state(api, oracle_built):   # test_ipddp_solver.cpp:1306-1335
    p = api.scalar_integrator_problem(8, path_constraint=True)
    o = api.Oracle(p); o.set_initial(p.x0); o.initialize()
    o.set_path_interior(0.42, 0.73)
    o.set_warm_start(True); o.initialize()
    S, Y, G = o.duals()
    assert np.max(np.abs(S - 0.42)) < 1e-12 and np.max(np.abs(Y - 0.73)) < 1e-12


def test_oracle_warm_start_preserves_terminal_interior_state(api, oracle_built):   # :1337-1360
    p = api.scalar_integrator_problem(8, terminal_inequality=True)
    o = api.Oracle(p); o.set_initial(p.x0); o.initialize()
    o.set_terminal_interior(0.37, 0.61)
    o.set_warm_start(True); o.initialize()
    ST, YT, GT, LT = o.terminal()
    assert abs(ST[0] - 0.37) < 1e-12 and abs(YT[0] - 0.61) < 1e-12


def test_oracle_warm_start_preserves_terminal_equality_multiplier(api, oracle_built):   # :1362-1380
    p = api.scalar_integrator_problem(8, terminal_equality=True)
    o = api.Oracle(p); o.set_initial(p.x0); o.initialize()
    o.set_terminal_eq_multiplier([0.53])
    o.set_warm_start(True); o.initialize()
    ST, YT, GT, LT = o.terminal()
    assert abs(LT[0] - 0.53) < 1e-12


@pytest.mark.parametrize("solver", ["ipddp", "clddp"])
def test_oracle_public_warm_start_converges_within_bound(api, oracle_built, solver):
    """test_ipddp_solver.cpp:474-549 / test_clddp_solver.cpp:153-230: previous solution as the initial trajectory of
    a NEW solver with warm_start: converges, iterations <= cold + 5."""
    sv = api.SOLVER_IPDDP if solver == "ipddp" else api.SOLVER_CLDDP
    p = api.pendulum_problem(sv, True, 100)
    o = api.Oracle(p); o.set_initial(p.x0); r = o.solve(); X, U = o.trajectory()
    assert r["status"] in (api.STATUS_OPTIMAL, api.STATUS_ACCEPTABLE)
    o2 = api.Oracle(p); o2.set_warm_start(True); o2.set_initial(p.x0, U, X); r2 = o2.solve()
    assert r2["status"] in (api.STATUS_OPTIMAL, api.STATUS_ACCEPTABLE)
    assert r2["iterations"] <= r["iterations"] + 5


# ------------------------------------------------------------------------------------------------ GPU parity
gpu = pytest.mark.gpu


@gpu
def test_hip_warm_start_preserves_interior_state(api):
    p = api.scalar_integrator_problem(8, path_constraint=True)
    B = 3
    x0 = np.tile(p.x0, (B, 1))
    hs = api.HipBatchSolver(p, B); hs.set_initial(x0); hs.initialize()
    m = hs.m
    hs.set_duals(np.full((B, p.N, m), 0.42), np.full((B, p.N, m), 0.73))
    hs.set_warm_start(True); hs.initialize()
    S, Y, G = hs.duals()
    assert np.max(np.abs(S - 0.42)) < 1e-12 and np.max(np.abs(Y - 0.73)) < 1e-12
    hs.close()
    p = api.scalar_integrator_problem(8, terminal_inequality=True)
    hs = api.HipBatchSolver(p, B); hs.set_initial(np.tile(p.x0, (B, 1))); hs.initialize()
    hs.set_terminal_state(np.full((B, 1), 0.37), np.full((B, 1), 0.61), None)
    hs.set_warm_start(True); hs.initialize()
    ST, YT, GT, LT = hs.terminal()
    assert np.max(np.abs(ST - 0.37)) < 1e-12 and np.max(np.abs(YT - 0.61)) < 1e-12
    hs.close()
    p = api.scalar_integrator_problem(8, terminal_equality=True)
    hs = api.HipBatchSolver(p, B); hs.set_initial(np.tile(p.x0, (B, 1))); hs.initialize()
    hs.set_terminal_state(None, None, np.full((B, 1), 0.53))
    hs.set_warm_start(True); hs.initialize()
    ST, YT, GT, LT = hs.terminal()
    assert np.max(np.abs(LT - 0.53)) < 1e-12
    hs.close()


CASES = ["pendulum_ipddp_box", "pendulum_clddp_box", "cartpole_ipddp_box", "unicycle_ipddp_box_ball", "pendulum_ipddp_unc",
         "path_term_eq", "path_term_ineq"]


@gpu
@pytest.mark.parametrize("case", CASES)
def test_hip_warm_start_matches_oracle(api, case):
    """(1) cold solve; (2) NEW handle / NEW oracle with warm_start and the cold solution as the provided trajectory
    (the CDDP::solve() path); (3) the SAME handle / oracle object re-solved from a perturbed x0 (existing solver
    state).  Iteration count, status, sweep / rollout counts and cost must agree at every stage."""
    import test_gpu_parity as T
    p = T.make(api, case)
    B = 4
    x0 = api.batch_x0(p, B, 20261001, T.spread_for(p))
    U0 = api.batch_U0(p, B)
    hs = api.HipBatchSolver(p, B); hs.set_initial(x0, U0); hs.solve()
    r0 = hs.results(); X, U = hs.trajectory()
    # (2) provided-trajectory warm start on fresh objects
    p2 = T.make(api, case); p2.options.warm_start = 1
    hw = api.HipBatchSolver(p2, B); hw.set_initial(x0, U, X); hw.solve()
    rw = hw.results()
    dx = np.zeros_like(x0); dx[:, 0] = 0.03
    hw.set_initial_state(x0 + dx); hw.solve()          # (3) existing solver state, new x0, controls kept
    rr = hw.results()
    Xr, Ur = hw.trajectory()
    for b in range(B):
        o = api.Oracle(p); o.set_initial(x0[b], None if U0 is None else U0[b]); q0 = o.solve(); Xo, Uo = o.trajectory()
        assert q0["iterations"] == r0["iterations"][b] and q0["status"] == r0["status"][b]
        ow = api.Oracle(p2); ow.set_initial(x0[b], Uo, Xo); qw = ow.solve()
        assert qw["iterations"] == rw["iterations"][b] and qw["status"] == rw["status"][b], (case, b, "provided")
        assert qw["n_backward"] == rw["n_backward"][b] and qw["n_forward"] == rw["n_forward"][b]
        if np.isfinite(qw["final_objective"]):
            assert rel(rw["final_objective"][b], qw["final_objective"]) < 1e-7
        ow.update_initial(x0[b] + dx[b]); qr = ow.solve()
        assert qr["iterations"] == rr["iterations"][b] and qr["status"] == rr["status"][b], (case, b, "existing")
        assert rel(rr["final_objective"][b], qr["final_objective"]) < 1e-6
        Xq, Uq = ow.trajectory()
        if qr["status"] in (api.STATUS_OPTIMAL, api.STATUS_ACCEPTABLE):
            assert rel(Ur[b], Uq) < 1e-5
    hs.close(); hw.close()
