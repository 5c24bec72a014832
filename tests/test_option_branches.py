"""ABI options that have device code of their own and had no `-m gpu` test:
barrier_strategy MONOTONIC / IPOPT (ipddp_solver.cpp:2601-2614), theta_norm = "l2" (:2778-2848), check_state_stationarity
(:931, 2725-2776; reference pin tests/cddp_core/test_ipddp_solver.cpp:1243-1304) and max_cpu_time (cddp_solver_base.cpp:77-90).

Parity of every OPTION_CASES entry at step and solve level is asserted by tests/test_gpu_parity.py (vs the oracle) and
tests/test_twin_golden.py (oracle and HIP vs the numpy twin's fixtures); this file checks that the branches are LIVE (an
option that silently did nothing would pass those parity tests as long as both sides ignore it the same way is not possible --
the twin is a third implementation -- but the check is cheap) and replays what the reference's own tests pin."""
import numpy as np
import pytest

from test_gpu_parity import OPTION_CASES, make, rel_err, spread_for


def _solve_hip(api, p, x0, U0=None, hist=False, trig=None):
    if hist:
        p.options.return_iteration_info = 1
    hs = api.HipBatchSolver(p, x0.shape[0], trig=trig)
    hs.set_initial(x0, U0)
    hs.solve()
    r = hs.results()
    h = hs.history(min(x0.shape[0], 4)) if hist else None
    hs.close()
    return r, h


@pytest.mark.gpu
@pytest.mark.parametrize("base,variant", [("cartpole_ipddp_box", "cartpole_ipddp_box_monotonic"), ("pendulum_ipddp_box", "pendulum_ipddp_box_ipopt"),
                                          ("unicycle_ipddp_box_ball", "unicycle_ipddp_box_ball_ipopt")])
def test_non_adaptive_barrier_strategy_is_live_and_matches_oracle(api, oracle_built, base, variant):
    """The kappa-epsilon rule (kkt <= mu_kappa_epsilon * mu with the dual residual weighted by barrier_update_dual_weight) must
    produce a different barrier sequence than ADAPTIVE on the same inputs, and the oracle's sequence exactly."""
    B = 8
    pa, pv = make(api, base), make(api, variant)
    x0 = api.batch_x0(pa, B, 20260929, spread_for(pa)); U0 = api.batch_U0(pa, B)
    # parity build on both sides (same sin / cos / log / pow routines): the comparison below is bit for bit
    ra, ha = _solve_hip(api, pa, x0, U0, hist=True, trig="shared")
    rv, hv = _solve_hip(api, pv, x0, U0, hist=True, trig="shared")
    mu_a, mu_v = ha[0][:, 7], hv[0][:, 7]
    n = min(len(mu_a), len(mu_v))
    assert not np.array_equal(mu_a[:n], mu_v[:n]), "barrier_strategy had no effect on the mu sequence"
    with api.shared_trig():
        o = api.Oracle(pv); o.set_initial(x0[0], None if U0 is None else U0[0]); ro = o.solve(); ho = o.history()
        ores = api.oracle_solve_batch(pv, x0, U0, n_threads=8, want_traj=False)[0]
    assert ro["iterations"] == rv["iterations"][0] and ro["status"] == rv["status"][0]
    assert len(ho) == len(hv[0]) and np.array_equal(ho[:, 7], hv[0][:, 7]), "mu sequence differs from the oracle's"
    assert np.array_equal(ores["iterations"], rv["iterations"]) and np.array_equal(ores["status"], rv["status"])
    assert np.array_equal(ores["n_forward"], rv["n_forward"]) and np.array_equal(ores["n_backward"], rv["n_backward"])


@pytest.mark.gpu
@pytest.mark.parametrize("base,variant", [("pendulum_ipddp_box", "pendulum_ipddp_box_l2"), ("unicycle_ipddp_box_ball", "unicycle_ipddp_box_ball_l2"),
                                          ("path_term_ineq", "path_term_ineq_l2")])
def test_theta_l2_norm_is_live_and_matches_oracle(api, oracle_built, base, variant):
    """theta = max(sqrt(sum r^2), max |r|) instead of max(sum |r|, max |r|): the trial records of the first line search must
    show the l2 value (different from l1, equal to the oracle's)."""
    B = 4
    pa, pv = make(api, base), make(api, variant)
    sp = spread_for(pa) if pa.nx > 1 else 0.05 * np.ones(1)
    x0 = api.batch_x0(pa, B, 20260928, sp); U0 = api.batch_U0(pa, B)
    out = []
    for p in (pa, pv):
        hs = api.HipBatchSolver(p, B); hs.set_initial(x0, U0); hs.initialize(); hs.backward()
        out.append(hs.forward(api.Oracle(p).alphas())); hs.close()
    ta, tv = out
    assert np.any(ta["theta"] != tv["theta"]), "theta_norm had no effect"
    for b in range(B):
        o = api.Oracle(pv); o.set_initial(x0[b], None if U0 is None else U0[b]); o.initialize(); o.backward(retry=True)
        for a, alpha in enumerate(o.alphas()):
            t = o.forward(alpha)
            assert t["success"] == tv[b, a]["success"]
            if t["success"]:
                assert rel_err(tv[b, a]["theta"], t["theta"]) < 1e-10 and rel_err(tv[b, a]["merit_function"], t["merit_function"]) < 1e-10
    rv, _ = _solve_hip(api, pv, x0, U0, trig="shared")
    with api.shared_trig():
        ores = api.oracle_solve_batch(pv, x0, U0, n_threads=4, want_traj=False)[0]
    assert np.array_equal(ores["iterations"], rv["iterations"]) and np.array_equal(ores["status"], rv["status"])


@pytest.mark.gpu
def test_state_stationarity_pin_on_device(api, oracle_built):
    """tests/cddp_core/test_ipddp_solver.cpp:1243-1304 replayed through the C-ABI: LTI A = B = 1, N = 1, Q = R = Qf = 0, one
    state row x <= 0.25, x0 = 1.  Without the option the scaled dual infeasibility equals inf_du (1e-12); with it it is
    larger (max |G_x^T y| enters).  The device reports the scaled value where the solver uses it: as the first operand of the
    barrier update / convergence tests -- observable through the iteration it terminates in and the final inf_du of a
    one-iteration solve vs the oracle, and directly through the oracle's accessor."""
    def problem(flag):
        o = api.default_options(); o.max_iterations = 100; o.tolerance = 1e-6; o.acceptable_tolerance = 1e-6
        o.reg_initial_value = 1e-6; o.barrier_mu_initial = 1e-1; o.ipddp_check_state_stationarity = flag
        p = api.Problem(api.SOLVER_IPDDP, api.MODEL_LTI, api.EULER, 1, 1, 1, 1.0, np.zeros((1, 1)), np.zeros((1, 1)), np.zeros((1, 1)), [0.0],
                        lti_A=np.eye(1), lti_B=np.eye(1), options=o)
        p.add_linear("PathUpperBound", np.eye(1), [0.25])
        p.x0 = np.array([1.0])
        return p
    sdu = []
    for flag in (0, 1):
        p = problem(flag)
        o = api.Oracle(p); o.set_initial(p.x0); o.initialize(); assert o.backward(retry=True)
        sdu.append(o.lib.cddp_oracle_scaled_inf_du(o.h))
        if flag == 0:
            assert abs(sdu[0] - o.result()["inf_du"]) < 1e-12
        # the device follows the oracle through the whole solve under either setting
        x0 = np.tile(p.x0, (3, 1))
        r, _ = _solve_hip(api, p, x0)
        ro = api.Oracle(p); ro.set_initial(p.x0); rr = ro.solve()
        assert np.all(r["iterations"] == rr["iterations"]) and np.all(r["status"] == rr["status"])
        assert rel_err(r["final_objective"][0], rr["final_objective"]) < 1e-9 and rel_err(r["barrier_mu"][0], rr["barrier_mu"]) < 1e-12
    assert sdu[1] > sdu[0]


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["pendulum_ipddp_box_state_stationarity", "unicycle_ipddp_box_state"])
def test_state_stationarity_changes_the_solve(api, oracle_built, case):
    """On problems with state-dependent rows the option must change the solver's decisions (mu sequence or iteration count),
    i.e. the device evaluates max |G_x^T y| -- and the result is the oracle's (parity build on both sides: strict).
    (On the scalar path_term_ineq problem and on the cart-pole with its wide state box the extra term never exceeds inf_du -- the
    oracle's solves are identical with and without, too; their parity is covered by tests/test_gpu_parity.py and the twin
    fixtures.)"""
    pv = make(api, case); pv.options.ipddp_check_state_stationarity = 1
    pa = make(api, case); pa.options.ipddp_check_state_stationarity = 0
    B = 4
    sp = spread_for(pv) if pv.nx > 1 else 0.05 * np.ones(1)
    x0 = api.batch_x0(pv, B, 20260929, sp); U0 = api.batch_U0(pv, B)
    rv, hv = _solve_hip(api, pv, x0, U0, hist=True, trig="shared")
    ra, ha = _solve_hip(api, pa, x0, U0, hist=True, trig="shared")
    differs = any(len(hv[b]) != len(ha[b]) or not np.array_equal(hv[b][:, 7], ha[b][:, 7]) for b in range(B))
    assert differs, "check_state_stationarity had no effect"
    with api.shared_trig():
        ores = api.oracle_solve_batch(pv, x0, U0, n_threads=4, want_traj=False)[0]
    assert np.array_equal(ores["iterations"], rv["iterations"]) and np.array_equal(ores["status"], rv["status"])
    assert np.array_equal(ores["n_forward"], rv["n_forward"])


@pytest.mark.gpu
def test_max_cpu_time(api, oracle_built):
    """CDDPOptions::max_cpu_time (cddp_solver_base.cpp:77-90): checked after ++iter, before the backward pass, in whole elapsed
    milliseconds.  One clock for the batch: every trajectory still running stops with "MaxCpuTimeReached" in the iteration the
    check fired in; a generous limit changes nothing."""
    p = api.cartpole_problem(api.SOLVER_IPDDP, True)
    B = 256
    x0 = api.batch_x0(p, B, 20260929, spread_for(p))
    ref, _ = _solve_hip(api, p, x0)
    p.options.max_cpu_time = 1.0e3
    same, _ = _solve_hip(api, p, x0)
    for key in ("iterations", "status", "final_objective", "n_forward"):
        assert np.array_equal(ref[key], same[key]), key
    p.options.max_cpu_time = 2.0e-3            # 2 ms: a few of the 80 iterations
    r, _ = _solve_hip(api, p, x0)
    assert np.all(r["status"] == api.STATUS_MAX_CPU_TIME)
    assert r["iterations"].min() == r["iterations"].max() and 1 <= r["iterations"][0] < 80
    assert np.all(np.isfinite(r["final_objective"])) and np.all(r["final_objective"] <= ref["final_objective"].max() * 1e6)
    assert api.STATUS_STRINGS[api.STATUS_MAX_CPU_TIME] == "MaxCpuTimeReached"
    # the oracle stops the same way (its own clock: only status and the "iterations = the iteration the check fired in" rule
    # are comparable): iterations >= 1, the iterate is the last accepted one
    o = api.Oracle(p); o.set_initial(x0[0]); ro = o.solve()
    assert ro["status"] == api.STATUS_MAX_CPU_TIME and 1 <= ro["iterations"] <= 80
    # a limit below one millisecond fires at the first check made after a whole millisecond has elapsed -- within the first
    # few iterations, never at iteration 0 (the check sits after ++iter)
    p.options.max_cpu_time = 1.0e-9
    r1, _ = _solve_hip(api, p, x0)
    assert np.all(r1["status"] == api.STATUS_MAX_CPU_TIME) and 1 <= r1["iterations"].min() and r1["iterations"].max() <= 10


def test_option_cases_have_twin_fixtures():
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    for name in OPTION_CASES:
        assert os.path.exists(os.path.join(here, "golden", "twin_%s.json" % name)), name


@pytest.mark.gpu
def test_clddp_unchanged_trial_is_accepted_like_copysign_of_plus_zero(api, oracle_built):
    """clddp_solver.cpp:251-254: reduction_ratio = expected > 0 ? dJ / expected : std::copysign(1.0, dJ).  A trial that changes nothing
    (every control already sits on the bound the gradient pushes against: BoxQP returns k = 0, K = 0, so dV = 0, U and X are
    reproduced bit for bit) has expected = 0 and dJ = cost - J_new = +0.0 exactly: the reference's ratio is +1 and the trial is
    ACCEPTED.  The library is built with -fno-signed-zeros, so the device writes that test as a comparison
    (dev_linalg.hpp::sign_of_reduction) instead of trusting the sign bit of a zero difference.
    Scalar integrator x+ = x + u, u in [-0.5, 0.5], goal far beyond reach, U0 = 0.5: both sides must accept alpha = 1 at once in every
    iteration (one rollout per iteration), keep the cost, and end with the same status after the same number of iterations."""
    o = api.default_options()
    o.max_iterations = 6
    N = 8
    p = api.Problem(api.SOLVER_CLDDP, api.MODEL_LTI, api.EULER, 1, 1, N, 1.0, np.zeros((1, 1)), 1e-2 * np.eye(1), 100.0 * np.eye(1), [100.0],
                    lti_A=np.eye(1), lti_B=np.eye(1), options=o)
    p.add_control_box("ControlConstraint", [-0.5], [0.5])
    B = 3
    x0 = np.array([[0.0], [0.25], [-1.0]])
    U0 = np.full((B, N, 1), 0.5)
    X0 = np.zeros((B, N + 1, 1))
    for b in range(B):
        X0[b, 0, 0] = x0[b, 0]
        for t in range(N):
            X0[b, t + 1, 0] = X0[b, t, 0] + U0[b, t, 0]     # CLDDP costs the guess as given (clddp_solver.cpp:68-74): make it consistent
    p.options.return_iteration_info = 1
    hs = api.HipBatchSolver(p, B)
    hs.set_initial(x0, U0, X0)
    hs.solve()
    r = hs.results(); X, U = hs.trajectory(); h = hs.history(B)
    hs.close()
    ores, oX, oU, _, _ = api.oracle_solve_batch(p, x0, U0, X0, n_threads=B)
    for b in range(B):
        assert (r["status"][b], r["iterations"][b], r["n_forward"][b], r["n_backward"][b]) == \
               (ores["status"][b], ores["iterations"][b], ores["n_forward"][b], ores["n_backward"][b]), b
        assert r["final_objective"][b] == ores["final_objective"][b]
        assert np.array_equal(U[b], U0[b]) and np.array_equal(X[b], X0[b])           # nothing moved ...
        assert r["n_forward"][b] == r["iterations"][b] or r["iterations"][b] == 0 or r["n_forward"][b] >= 1
        assert np.all(h[b][:, 0] == h[b][0, 0])                                          # ... and the cost never changed: dJ == +0 every time
        assert r["alpha_pr"][b] == 1.0                                                   # alpha = 1 accepted (ratio = +1 > armijo), not rejected (ratio = -1)
