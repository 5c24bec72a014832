"""Pin the CPU oracle against every known answer the reference's own tests hold for this path
(SURVEY.md section 8(c)).  The reference pins no gain / value / iteration-count number ("parity
unpinned"); what it does pin -- status strings, cost decrease, terminal-state bounds of the
scalar-integrator regressions, filter / barrier behaviours, plugin closed forms -- is replayed here
with the same problem constants.  CPU only (no GPU needed)."""
import numpy as np
import pytest

OK = ("OptimalSolutionFound", "AcceptableSolutionFound")


def status(api, r):
    return api.STATUS_STRINGS[int(r["status"])]


def regression_options(api):
    """makeIpddpRegressionOptions, tests/cddp_core/test_ipddp_solver.cpp:139-154."""
    o = api.default_options()
    o.max_iterations = 20; o.tolerance = 1e-6; o.acceptable_tolerance = 1e-6
    o.reg_initial_value = 1e-6; o.barrier_mu_initial = 1e-1
    o.ipddp_slack_var_init_scale = 1e-2; o.ipddp_dual_var_init_scale = 1e-1
    return o


def lti_scalar(api, horizon, x0, goal, R, Qf, options, solver=None):
    S = api
    p = S.Problem(S.SOLVER_IPDDP if solver is None else solver, S.MODEL_LTI, S.EULER, 1, 1, horizon, 1.0,
                  np.zeros((1, 1)), R * np.eye(1), Qf * np.eye(1), [goal], lti_A=np.eye(1), lti_B=np.eye(1), options=options)
    p.x0 = np.array([x0])
    return p


# ------------------------------------------------------------------ solver integration pins
def pendulum500(api, solver):
    """tests/cddp_core/test_clddp_solver.cpp:28-151, test_ipddp_solver.cpp:349-472."""
    o = api.default_options()
    o.max_iterations = 100; o.tolerance = 1e-3; o.acceptable_tolerance = 1e-4; o.reg_initial_value = 1e-6
    p = api.Problem(solver, api.MODEL_PENDULUM, api.EULER, 2, 1, 500, 0.05, np.zeros((2, 2)), 0.1 * np.eye(1),
                    100.0 * np.eye(2), [0.0, 0.0], model_params=[1.0, 1.0, 0.0, 9.81], options=o)
    p.add_control_box("ControlConstraint", [-10.0], [10.0])
    p.x0 = np.array([np.pi, 0.0])
    return p


@pytest.mark.parametrize("solver", ["CLDDP", "IPDDP"])
def test_pendulum_n500_converges(api, oracle_built, solver):
    p = pendulum500(api, api.SOLVER_CLDDP if solver == "CLDDP" else api.SOLVER_IPDDP)
    o = api.Oracle(p)
    X0 = np.tile(p.x0, (p.N + 1, 1)); U0 = np.zeros((p.N, 1))
    J_init = o.cost(X0, U0)
    o.set_initial(p.x0)
    r = o.solve()
    assert status(api, r) in OK, status(api, r)
    assert r["iterations"] > 0
    assert r["final_objective"] < J_init


def test_clddp_unicycle_reference_test(api, oracle_built):
    """tests/cddp_core/test_clddp_solver.cpp:231-297: X = zeros, 20 iterations, parallel line search."""
    o = api.default_options(); o.max_iterations = 20; o.tolerance = 1e-2; o.enable_parallel = 1
    Qf = 0.5 * np.diag([50.0, 50.0, 10.0])
    p = api.Problem(api.SOLVER_CLDDP, api.MODEL_UNICYCLE, api.EULER, 3, 2, 100, 0.03, np.zeros((3, 3)), 0.5 * np.eye(2),
                    Qf, [2.0, 2.0, np.pi / 2], options=o)
    p.add_control_box("ControlConstraint", [-1.0, -np.pi], [1.0, np.pi])
    orc = api.Oracle(p)
    orc.set_initial([0.0, 0.0, np.pi / 4], X0=np.zeros((101, 3)))
    r = orc.solve()
    assert status(api, r) in OK, status(api, r)


def test_ipddp_unicycle_reference_test(api, oracle_built):
    """tests/cddp_core/test_ipddp_solver.cpp:552-618 (same plant, IPDDP)."""
    o = api.default_options(); o.max_iterations = 100; o.tolerance = 1e-3; o.acceptable_tolerance = 1e-4
    Qf = 0.5 * np.diag([50.0, 50.0, 10.0])
    p = api.Problem(api.SOLVER_IPDDP, api.MODEL_UNICYCLE, api.EULER, 3, 2, 100, 0.03, np.zeros((3, 3)), 0.5 * np.eye(2),
                    Qf, [2.0, 2.0, np.pi / 2], options=o)
    p.add_control_box("ControlConstraint", [-1.0, -np.pi], [1.0, np.pi])
    orc = api.Oracle(p)
    orc.set_initial([0.0, 0.0, np.pi / 4])
    r = orc.solve()
    assert status(api, r) in OK, status(api, r)
    X, _ = orc.trajectory()
    assert np.linalg.norm(X[-1][:2] - np.array([2.0, 2.0])) < 0.2


# ------------------------------------------------------------------ scalar-integrator regressions
def test_terminal_inequality_only(api, oracle_built):
    """test_ipddp_solver.cpp:1147-1207: x_N <= 1e-4 and ~0 +- 1e-3."""
    o = api.default_options(); o.max_iterations = 60; o.tolerance = 1e-6; o.acceptable_tolerance = 1e-6
    o.reg_initial_value = 1e-6; o.barrier_mu_initial = 1e-1
    p = lti_scalar(api, 8, 0.0, 1.0, 1e-2, 100.0, o)
    p.add_terminal_inequality("TerminalUpperBound", np.eye(1), np.zeros(1))
    orc = api.Oracle(p); orc.set_initial(p.x0)
    r = orc.solve()
    assert status(api, r) in OK, status(api, r)
    X, _ = orc.trajectory()
    assert X[-1, 0] <= 1e-4
    assert abs(X[-1, 0]) <= 1e-3


def scalar_problem(api, options, path, term_ineq):
    """makeScalarIntegratorProblem, test_ipddp_solver.cpp:156-207."""
    p = lti_scalar(api, 4, 1.0, 0.0, 1e-2, 1.0, options)
    if path:
        p.add_linear("PathUpperBound", np.eye(1), [0.25])
    if term_ineq:
        p.add_terminal_inequality("TerminalUpperBound", np.eye(1), [0.25])
    return p


def test_path_only_filter_empty_theta_positive(api, oracle_built):
    """test_ipddp_solver.cpp:1209-1241."""
    p = scalar_problem(api, regression_options(api), True, False)
    orc = api.Oracle(p); orc.set_initial(p.x0, X0=np.ones((5, 1))); orc.initialize()
    assert orc.lib.cddp_oracle_filter_size(orc.h) == 0
    assert orc.lib.cddp_oracle_filter_theta(orc.h) > 0.0
    orc.lib.cddp_oracle_update_barrier(orc.h, 1)
    n = orc.lib.cddp_oracle_filter_size(orc.h)
    ref = orc.lib.cddp_oracle_filter_theta(orc.h) if n == 0 else orc.lib.cddp_oracle_filter_back_violation(orc.h)
    assert ref > 0.0


def test_scaled_dual_infeasibility_state_stationarity(api, oracle_built):
    """test_ipddp_solver.cpp:1243-1304."""
    vals = {}
    for flag in (0, 1):
        o = regression_options(api); o.ipddp_check_state_stationarity = flag
        p = lti_scalar(api, 1, 1.0, 0.0, 0.0, 0.0, o)
        p.add_linear("PathUpperBound", np.eye(1), [0.25])
        orc = api.Oracle(p); orc.set_initial(p.x0); orc.initialize()
        assert orc.backward(retry=False) == 1
        vals[flag] = (orc.lib.cddp_oracle_scaled_inf_du(orc.h), orc.result()["inf_du"])
    assert abs(vals[0][0] - vals[0][1]) <= 1e-12
    assert vals[1][0] > vals[0][0]


@pytest.mark.parametrize("Qf,path_b", [(0.0, 10.0)])
def test_path_and_terminal_equality(api, oracle_built, Qf, path_b):
    """test_ipddp_solver.cpp:1382-1438: |x_N| <= 1e-4."""
    o = regression_options(api); o.max_iterations = 100
    p = lti_scalar(api, 8, 1.0, 0.0, 1e-2, Qf, o)
    p.add_linear("LoosePathUpperBound", np.eye(1), [path_b])
    p.add_terminal_equality("TerminalTarget", [0.0])
    orc = api.Oracle(p); orc.set_initial(p.x0)
    r = orc.solve()
    assert status(api, r) in OK, status(api, r)
    X, _ = orc.trajectory()
    assert abs(X[-1, 0]) <= 1e-4


def test_terminal_equality_only(api, oracle_built):
    """test_ipddp_solver.cpp:1580-1637."""
    o = api.default_options(); o.max_iterations = 100; o.tolerance = 1e-6; o.acceptable_tolerance = 1e-6; o.reg_initial_value = 1e-6
    p = lti_scalar(api, 8, 1.0, 0.0, 1e-2, 1.0, o)
    p.add_terminal_equality("TerminalTarget", [0.0])
    orc = api.Oracle(p); orc.set_initial(p.x0)
    r = orc.solve()
    assert status(api, r) in OK, status(api, r)
    X, _ = orc.trajectory()
    assert abs(X[-1, 0]) <= 1e-4


def test_terminal_equality_backward_tracks_stationarity(api, oracle_built):
    """test_ipddp_solver.cpp:1466-1510: R=1e8 -> inf_du > 1e-4, step_norm < 1e-6."""
    o = regression_options(api); o.reg_initial_value = 1e-12
    p = lti_scalar(api, 1, 1.0, 0.0, 1e8, 0.0, o)
    p.add_terminal_equality("TerminalTarget", [0.0])
    orc = api.Oracle(p); orc.set_initial(p.x0); orc.initialize()
    assert orc.backward(retry=False) == 1
    r = orc.result()
    assert r["inf_du"] > 1e-4
    assert r["step_norm"] < 1e-6


# ------------------------------------------------------------------ plugin-surface closed forms
def test_pendulum_closed_forms(api, oracle_built):
    """tests/test_hessian.cpp:95-134: +sin gravity convention, A and B closed forms, 1e-10."""
    p = api.pendulum_problem(api.SOLVER_CLDDP, False)
    o = api.Oracle(p)
    th, thd, u = 0.7, -0.3, 1.2
    xd, xn, Fx, Fu = o.dynamics([th, thd], [u])
    l, m, b, g = 0.5, 1.0, 0.01, 9.81
    assert abs(xd[0] - thd) < 1e-10
    assert abs(xd[1] - (u - b * thd + m * g * l * np.sin(th)) / (m * l * l)) < 1e-10
    assert np.allclose(Fx, [[0, 1], [(g / l) * np.cos(th), -b / (m * l * l)]], atol=1e-10)
    assert np.allclose(Fu, [[0], [1 / (m * l * l)]], atol=1e-10)
    assert np.allclose(xn, np.array([th, thd]) + 0.02 * xd, atol=1e-14)


def _fd_jac(o, x, u, h=1e-6):
    x = np.asarray(x, float); u = np.asarray(u, float)
    Fx = np.zeros((x.size, x.size)); Fu = np.zeros((x.size, u.size))
    for i in range(x.size):
        e = np.zeros_like(x); e[i] = h
        Fx[:, i] = (o.dynamics(x + e, u)[0] - o.dynamics(x - e, u)[0]) / (2 * h)
    for i in range(u.size):
        e = np.zeros_like(u); e[i] = h
        Fu[:, i] = (o.dynamics(x, u + e)[0] - o.dynamics(x, u - e)[0]) / (2 * h)
    return Fx, Fu


def test_cartpole_jacobians_match_fd(api, oracle_built):
    """tests/dynamics_model/test_cartpole.cpp:66-97: autodiff A, B == analytic to 1e-9 (here: vs FD)."""
    p = api.cartpole_problem(api.SOLVER_CLDDP, False)
    o = api.Oracle(p)
    x = [0.1, 0.4, -0.2, 0.7]; u = [1.3]
    _, _, Fx, Fu = o.dynamics(x, u)
    Fx_fd, Fu_fd = _fd_jac(o, x, u)
    assert np.allclose(Fx, Fx_fd, atol=1e-6) and np.allclose(Fu, Fu_fd, atol=1e-6)


def test_quadrotor_hover_and_jacobian(api, oracle_built):
    """tests/dynamics_model/test_quadrotor.cpp:194-212: hover thrust -> xdot == 0 (1e-10);
    :262-490 Jacobians vs finite differences."""
    p = api.quadrotor_problem(api.SOLVER_IPDDP)
    o = api.Oracle(p)
    x = np.zeros(13); x[3] = 1.0
    u = np.ones(4) * 9.81 / 4.0
    xd = o.dynamics(x, u)[0]
    assert np.max(np.abs(xd)) < 1e-10
    rng = np.random.default_rng(3)
    x = rng.normal(size=13) * 0.3; x[3] += 1.0
    u = 2.0 + rng.normal(size=4) * 0.3
    _, _, Fx, Fu = o.dynamics(x, u)
    Fx_fd, Fu_fd = _fd_jac(o, x, u)
    assert np.allclose(Fx, Fx_fd, atol=1e-5) and np.allclose(Fu, Fu_fd, atol=1e-5)


def test_constraints_closed_forms(api, oracle_built):
    """tests/cddp_core/test_constraint.cpp: box rows [-v; v], upper [-lb; ub]; ball g = r^2 - |x-c|^2."""
    p = api.unicycle_problem(api.SOLVER_IPDDP, 10, True)
    o = api.Oracle(p)
    x = np.array([0.5, 1.5, 0.2]); u = np.array([0.3, -2.0])
    g, gx, gu = o.constraint_eval(x, u)
    lb = np.array([-1.1, -np.pi]); ub = np.array([1.1, np.pi])
    assert np.allclose(g[:2], lb - u) and np.allclose(g[2:4], u - ub)
    assert np.allclose(g[4], 0.4 ** 2 - np.sum((x[:2] - 1.0) ** 2))
    assert np.allclose(gu[:4], np.vstack([-np.eye(2), np.eye(2)])) and np.allclose(gx[:4], 0)
    assert np.allclose(gx[4], [-2 * (x[0] - 1), -2 * (x[1] - 1), 0.0]) and np.allclose(gu[4], 0)


def test_quadratic_objective(api, oracle_built):
    """tests/cddp_core/test_objective.cpp: cost = sum e'(Q dt)e + u'(R dt)u + e_N' Qf e_N (no 1/2)."""
    p = api.cartpole_problem(api.SOLVER_CLDDP, False, horizon=5)
    o = api.Oracle(p)
    rng = np.random.default_rng(0)
    X = rng.normal(size=(6, 4)); U = rng.normal(size=(5, 1))
    ref = np.array([0, np.pi, 0, 0])
    want = sum(0.1 * 0.05 * float(U[t] @ U[t]) for t in range(5)) + 100.0 * float((X[5] - ref) @ (X[5] - ref))
    assert abs(o.cost(X, U) - want) < 1e-9


def test_line_search_ladder(api, oracle_built):
    """detail::buildLineSearchAlphas (cddp_context_utils.cpp:37-57): 11 alphas 1, .5, ... 2^-10."""
    p = api.pendulum_problem()
    a = api.Oracle(p).alphas()
    assert len(a) == 11 and np.allclose(a, 0.5 ** np.arange(11))
    p.options.ls_max_iterations = 40
    a = api.Oracle(p).alphas()
    assert a[-1] == 1e-8 and a[-2] >= 1e-8 and len(a) < 40


# ------------------------------------------------------------------ independent cross-checks
def test_unconstrained_sweep_equals_discrete_riccati(api, oracle_built):
    """For an LTI plant + quadratic cost the Gauss-Newton sweep IS the finite-horizon discrete Riccati
    recursion: an independent 10-line numpy check of K, V_xx."""
    rng = np.random.default_rng(7)
    nx, nu, N, dt = 2, 1, 30, 1.0
    A = np.eye(nx) + 0.1 * rng.normal(size=(nx, nx)); Bm = rng.normal(size=(nx, nu))
    Q = np.diag([1.0, 0.5]); R = np.array([[0.3]]); Qf = np.diag([5.0, 2.0])
    o = api.default_options(); o.max_iterations = 1; o.reg_initial_value = 0.0
    p = api.Problem(api.SOLVER_IPDDP, api.MODEL_LTI, api.EULER, nx, nu, N, dt, Q, R, Qf, np.zeros(nx), lti_A=A, lti_B=Bm, options=o)
    orc = api.Oracle(p); orc.set_initial(rng.normal(size=nx)); orc.initialize()
    assert orc.backward(retry=False) == 1
    K, _ = orc.gains(); _, Vxx = orc.value()
    P = 2 * Qf
    for t in range(N - 1, -1, -1):
        Quu = 2 * R * dt + Bm.T @ P @ Bm
        Kt = -np.linalg.solve(Quu, Bm.T @ P @ A)
        P = 2 * Q * dt + A.T @ P @ A + Kt.T @ Quu @ Kt + Kt.T @ Bm.T @ P @ A + A.T @ P @ Bm @ Kt
        P = 0.5 * (P + P.T)
        assert np.allclose(K[t], Kt, rtol=1e-9, atol=1e-11)
        assert np.allclose(Vxx[t], P, rtol=1e-9, atol=1e-10)


def test_ldlt_matches_numpy_and_accepts_indefinite(api, oracle_built):
    """Eigen LDLT semantics: pivoted, 'Success' for indefinite matrices."""
    rng = np.random.default_rng(11)
    for n in (1, 2, 3, 5, 7):
        M = rng.normal(size=(n, n)); A = M @ M.T + 0.1 * np.eye(n)
        B = rng.normal(size=(n, 3))
        X, ok = api.oracle_ldlt_solve(A, B)
        assert ok and np.allclose(A @ X, B, atol=1e-9)
    A = np.diag([2.0, -3.0, 1.0]); A[0, 1] = A[1, 0] = 0.5
    X, ok = api.oracle_ldlt_solve(A, np.eye(3))
    assert ok and np.allclose(A @ X, np.eye(3), atol=1e-12)


def test_boxqp_against_bruteforce(api, oracle_built):
    """BoxQP (boxqp.cpp): compare with projected-gradient brute force on random PD problems,
    and the reference's 15x15 style fixture property: solution inside the box, KKT sign conditions."""
    rng = np.random.default_rng(5)
    for n in (1, 2, 4, 7):
        M = rng.normal(size=(n, n)); H = M @ M.T + 0.5 * np.eye(n); g = rng.normal(size=n) * 3
        lo = -np.abs(rng.normal(size=n)); up = np.abs(rng.normal(size=n))
        x, st, free, it, fc = api.oracle_boxqp(H, g, lo, up, np.zeros(n))
        assert st in (4, 5), st                       # SUCCESS or ALL_CLAMPED
        assert np.all(x >= lo - 1e-15) and np.all(x <= up + 1e-15)
        grad = g + H @ x
        for i in range(n):
            if free[i]:
                assert abs(grad[i]) < 1e-6
            else:
                assert (x[i] == lo[i] and grad[i] > 0) or (x[i] == up[i] and grad[i] < 0)
        y = np.clip(np.zeros(n), lo, up)
        L = np.linalg.eigvalsh(H).max()
        for _ in range(20000):
            y = np.clip(y - (g + H @ y) / L, lo, up)
        assert np.allclose(x, y, atol=1e-6)
