"""The reference's Python portfolio regression (python/tests/test_portfolio.py:20-44) restated: the three demos of
examples/python_portfolio_lib.py that run on built-in plants (pendulum / cart-pole CLDDP, unicycle CLDDP -> IPDDP with the
ball obstacle) with the reference's thresholds, committed in tests/golden/portfolio_thresholds.json.

CPU: the oracle meets the thresholds (so they are meaningful for this restatement of the algorithm).
GPU: the pycddp-compatible front end (cddp-cpp_amd/pycddp_amd.py) meets them AND returns the oracle's solution --
iteration count, status, objective, trajectory -- for every demo, and `solve_batch` equals `oracle_solve_batch` row by row."""
import importlib.util
import json
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TH = json.load(open(os.path.join(REPO, "tests", "golden", "portfolio_thresholds.json")))


def _pycddp():
    name = "pycddp_amd"
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(name, os.path.join(REPO, "cddp-cpp_amd", "pycddp_amd.py"))
    mod = importlib.util.module_from_spec(spec); sys.modules[name] = mod; spec.loader.exec_module(mod)
    return mod


def _opts(api, max_iterations, **kw):
    o = api.default_options(); o.max_iterations = max_iterations; o.return_iteration_info = 1
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def _rollout(api, p, x0, U):
    o = api.Oracle(p)
    X = [np.asarray(x0, float)]
    for t in range(len(U)):
        X.append(o.dynamics(X[-1], U[t], t * p.dt)[1])
    return np.array(X)


# ---- the demos as C-ABI problems (python_portfolio_lib.py:282-470) ------------------------------------------------
def pendulum_demo(api):
    dt, N = 0.05, 120
    o = _opts(api, 150, tolerance=1e-5, acceptable_tolerance=1e-4, reg_initial_value=1e-6)
    p = api.Problem(api.SOLVER_CLDDP, api.MODEL_PENDULUM, api.EULER, 2, 1, N, dt, 0.1 * np.eye(2), 0.02 * np.eye(1), 200.0 * np.eye(2),
                    [np.pi, 0.0], model_params=[0.5, 1.0, 0.01, 9.81], options=o)
    p.add_control_box("control_limits", [-30.0], [30.0])          # NOT named "ControlConstraint": CLDDP runs unbounded (Appendix A.3)
    p.x0 = np.zeros(2)
    U = np.array([[8.0] if i < 25 else [0.0] for i in range(N)])
    return p, U, _rollout(api, p, p.x0, U), np.array([np.pi, 0.0])


def cartpole_demo(api):
    dt, N = 0.05, 100
    o = _opts(api, 120, tolerance=1e-6, acceptable_tolerance=1e-5, reg_initial_value=1e-5)
    xref = np.array([0.0, np.pi, 0.0, 0.0])
    p = api.Problem(api.SOLVER_CLDDP, api.MODEL_CARTPOLE, api.RK4, 4, 1, N, dt, np.zeros((4, 4)), 0.1 * np.eye(1), 80.0 * np.eye(4), xref,
                    model_params=[1.0, 0.2, 0.5, 9.81, 0.0], options=o)
    p.add_control_box("force_limits", [-5.0], [5.0])
    p.x0 = np.zeros(4)
    return p, np.zeros((N, 1)), np.tile(p.x0, (N + 1, 1)), xref


def unicycle_demo(api, solver, with_ball):
    dt, N = 0.03, 100
    o = _opts(api, 100, tolerance=1e-4)
    xref = np.array([2.0, 2.0, np.pi / 2])
    p = api.Problem(solver, api.MODEL_UNICYCLE, api.EULER, 3, 2, N, dt, np.zeros((3, 3)), 0.05 * np.eye(2), np.diag([100.0, 100.0, 50.0]), xref, options=o)
    p.add_control_box("control_limits", [-1.1, -np.pi], [1.1, np.pi])
    if with_ball:
        p.add_ball("obstacle", 0.4, [1.0, 1.0])
    p.x0 = np.array([0.0, 0.0, np.pi / 4])
    return p, np.zeros((N, 2)), np.tile(p.x0, (N + 1, 1)), xref


def _oracle_solve(api, p, U0, X0):
    o = api.Oracle(p); o.set_initial(p.x0, U0, X0); r = o.solve()
    X, U = o.trajectory()
    return r, X, U


def _check_thresholds(name, X, U, xref, inf_pr=None):
    t = TH[name]
    err = float(np.linalg.norm(X[-1] - xref))
    assert err < t["final_error_max"], (name, err)
    if "theta_max_min" in t:
        assert X[:, 0].max() > t["theta_max_min"] and np.abs(U[:, 0]).max() > t["abs_control_max_min"]
    if "final_primal_infeasibility_max" in t:
        assert inf_pr < t["final_primal_infeasibility_max"], (name, inf_pr)
    return err


def test_oracle_meets_the_portfolio_thresholds(api, oracle_built):
    p, U0, X0, xref = pendulum_demo(api)
    r, X, U = _oracle_solve(api, p, U0, X0)
    _check_thresholds("pendulum", X, U, xref)
    p, U0, X0, xref = cartpole_demo(api)
    r, X, U = _oracle_solve(api, p, U0, X0)
    _check_thresholds("cartpole", X, U, xref)
    pb, U0, X0, xref = unicycle_demo(api, api.SOLVER_CLDDP, False)          # the CLDDP baseline seeds the IPDDP solve (:395-405)
    rb, Xb, Ub = _oracle_solve(api, pb, U0, X0)
    p, _, _, _ = unicycle_demo(api, api.SOLVER_IPDDP, True)
    r, X, U = _oracle_solve(api, p, Ub, Xb)
    _check_thresholds("unicycle", X, U, xref, r["inf_pr"])


def _facade_problem(pycddp, which, xs=None, us=None):
    if which == "pendulum":
        dt, N, x0, xref = 0.05, 120, np.zeros(2), np.array([np.pi, 0.0])
        o = pycddp.CDDPOptions(); o.max_iterations = 150; o.tolerance = 1e-5; o.acceptable_tolerance = 1e-4; o.regularization.initial_value = 1e-6
        s = pycddp.CDDP(x0, xref, N, dt, o)
        s.set_dynamical_system(pycddp.Pendulum(dt, length=0.5, mass=1.0, damping=0.01))
        s.set_objective(pycddp.QuadraticObjective(0.1 * np.eye(2), 0.02 * np.eye(1), 200.0 * np.eye(2), xref, [], dt))
        s.add_constraint("control_limits", pycddp.ControlConstraint(np.array([-30.0]), np.array([30.0])))
        return s, pycddp.SolverType.CLDDP
    if which == "cartpole":
        dt, N, x0, xref = 0.05, 100, np.zeros(4), np.array([0.0, np.pi, 0.0, 0.0])
        o = pycddp.CDDPOptions(); o.max_iterations = 120; o.tolerance = 1e-6; o.acceptable_tolerance = 1e-5; o.regularization.initial_value = 1e-5
        s = pycddp.CDDP(x0, xref, N, dt, o)
        s.set_dynamical_system(pycddp.CartPole(dt))
        s.set_objective(pycddp.QuadraticObjective(np.zeros((4, 4)), 0.1 * np.eye(1), 80.0 * np.eye(4), xref, [], dt))
        s.add_constraint("force_limits", pycddp.ControlConstraint(np.array([-5.0]), np.array([5.0])))
        return s, pycddp.SolverType.CLDDP
    dt, N, x0, xref = 0.03, 100, np.array([0.0, 0.0, np.pi / 4]), np.array([2.0, 2.0, np.pi / 2])
    o = pycddp.CDDPOptions(); o.max_iterations = 100; o.tolerance = 1e-4
    s = pycddp.CDDP(x0, xref, N, dt, o)
    s.set_dynamical_system(pycddp.Unicycle(dt))
    s.set_objective(pycddp.QuadraticObjective(np.zeros((3, 3)), 0.05 * np.eye(2), np.diag([100.0, 100.0, 50.0]), xref, [], dt))
    s.add_constraint("control_limits", pycddp.ControlConstraint(np.array([-1.1, -np.pi]), np.array([1.1, np.pi])))
    if which == "unicycle_ipddp":
        s.add_constraint("obstacle", pycddp.BallConstraint(0.4, np.array([1.0, 1.0])))
        return s, pycddp.SolverType.IPDDP
    return s, pycddp.SolverType.CLDDP


def _rel(a, b):
    return float(np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b))))


def _same_as_oracle(api, sol, r, X, U, tol_obj=1e-6, tol_traj=1e-4):
    """Same status and iteration count; objective and trajectories within the given bounds (the sweep-level 1e-8 bar is
    tests/test_gpu_parity*.py's; a whole solve is only determined to the solver's own 1e-4 tolerance)."""
    assert sol.iterations_completed == r["iterations"] and sol.status_message == api.STATUS_STRINGS[int(r["status"])]
    assert abs(sol.final_objective - r["final_objective"]) <= tol_obj * max(1.0, abs(r["final_objective"]))
    assert _rel(np.stack(sol.state_trajectory), X) < tol_traj and _rel(np.stack(sol.control_trajectory), U) < tol_traj


@pytest.mark.gpu
def test_pycddp_front_end_meets_thresholds_and_equals_oracle(api, oracle_built):
    pycddp = _pycddp()
    # pendulum: seeded swing-up
    p, U0, X0, xref = pendulum_demo(api)
    s, st = _facade_problem(pycddp, "pendulum")
    s.set_initial_trajectory(list(X0), list(U0))
    sol = s.solve(st)
    assert sol.solver_name == "CLDDP"
    _check_thresholds("pendulum", np.stack(sol.state_trajectory), np.stack(sol.control_trajectory), xref)
    _same_as_oracle(api, sol, *_oracle_solve(api, p, U0, X0))
    # cart-pole
    p, U0, X0, xref = cartpole_demo(api)
    s, st = _facade_problem(pycddp, "cartpole")
    s.set_initial_trajectory(list(X0), list(U0))
    sol = s.solve(st)
    _check_thresholds("cartpole", np.stack(sol.state_trajectory), np.stack(sol.control_trajectory), xref)
    _same_as_oracle(api, sol, *_oracle_solve(api, p, U0, X0))
    # unicycle: CLDDP baseline, then IPDDP with the obstacle seeded from it
    pb, U0, X0, xref = unicycle_demo(api, api.SOLVER_CLDDP, False)
    sb, stb = _facade_problem(pycddp, "unicycle_clddp")
    sb.set_initial_trajectory(list(X0), list(U0))
    base = sb.solve(stb)
    rb, Xb, Ub = _oracle_solve(api, pb, U0, X0)
    _same_as_oracle(api, base, rb, Xb, Ub)
    s, st = _facade_problem(pycddp, "unicycle_ipddp")
    s.set_initial_trajectory(list(base.state_trajectory), list(base.control_trajectory))
    sol = s.solve(st)
    assert sol.solver_name == "IPDDP"
    _check_thresholds("unicycle", np.stack(sol.state_trajectory), np.stack(sol.control_trajectory), xref, sol.final_primal_infeasibility)
    p, _, _, _ = unicycle_demo(api, api.SOLVER_IPDDP, True)
    _same_as_oracle(api, sol, *_oracle_solve(api, p, np.stack(base.control_trajectory), np.stack(base.state_trajectory)))


@pytest.mark.gpu
def test_solve_batch_rows_equal_oracle_solve_batch(api, oracle_built):
    pycddp = _pycddp()
    for which, name in (("cartpole", "cartpole"), ("unicycle_ipddp", "unicycle")):
        s, st = _facade_problem(pycddp, which)
        x0 = np.asarray(s._x0, float)
        rng = np.random.default_rng(20261021)
        x0s = [x0 + (0.0 if b == 0 else 1.0) * rng.uniform(-0.03, 0.03, size=x0.shape) for b in range(40)]
        sols = s.solve_batch(x0s, st)
        p = cartpole_demo(api)[0] if which == "cartpole" else unicycle_demo(api, api.SOLVER_IPDDP, True)[0]
        ores, oX, oU, _, _ = api.oracle_solve_batch(p, np.stack(x0s), n_threads=8)
        assert len(sols) == 40
        # yardstick per row: the oracle against itself with every sin / cos moved by <= 1 ulp (oracle/models.hpp::trig_noise).  A
        # row whose solve is ill-conditioned in the rounding moves by 1e-5 under that noise alone; the HIP row is held to
        # 20 x that movement (floors 1e-7 / 1e-6), and rows whose (status, iterations) survive the noise must keep them.
        import ctypes
        lib = ctypes.CDLL(api.ORACLE_LIB_PATH)
        try:
            lib.cddp_oracle_set_trig_noise(1)
            nres, nX, nU, _, _ = api.oracle_solve_batch(p, np.stack(x0s), n_threads=8)
        finally:
            lib.cddp_oracle_set_trig_noise(0)
        n_stable = 0
        for b, sol in enumerate(sols):
            stable = nres[b]["iterations"] == ores[b]["iterations"] and nres[b]["status"] == ores[b]["status"]
            if not stable:
                continue
            n_stable += 1
            y_obj = abs(nres[b]["final_objective"] - ores[b]["final_objective"]) / max(1.0, abs(ores[b]["final_objective"]))
            y_traj = max(_rel(nX[b], oX[b]), _rel(nU[b], oU[b]))
            _same_as_oracle(api, sol, ores[b], oX[b], oU[b], tol_obj=max(1e-7, 20 * y_obj), tol_traj=max(1e-6, 20 * y_traj))
        print("%s: %d of 40 rows keep (status, iterations) under libm-level noise and were compared" % (which, n_stable))
        assert n_stable >= 30
