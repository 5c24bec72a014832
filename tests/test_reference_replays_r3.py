"""More of the reference's own solver tests, replayed (round 3): the LogDDP tests that the LogDDP work of this round had left
(tests/cddp_core/test_logddp_solver.cpp)

  * :87-152   WarmStartRollsOutReusedControlGuess -- a second initialize() with warm_start re-rolls the state trajectory from the
              reused controls (the overwritten states are only a guess), inf_pr = 0 without constraints;
  * :693-900  SolveQuadrotor -- the N = 400 figure-eight quadrotor (per-step reference states, u in [0, 4]^4) under LogDDP: converges,
              |q_N| = 1 +- 0.1, position error < 0.5; then a warm start from the solution that takes no more than cold + 20 iterations.

The assertions of the reference are made of the oracle on the CPU and of the product (pycddp facade -> cddp_hip_plugin_solve: GPU
backward passes on the (13, 4, 0) stack-fed sweep, host forward passes on the built-in quadrotor's host evaluation) on the GPU, where
the product is additionally held to the oracle's iteration count and objective.  The car-parking tests of CLDDP and LogDDP
(test_clddp_solver.cpp:373-568, test_logddp_solver.cpp:492-691) are replayed in tests/cpp/test_host_api.cpp, next to the IPDDP one."""
import importlib.util
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.host_arithmetic   # host route of the library: glibc on both sides (tests/conftest.py)

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
OK = ("OptimalSolutionFound", "AcceptableSolutionFound")


@pytest.fixture(scope="module")
def pycddp(api):
    name = "pycddp_amd"
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(name, os.path.join(REPO, "cddp-cpp_amd", "pycddp_amd.py"))
    mod = importlib.util.module_from_spec(spec); sys.modules[name] = mod; spec.loader.exec_module(mod)
    return mod


def test_logddp_warm_start_rolls_out_the_reused_control_guess(api, oracle_built):
    """test_logddp_solver.cpp:87-152."""
    o = api.default_options()
    p = api.Problem(api.SOLVER_LOGDDP, api.MODEL_PENDULUM, api.EULER, 2, 1, 4, 0.05, np.zeros((2, 2)), np.eye(1), np.eye(2), np.zeros(2),
                    model_params=[1.0, 1.0, 0.0, 9.81], options=o)
    x0 = np.array([np.pi, 0.2]); U = 0.1 * (np.arange(4) + 1.0).reshape(4, 1)
    Xg = np.zeros((5, 2)); Xg[0] = x0
    orc = api.Oracle(p); orc.set_initial(x0, U, Xg); orc.initialize()
    X42 = np.full((5, 2), 42.0); X42[0] = x0
    orc.set_initial(x0, U, X42)               # cddp_solver.X_[t] = 42 for t >= 1
    orc.set_warm_start(True); orc.initialize()
    X, Un = orc.trajectory()
    assert np.allclose(X[0], x0, rtol=0, atol=1e-12)
    x = x0.copy()
    for t in range(4):
        x = orc.dynamics(x, U[t])[1]
        assert np.max(np.abs(X[t + 1] - x)) < 1e-12
    assert orc.result()["inf_pr"] == 0.0


def _logddp_quadrotor(api):
    """test_logddp_solver.cpp:693-820: the problem of the IPDDP / CLDDP quadrotor tests with LogDDP's options."""
    p = api.quadrotor_figure8_problem(api.SOLVER_IPDDP)
    p.c.solver = api.SOLVER_LOGDDP
    o = p.options
    o.max_iterations = 100; o.tolerance = 1e-5; o.acceptable_tolerance = 1e-5; o.enable_parallel = 0
    o.logddp_mu_initial = 1e-1; o.logddp_relaxed_delta = 1e-5; o.logddp_mu_update_factor = 0.2; o.reg_initial_value = 1e-4
    p._rebuild()
    return p


def _hover(api, p):
    U0 = api.batch_U0(p, 1)[0]
    o = api.Oracle(p)
    X0 = np.zeros((p.N + 1, p.nx)); X0[0] = p.x0
    for i in range(p.N):
        X0[i + 1] = o.dynamics(X0[i], U0[i])[1]
    return U0, X0


def _asserts(p, status, iterations, X):
    assert status in OK, status
    assert iterations > 0
    assert abs(np.linalg.norm(X[-1, 3:7]) - 1.0) < 0.1
    assert np.linalg.norm(X[-1, :3] - p.x_ref[:3]) < 0.5


def test_oracle_passes_the_reference_logddp_quadrotor_test(api, oracle_built):
    p = _logddp_quadrotor(api)
    U0, X0 = _hover(api, p)
    o = api.Oracle(p, fast=True); o.set_initial(p.x0, U0, X0); r = o.solve()
    X, U = o.trajectory()
    _asserts(p, api.STATUS_STRINGS[int(r["status"])], r["iterations"], X)
    # warm start from the solution (:895-960): converges, iterations <= cold + 20
    pw = _logddp_quadrotor(api); pw.options.warm_start = 1; pw.options.max_iterations = 150; pw._rebuild()
    ow = api.Oracle(pw, fast=True); ow.set_warm_start(True); ow.set_initial(pw.x0, U, X); rw = ow.solve()
    assert api.STATUS_STRINGS[int(rw["status"])] in OK
    assert rw["iterations"] <= r["iterations"] + 20


@pytest.mark.gpu
def test_product_passes_the_reference_logddp_quadrotor_test(api, pycddp, oracle_built):
    p = _logddp_quadrotor(api)
    U0, X0 = _hover(api, p)
    o = pycddp.CDDPOptions(); o.verbose = False; o.print_solver_header = False
    o.max_iterations = 100; o.tolerance = 1e-5; o.acceptable_tolerance = 1e-5; o.regularization.initial_value = 1e-4
    o.log_barrier.barrier.mu_initial = 1e-1; o.log_barrier.relaxed_log_barrier_delta = 1e-5; o.log_barrier.barrier.mu_update_factor = 0.2
    dt, N = p.dt, p.N
    inertia = np.diag([7.782e-3, 7.782e-3, 1.439e-2])
    refs = [p.x_ref_traj[t].copy() for t in range(N + 1)]
    sv = pycddp.CDDP(p.x0, p.x_ref, N, dt, o)
    sv.set_dynamical_system(pycddp.Quadrotor(dt, 1.2, inertia, 0.165, "rk4"))
    sv.set_objective(pycddp.QuadraticObjective(p.Q, p.R, p.Qf, p.x_ref, refs, dt))
    sv.add_constraint("ControlConstraint", pycddp.ControlConstraint(np.zeros(4), 4.0 * np.ones(4)))
    sv.set_initial_trajectory(list(X0), list(U0))
    s = sv.solve(pycddp.SolverType.LogDDP)
    X = np.stack(s.state_trajectory)
    print("LogDDP quadrotor:", s.status_message, s.iterations_completed, s.final_objective)
    _asserts(p, s.status_message, s.iterations_completed, X)
    U = np.stack(s.control_trajectory)
    orc = api.Oracle(p, fast=False); orc.set_initial(p.x0, U0, X0); r = orc.solve()
    assert (s.status_message, s.iterations_completed) == (api.STATUS_STRINGS[int(r["status"])], int(r["iterations"]))
    assert abs(s.final_objective - r["final_objective"]) <= 1e-8 * max(1.0, abs(r["final_objective"]))
