"""Host plug-in solve (g1: north_star "keeps cddp-cpp's DynamicsModel / Constraint / Objective plugin surface").

`cddp_hip_plugin_solve` runs CDDP::solve() for arbitrary HOST DynamicalSystem / Objective / Constraint subclasses: batched
backward passes on the GPU (stack-fed sweeps), forward passes and outer loop on the host (csrc/plugin_solve.hip).  Checked here

  * against the CPU oracle, on problems the oracle also knows (pendulum, unicycle, LTI) re-implemented as PYTHON plug-ins --
    the plug-in path must reproduce the built-in path's decisions: status, iteration count, sweep / rollout counts, objective;
  * against the numpy twin on the reference's user-defined `QuadraticScalarSystem` (tests/cddp_core/test_ipddp_solver.cpp:291-346),
    a plant neither the device nor the oracle has;
  * by the reference's own Python tests restated (python/tests/test_custom_dynamics.py, test_nonlinear_objective.py) on the
    pycddp-compatible front end (cddp-cpp_amd/pycddp_amd.py).
"""
import importlib.util
import math
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.host_arithmetic   # host route of the library: glibc on both sides (tests/conftest.py)

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "oracle", "twin"))


@pytest.fixture(scope="module")
def pycddp(api):
    name = "pycddp_amd"
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(name, os.path.join(REPO, "cddp-cpp_amd", "pycddp_amd.py"))
    mod = importlib.util.module_from_spec(spec); sys.modules[name] = mod; spec.loader.exec_module(mod)
    return mod


# ---------------------------------------------------------------------------------------------- Python plug-ins
def make_plants(pycddp):
    class PyPendulum(pycddp.DynamicalSystem):     # pendulum.cpp:29-66 (+sin convention), analytic Jacobians
        def __init__(self, dt, length, mass, damping):
            super().__init__(2, 1, dt, "euler"); self.l, self.m, self.b, self.g = length, mass, damping, 9.81
        def get_continuous_dynamics(self, x, u, t=0.0):
            return np.array([x[1], (u[0] - self.b * x[1] + self.m * self.g * self.l * math.sin(x[0])) / (self.m * self.l * self.l)])
        def get_state_jacobian(self, x, u, t=0.0):
            return np.array([[0.0, 1.0], [(self.g / self.l) * math.cos(x[0]), -self.b / (self.m * self.l * self.l)]])
        def get_control_jacobian(self, x, u, t=0.0):
            return np.array([[0.0], [1.0 / (self.m * self.l * self.l)]])

    class PyUnicycle(pycddp.DynamicalSystem):     # unicycle.cpp:28-66
        def __init__(self, dt):
            super().__init__(3, 2, dt, "euler")
        def get_continuous_dynamics(self, x, u, t=0.0):
            return np.array([u[0] * math.cos(x[2]), u[0] * math.sin(x[2]), u[1]])
        def get_state_jacobian(self, x, u, t=0.0):
            A = np.zeros((3, 3)); A[0, 2] = -u[0] * math.sin(x[2]); A[1, 2] = u[0] * math.cos(x[2]); return A
        def get_control_jacobian(self, x, u, t=0.0):
            return np.array([[math.cos(x[2]), 0.0], [math.sin(x[2]), 0.0], [0.0, 1.0]])

    class QuadraticScalarSystem(pycddp.DynamicalSystem):   # tests/cddp_core/test_ipddp_solver.cpp:291-346
        def __init__(self):
            super().__init__(1, 1, 1.0, "euler")
        def get_discrete_dynamics(self, x, u, t=0.0): return np.array([x[0] + u[0] + 0.5 * x[0] * x[0]])
        def get_state_jacobian(self, x, u, t=0.0): return np.array([[1.0 + x[0]]])
        def get_control_jacobian(self, x, u, t=0.0): return np.eye(1)
        def get_state_hessian(self, x, u, t=0.0): return [np.eye(1)]
        def get_control_hessian(self, x, u, t=0.0): return [np.zeros((1, 1))]
        def get_cross_hessian(self, x, u, t=0.0): return [np.zeros((1, 1))]

    return PyPendulum, PyUnicycle, QuadraticScalarSystem


def _options(pycddp, **kw):
    o = pycddp.CDDPOptions(); o.verbose = False; o.print_solver_header = False
    for k, v in kw.items():
        setattr(o, k, v)
    return o


# ---------------------------------------------------------------------------------------------- CPU: surface
def test_python_plugin_surface(pycddp):
    """python/tests/test_custom_dynamics.py:52-75 + the NonlinearObjective finite-difference defaults (objective.cpp:188-288,
    helper.hpp:34-209) against closed forms."""
    class DoubleIntegrator(pycddp.DynamicalSystem):
        def __init__(self, dt): super().__init__(2, 1, dt, "euler")
        def get_continuous_dynamics(self, state, control, time=0.0): return np.array([state[1], control[0]])
    s = DoubleIntegrator(0.1)
    assert s.state_dim == 2 and s.control_dim == 1
    np.testing.assert_allclose(s.get_continuous_dynamics(np.array([0.0, 1.0]), np.array([0.5])), [1.0, 0.5])
    np.testing.assert_allclose(s.get_discrete_dynamics(np.array([0.0, 1.0]), np.array([0.5])), [0.1, 1.05], atol=1e-10)
    with pytest.raises(RuntimeError, match="do not support getContinuousDynamicsAutodiff"):
        s.get_state_jacobian(np.zeros(2), np.zeros(1))

    class Obj(pycddp.NonlinearObjective):
        def running_cost(self, x, u, index): return float(x @ x + 0.1 * u @ u + 0.3 * x[0] * u[0])
        def terminal_cost(self, x): return float(10.0 * x @ x)
    ob = Obj(0.1)
    x = np.array([0.7, -0.4]); u = np.array([0.25])
    np.testing.assert_allclose(ob.get_running_cost_state_gradient(x, u, 0), [2 * 0.7 + 0.3 * 0.25, -0.8], atol=1e-8)
    np.testing.assert_allclose(ob.get_running_cost_control_gradient(x, u, 0), [0.05 + 0.21], atol=1e-8)
    np.testing.assert_allclose(ob.get_running_cost_state_hessian(x, u, 0), 2 * np.eye(2), atol=1e-4)
    np.testing.assert_allclose(ob.get_running_cost_control_hessian(x, u, 0), [[0.2]], atol=1e-4)
    np.testing.assert_allclose(ob.get_final_cost_gradient(x), 20 * x, atol=1e-7)
    np.testing.assert_allclose(ob.get_final_cost_hessian(x), 20 * np.eye(2), atol=1e-3)
    assert ob.evaluate([x, x, x], [u, u]) == pytest.approx(2 * ob.running_cost(x, u, 0) + ob.terminal_cost(x))
    q = pycddp.QuadraticObjective(np.eye(2), 0.1 * np.eye(1), 10 * np.eye(2), np.zeros(2), [], 0.1)
    np.testing.assert_allclose(q.get_running_cost_state_gradient(x, u, 0), _fd(lambda s_: q.running_cost(s_, u, 0), x), atol=1e-7)


def _fd(f, x, h=1e-6):
    g = np.zeros(x.size)
    for i in range(x.size):
        e = np.zeros(x.size); e[i] = h
        g[i] = (f(x + e) - f(x - e)) / (2 * h)
    return g


# ---------------------------------------------------------------------------------------------- GPU: plug-in path vs oracle
def _solve_plugin(pycddp, system, Q, R, Qf, goal, N, dt, opts, cons, x0s, solver, U0=None):
    sv = pycddp.CDDP(x0s[0], goal, N, dt, opts)
    sv.set_dynamical_system(system)
    sv.set_objective(pycddp.QuadraticObjective(Q, R, Qf, goal, [], dt))
    for name, c in cons:
        sv.add_constraint(name, c)
    if U0 is not None:
        sv.set_initial_trajectory([x0s[0]] * (N + 1), [U0] * N)
    assert sv._needs_host_plugins()
    return sv.solve_batch(x0s, solver)


@pytest.mark.gpu
@pytest.mark.parametrize("solver", ["IPDDP", "CLDDP"])
def test_python_pendulum_plugin_matches_the_oracle(api, pycddp, oracle_built, solver):
    """examples/cddp_pendulum.cpp:24-65 with the plant re-implemented as a Python plug-in: the host plug-in solve must make the
    oracle's decisions (and therefore the built-in device path's) on every trajectory -- status, iterations, sweep and rollout
    counts -- and reach its objective / trajectory."""
    PyPendulum, _, _ = make_plants(pycddp)
    sk = api.SOLVER_IPDDP if solver == "IPDDP" else api.SOLVER_CLDDP
    p = api.pendulum_problem(sk, True)
    B = 6
    x0 = api.batch_x0(p, B, 20260929, [0.1, 0.1])
    o = _options(pycddp, max_iterations=30, tolerance=1e-4, acceptable_tolerance=1e-5)
    o.regularization.initial_value = 1e-6
    sols = _solve_plugin(pycddp, PyPendulum(0.02, 0.5, 1.0, 0.01), np.zeros((2, 2)), 0.1 * np.eye(1), 100.0 * np.eye(2), np.zeros(2), 100, 0.02, o,
                         [("ControlConstraint", pycddp.ControlConstraint(np.array([-20.0]), np.array([20.0])))], list(x0), pycddp.SolverType[solver])
    ores, oX, oU, oK, _ = api.oracle_solve_batch(p, x0, None, None, n_threads=B)
    for b in range(B):
        s = sols[b]
        assert s.status_message == api.STATUS_STRINGS[int(ores["status"][b])], (b, s.status_message)
        assert s.iterations_completed == ores["iterations"][b], (b, s.iterations_completed, ores["iterations"][b])
        assert abs(s.final_objective - ores["final_objective"][b]) <= 1e-9 * max(1.0, abs(ores["final_objective"][b]))
        assert np.max(np.abs(np.stack(s.state_trajectory) - oX[b])) < 1e-8 and np.max(np.abs(np.stack(s.control_trajectory) - oU[b])) < 1e-7
        assert np.max(np.abs(np.stack(s.feedback_gains) - oK[b]) / np.maximum(1.0, np.abs(oK[b]))) < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("solver", ["IPDDP", "CLDDP"])
def test_plugin_host_threads_do_not_change_results(api, pycddp, solver):
    """cddp_hip_plugin_set_host_threads (round 6): the per-trajectory host work of the plug-in solve -- derivative fill, forward passes, updates
    -- on four threads (the Python callbacks then arrive from worker threads and serialise on the interpreter lock) gives the bits of one thread."""
    PyPendulum, _, _ = make_plants(pycddp)
    p = api.pendulum_problem(api.SOLVER_IPDDP if solver == "IPDDP" else api.SOLVER_CLDDP, True)
    B = 9
    x0 = api.batch_x0(p, B, 20260930, [0.1, 0.1])
    lib = api.load_hip()

    def run():
        o = _options(pycddp, max_iterations=25, tolerance=1e-4, acceptable_tolerance=1e-5)
        o.regularization.initial_value = 1e-6
        return _solve_plugin(pycddp, PyPendulum(0.02, 0.5, 1.0, 0.01), np.zeros((2, 2)), 0.1 * np.eye(1), 100.0 * np.eye(2), np.zeros(2), 100, 0.02, o,
                             [("ControlConstraint", pycddp.ControlConstraint(np.array([-20.0]), np.array([20.0])))], list(x0), pycddp.SolverType[solver])
    try:
        assert lib.cddp_hip_plugin_set_host_threads(1) == 0
        one = run()
        assert lib.cddp_hip_plugin_set_host_threads(4) == 0
        four = run()
    finally:
        lib.cddp_hip_plugin_set_host_threads(1)
    assert lib.cddp_hip_plugin_set_host_threads(-1) != 0
    for a, b in zip(one, four):
        assert a.status_message == b.status_message and a.iterations_completed == b.iterations_completed and a.final_objective == b.final_objective
        assert np.array_equal(np.stack(a.state_trajectory), np.stack(b.state_trajectory)) and np.array_equal(np.stack(a.control_trajectory), np.stack(b.control_trajectory))
        assert np.array_equal(np.stack(a.feedback_gains), np.stack(b.feedback_gains))


@pytest.mark.gpu
def test_c_plugin_matches_the_oracle_on_many_threads(api, oracle_built, tmp_path):
    """A user plug-in written in C (tests/cpp/pendulum_plugin.c: the plant, objective and control box of examples/cddp_pendulum.cpp as callbacks),
    solved with the batch's host work on eight threads: every trajectory makes the oracle's decisions (the callbacks are the built-in pendulum's
    arithmetic), and cddp_hip_plugin_last_stats accounts for the call."""
    import ctypes as C
    import subprocess
    so = str(tmp_path / "pendulum_plugin.so")
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-o", so, os.path.join(REPO, "tests", "cpp", "pendulum_plugin.c"), "-lm"])
    pl = C.CDLL(so)

    class Params(C.Structure):
        _fields_ = [(n, C.c_double) for n in ("dt", "length", "mass", "damping", "gravity", "Qf", "R", "umax")]
    prm = Params(0.02, 0.5, 1.0, 0.01, 9.81, 100.0, 0.1, 20.0)
    p = api.pendulum_problem(api.SOLVER_IPDDP, True)
    lib = api.load_hip()
    ps = api.PluginStruct()
    ps.abi_version = api.ABI_VERSION; ps.options_bytes = C.sizeof(api.Options); ps.user = C.cast(C.pointer(prm), C.c_void_p)
    ps.nx, ps.nu, ps.n_constraints = 2, 1, 1
    ps.constraint_dims[0] = 2
    for field, sym, ftype in (("discrete_dynamics", "pend_dynamics", api._F_DYN), ("jacobians", "pend_jacobians", api._F_JAC), ("running_cost", "pend_running_cost", api._F_RC),
                              ("terminal_cost", "pend_terminal_cost", api._F_TC), ("running_cost_derivatives", "pend_running_cost_derivatives", api._F_RCD),
                              ("terminal_cost_derivatives", "pend_terminal_cost_derivatives", api._F_TCD), ("constraints", "pend_constraints", api._F_CON)):
        setattr(ps, field, C.cast(getattr(pl, sym), ftype))
    B, N = 40, 100
    x0 = api.batch_x0(p, B, 20260932, [0.1, 0.1])
    res = np.zeros(B, dtype=api.RESULT_DTYPE); X = np.zeros((B, N + 1, 2)); U = np.zeros((B, N, 1))
    try:
        assert lib.cddp_hip_plugin_set_host_threads(8) == 0
        rc = lib.cddp_hip_plugin_solve(C.byref(ps), int(api.SOLVER_IPDDP), N, C.c_double(0.02), C.byref(p.options), 0, B, api._ptr(x0), None, None,
                                       res.ctypes.data_as(C.c_void_p), api._ptr(X), api._ptr(U), None)
    finally:
        lib.cddp_hip_plugin_set_host_threads(1)
    assert rc == 0, lib.cddp_hip_last_error().decode()
    tot, gpu, ker = C.c_double(), C.c_double(), C.c_double(); sw, th = C.c_int(), C.c_int()
    assert lib.cddp_hip_plugin_last_stats(C.byref(tot), C.byref(gpu), C.byref(ker), C.byref(sw), C.byref(th)) == 0
    assert th.value == 8 and sw.value >= int(res["iterations"].max()) and 0.0 < ker.value <= gpu.value <= tot.value
    ores, oX, oU, _, _ = api.oracle_solve_batch(p, x0, None, None, n_threads=8)
    assert np.array_equal(res["iterations"], ores["iterations"]) and np.array_equal(res["status"], ores["status"])
    assert np.array_equal(res["n_backward"], ores["n_backward"]) and np.array_equal(res["n_forward"], ores["n_forward"])
    assert np.max(np.abs(res["final_objective"] - ores["final_objective"]) / np.maximum(1.0, np.abs(ores["final_objective"]))) < 1e-9
    assert np.max(np.abs(X - oX)) < 1e-8 and np.max(np.abs(U - oU)) < 1e-7


@pytest.mark.gpu
def test_python_unicycle_plugin_with_box_and_ball_matches_the_oracle(api, pycddp, oracle_built):
    """nx = 3, nu = 2, two constraint objects (control box 'control_limits' + ball 'obstacle': m = 5, one row reads x): the stacking
    order, the constraint-major merit / violation sums and the G_x terms of the plug-in path against the oracle."""
    _, PyUnicycle, _ = make_plants(pycddp)
    p = api.unicycle_problem(api.SOLVER_IPDDP, 100, True)
    B = 4
    x0 = api.batch_x0(p, B, 20260929, [0.05, 0.05, 0.05])
    o = _options(pycddp, max_iterations=100, tolerance=1e-4, acceptable_tolerance=1e-6)
    sols = _solve_plugin(pycddp, PyUnicycle(0.03), np.zeros((3, 3)), 0.05 * np.eye(2), np.diag([100.0, 100.0, 50.0]), np.array([2.0, 2.0, math.pi / 2]), 100, 0.03, o,
                         [("control_limits", pycddp.ControlConstraint(np.array([-1.1, -math.pi]), np.array([1.1, math.pi]))),
                          ("obstacle", pycddp.BallConstraint(0.4, np.array([1.0, 1.0])))], list(x0), pycddp.SolverType.IPDDP, U0=np.array([0.5, 0.1]))
    U0 = api.batch_U0(p, B)
    ores, oX, oU, _, _ = api.oracle_solve_batch(p, x0, U0, None, n_threads=B)
    for b in range(B):
        s = sols[b]
        assert s.status_message == api.STATUS_STRINGS[int(ores["status"][b])] and s.iterations_completed == ores["iterations"][b], (b, s.status_message, s.iterations_completed, ores["iterations"][b])
        assert abs(s.final_objective - ores["final_objective"][b]) <= 1e-8 * max(1.0, abs(ores["final_objective"][b]))
        assert np.max(np.abs(np.stack(s.state_trajectory) - oX[b])) < 1e-6


@pytest.mark.gpu
def test_quadratic_scalar_system_against_the_twin(api, pycddp):
    """The reference's user-defined QuadraticScalarSystem (a plant with its own getDiscreteDynamics and Jacobians that are NOT the
    derivative of it): IPDDP with a control box, Gauss-Newton and full DDP (use_ilqr = false: its Hessian callbacks), against the
    numpy twin driven by the same plug-in -- iteration count, status, objective, trajectory."""
    import cddp_twin as T
    _, _, QSS = make_plants(pycddp)

    class TwinQSS:
        nx, nu, discrete = 1, 1, True
        def step(self, x, u, t): return np.array([x[0] + u[0] + 0.5 * x[0] * x[0]])
        def jac(self, x, u, t): return np.array([[1.0 + x[0]]]), np.eye(1)
        def hess(self, x, u, t): return np.ones((1, 1, 1)), np.zeros((1, 1, 1)), np.zeros((1, 1, 1))

    for use_ilqr in (True, False):
        N, dt = 8, 1.0
        o = _options(pycddp, max_iterations=40, tolerance=1e-6, acceptable_tolerance=1e-6, use_ilqr=use_ilqr)
        o.regularization.initial_value = 1e-6; o.ipddp.barrier.mu_initial = 1e-1
        sv = pycddp.CDDP(np.array([0.3]), np.zeros(1), N, dt, o)
        sv.set_dynamical_system(QSS())
        sv.set_objective(pycddp.QuadraticObjective(np.eye(1), 0.1 * np.eye(1), 10.0 * np.eye(1), np.zeros(1), [], dt))
        sv.add_constraint("ControlConstraint", pycddp.ControlConstraint(np.array([-0.5]), np.array([0.5])))
        sol = sv.solve(pycddp.SolverType.IPDDP)
        tw = T.Twin(dict(solver="IPDDP", model=TwinQSS(), integrator="euler", dt=dt, N=N, Q=np.eye(1), R=0.1 * np.eye(1), Qf=10.0 * np.eye(1), xref=[0.0],
                         constraints={"ControlConstraint": T.ControlBox([-0.5], [0.5])},
                         options=dict(max_iterations=40, tolerance=1e-6, acceptable_tolerance=1e-6, reg_initial_value=1e-6, mu_initial=1e-1, use_ilqr=use_ilqr)))
        tw.set_initial(np.array([0.3]), None)
        r = tw.solve()
        assert sol.status_message == T.STATUS[r["status"]] and sol.iterations_completed == r["iterations"], (use_ilqr, sol.status_message, sol.iterations_completed, r)
        assert abs(sol.final_objective - r["final_objective"]) < 1e-9 * max(1.0, abs(r["final_objective"]))
        assert np.max(np.abs(np.stack(sol.state_trajectory) - tw.X)) < 1e-8 and np.max(np.abs(np.stack(sol.control_trajectory) - tw.U)) < 1e-8


# ---------------------------------------------------------------------------------------------- GPU: the reference's Python tests
@pytest.mark.gpu
def test_custom_dynamics_with_solver(pycddp):
    """python/tests/test_custom_dynamics.py:78-113: a Python double integrator through CLDDP; the box is named "ctrl", so CLDDP
    runs unbounded (clddp_solver.cpp:85-86) exactly as in the reference."""
    class DoubleIntegrator(pycddp.DynamicalSystem):
        def __init__(self, dt): super().__init__(2, 1, dt, "euler")
        def get_continuous_dynamics(self, state, control, time=0.0): return np.array([state[1], control[0]])
        def get_state_jacobian(self, state, control, time=0.0): return np.array([[0.0, 1.0], [0.0, 0.0]])
        def get_control_jacobian(self, state, control, time=0.0): return np.array([[0.0], [1.0]])
    dt, horizon = 0.1, 20
    opts = _options(pycddp, max_iterations=30, enable_parallel=True, num_threads=2)
    solver = pycddp.CDDP(np.array([1.0, 0.0]), np.zeros(2), horizon, dt, opts)
    solver.set_dynamical_system(DoubleIntegrator(dt))
    solver.set_objective(pycddp.QuadraticObjective(np.zeros((2, 2)), 0.1 * np.eye(1), 10.0 * np.eye(2), np.zeros(2), [], dt))
    solver.add_constraint("ctrl", pycddp.ControlConstraint(np.array([-5.0]), np.array([5.0])))
    solution = solver.solve(pycddp.SolverType.CLDDP)
    assert solution.solver_name == "CLDDP" and solution.status_message and solution.iterations_completed >= 0
    assert len(solution.time_points) == horizon + 1 and len(solution.state_trajectory) == horizon + 1
    assert len(solution.control_trajectory) == horizon and len(solution.feedback_gains) == horizon
    assert np.isfinite(solution.final_objective) and np.isfinite(solution.final_step_length) and np.isfinite(solution.final_regularization)
    assert solution.solve_time_ms >= 0
    # LQ problem: CLDDP converges to the Riccati optimum, and a named "ControlConstraint" clamps where "ctrl" did not
    assert solution.status_message in ("OptimalSolutionFound", "AcceptableSolutionFound")
    assert np.linalg.norm(solution.state_trajectory[-1]) < 0.2
    solver2 = pycddp.CDDP(np.array([1.0, 0.0]), np.zeros(2), horizon, dt, opts)
    solver2.set_dynamical_system(DoubleIntegrator(dt))
    solver2.set_objective(pycddp.QuadraticObjective(np.zeros((2, 2)), 0.1 * np.eye(1), 10.0 * np.eye(2), np.zeros(2), [], dt))
    solver2.add_constraint("ControlConstraint", pycddp.ControlConstraint(np.array([-0.3]), np.array([0.3])))
    s2 = solver2.solve(pycddp.SolverType.CLDDP)
    assert np.max(np.abs(np.stack(s2.control_trajectory))) <= 0.3 + 1e-12 and np.max(np.abs(np.stack(solution.control_trajectory))) > 0.3


@pytest.mark.gpu
def test_python_callback_exceptions_surface_to_python(pycddp):
    """python/tests/test_custom_dynamics.py:116-139: an exception raised inside a Python callback reaches the caller."""
    class Exploding(pycddp.DynamicalSystem):
        def __init__(self, dt): super().__init__(2, 1, dt, "euler"); self.calls = 0
        def get_continuous_dynamics(self, state, control, time=0.0): return np.array([state[1], control[0]])
        def get_state_jacobian(self, state, control, time=0.0): return np.array([[0.0, 1.0], [0.0, 0.0]])
        def get_control_jacobian(self, state, control, time=0.0): return np.array([[0.0], [1.0]])
        def get_discrete_dynamics(self, state, control, time=0.0):
            self.calls += 1
            if self.calls > 70:
                raise RuntimeError("boom from Python dynamics")
            return state + self.timestep * self.get_continuous_dynamics(state, control, time)
    dt = 0.1
    solver = pycddp.CDDP(np.array([1.0, 0.0]), np.zeros(2), 60, dt, _options(pycddp, max_iterations=2))
    solver.set_dynamical_system(Exploding(dt))
    solver.set_objective(pycddp.QuadraticObjective(np.eye(2), 0.1 * np.eye(1), 10.0 * np.eye(2), np.zeros(2), [], dt))
    solver.add_constraint("ctrl", pycddp.ControlConstraint(np.array([-5.0]), np.array([5.0])))
    with pytest.raises(RuntimeError, match="boom from Python dynamics"):
        solver.solve(pycddp.SolverType.IPDDP)


@pytest.mark.gpu
def test_python_dynamics_autodiff_path_raises_clear_error(pycddp):
    """python/tests/test_custom_dynamics.py:142-165: a Python plant without Jacobians cannot fall back to autodiff."""
    class Minimal(pycddp.DynamicalSystem):
        def __init__(self, dt): super().__init__(2, 1, dt, "euler")
        def get_continuous_dynamics(self, state, control, time=0.0): return np.array([state[1], control[0]])
    dt = 0.1
    solver = pycddp.CDDP(np.array([1.0, 0.0]), np.zeros(2), 8, dt, _options(pycddp, max_iterations=2))
    solver.set_dynamical_system(Minimal(dt))
    solver.set_objective(pycddp.QuadraticObjective(np.eye(2), 0.1 * np.eye(1), 10.0 * np.eye(2), np.zeros(2), [], dt))
    solver.add_constraint("ctrl", pycddp.ControlConstraint(np.array([-5.0]), np.array([5.0])))
    with pytest.raises(RuntimeError, match="do not support getContinuousDynamicsAutodiff"):
        solver.solve(pycddp.SolverType.CLDDP)


@pytest.mark.gpu
@pytest.mark.parametrize("solver", ["CLDDP", "IPDDP"])
def test_python_nonlinear_objective_dispatches_through_solver(api, pycddp, oracle_built, solver):
    """python/tests/test_nonlinear_objective.py:30-62 (the reference drives it through LogDDP; here through the two device cores): a
    Python NonlinearObjective with finite-difference derivatives on the built-in LTISystem.  The cost is quadratic, so the result
    must agree with the same problem posed with QuadraticObjective on the built-in device path (1e-5: FD derivatives)."""
    class Counting(pycddp.NonlinearObjective):
        def __init__(self, timestep, counters): super().__init__(timestep); self._c = counters
        def evaluate(self, states, controls):
            self._c["evaluate"] += 1
            return super().evaluate(states, controls)
        def running_cost(self, state, control, index):
            self._c["running_cost"] += 1
            return float(state @ state + 0.1 * control @ control)
        def terminal_cost(self, final_state):
            self._c["terminal_cost"] += 1
            return float(10.0 * final_state @ final_state)
    dt, horizon = 0.1, 15
    counters = {"evaluate": 0, "running_cost": 0, "terminal_cost": 0}
    A = np.eye(2) + dt * np.array([[0.0, 1.0], [0.0, 0.0]]); Bm = dt * np.array([[0.0], [1.0]])
    opts = _options(pycddp, max_iterations=20)

    def build(obj):
        sv = pycddp.CDDP(np.array([1.0, 0.0]), np.zeros(2), horizon, dt, opts)
        sv.set_dynamical_system(pycddp.LTISystem(A, Bm, dt))
        sv.set_objective(obj)
        sv.add_constraint("ControlConstraint", pycddp.ControlConstraint(np.array([-2.0]), np.array([2.0])))
        return sv
    sol = build(Counting(dt, counters)).solve(pycddp.SolverType[solver])
    assert sol.solver_name == solver and sol.status_message
    assert counters["running_cost"] > 0 and counters["terminal_cost"] > 0
    ref = build(pycddp.QuadraticObjective(np.eye(2) / dt, 0.1 * np.eye(1) / dt, 10.0 * np.eye(2), np.zeros(2), [], dt)).solve(pycddp.SolverType[solver])
    assert abs(sol.final_objective - ref.final_objective) < 1e-5 * max(1.0, abs(ref.final_objective)), (sol.final_objective, ref.final_objective)
    assert np.max(np.abs(np.stack(sol.state_trajectory) - np.stack(ref.state_trajectory))) < 1e-3


def _twin_plants():
    class TwinQSS:        # tests/cddp_core/test_ipddp_solver.cpp:291-346
        nx, nu, discrete = 1, 1, True
        def step(self, x, u, t): return np.array([x[0] + u[0] + 0.5 * x[0] * x[0]])
        def jac(self, x, u, t): return np.array([[1.0 + x[0]]]), np.eye(1)
        def hess(self, x, u, t): return np.ones((1, 1, 1)), np.zeros((1, 1, 1)), np.zeros((1, 1, 1))

    class TwinDI:         # double integrator, python/tests/test_custom_dynamics.py
        nx, nu, discrete = 2, 1, False
        def f(self, x, u, t): return np.array([x[1], u[0]])
        def jac(self, x, u, t): return np.array([[0.0, 1.0], [0.0, 0.0]]), np.array([[0.0], [1.0]])
        def hess(self, x, u, t): return np.zeros((2, 2, 2)), np.zeros((2, 1, 1)), np.zeros((2, 1, 2))
    return TwinQSS, TwinDI


TERMINAL_PLUGIN_CASES = {
    # name: (plant, N, dt, x0, Q, R, Qf, path box or None, terminal spec {name: ("eq", target) | ("ineq", A, b)}, use_ilqr)
    "qss_term_eq_ilqr": ("qss", 8, 1.0, [1.0], 0.0, 1e-2, 0.0, None, {"TerminalTarget": ("eq", [0.0])}, True),
    "qss_term_eq_ddp": ("qss", 8, 1.0, [1.0], 0.0, 1e-2, 0.0, None, {"TerminalTarget": ("eq", [0.0])}, False),      # test_ipddp_solver.cpp:1512-1578 as a solve
    "qss_box_term_eq": ("qss", 8, 1.0, [0.6], 0.1, 1e-1, 0.0, (-0.5, 0.5), {"TerminalTarget": ("eq", [0.0])}, True),
    "qss_term_ineq": ("qss", 8, 1.0, [0.5], 1.0, 1e-1, 1.0, None, {"TerminalBand": ("ineq", [[1.0], [-1.0]], [0.05, 0.05])}, True),
    "di_box_term_eq": ("di", 12, 0.1, [1.0, 0.0], 1.0, 0.1, 1.0, (-8.0, 8.0), {"T": ("eq", [0.0, 0.0])}, True),
    "di_term_eq_and_ineq": ("di", 12, 0.1, [1.0, 0.0], 1.0, 0.1, 1.0, None, {"A_ineq": ("ineq", [[0.0, 1.0]], [0.4]), "B_eq": ("eq", [0.2, 0.0])}, True),
    "di_box_term_ineq": ("di", 12, 0.1, [1.0, 0.0], 1.0, 0.1, 10.0, (-6.0, 6.0), {"T": ("ineq", [[1.0, 0.0], [0.0, 1.0]], [0.3, 0.3])}, True),
}


@pytest.mark.gpu
@pytest.mark.parametrize("case", list(TERMINAL_PLUGIN_CASES))
def test_plugin_solve_with_terminal_constraints_against_the_twin(api, pycddp, case):
    """user plants with TERMINAL constraints on the plug-in route (cddp_hip_plugin_solve_terminal) -- the reference pairs
    its user-defined QuadraticScalarSystem with a TerminalEqualityConstraint and use_ilqr = false (tests/cddp_core/test_ipddp_solver.cpp:292-346,
    1512-1578).  Terminal equality (reduced LQR on the GPU: CDDP_HIP_STACKS_IPDDP_TERM_EQ), terminal inequality (barrier terms in the terminal
    value), both, with and without a path box, Gauss-Newton and full DDP: status, iterations, objective, trajectory against the numpy twin."""
    import cddp_twin as T
    plant, N, dt, x0, Qs, Rs, Qfs, box, term, use_ilqr = TERMINAL_PLUGIN_CASES[case]
    _, _, QSS = make_plants(pycddp)
    TwinQSS, TwinDI = _twin_plants()

    class DI(pycddp.DynamicalSystem):
        def __init__(self): super().__init__(2, 1, 0.1, "euler")
        def get_continuous_dynamics(self, s, c, t=0.0): return np.array([s[1], c[0]])
        def get_state_jacobian(self, s, c, t=0.0): return np.array([[0.0, 1.0], [0.0, 0.0]])
        def get_control_jacobian(self, s, c, t=0.0): return np.array([[0.0], [1.0]])
        def get_state_hessian(self, s, c, t=0.0): return [np.zeros((2, 2)), np.zeros((2, 2))]
        def get_control_hessian(self, s, c, t=0.0): return [np.zeros((1, 1)), np.zeros((1, 1))]
        def get_cross_hessian(self, s, c, t=0.0): return [np.zeros((1, 2)), np.zeros((1, 2))]
    nx = 1 if plant == "qss" else 2
    okw = dict(max_iterations=40, tolerance=1e-6, acceptable_tolerance=1e-6, use_ilqr=use_ilqr)
    o = _options(pycddp, **okw)
    o.regularization.initial_value = 1e-6; o.ipddp.barrier.mu_initial = 1e-1
    x0 = np.array(x0, float); goal = np.zeros(nx)
    sv = pycddp.CDDP(x0, goal, N, dt, o)
    sv.set_dynamical_system(QSS() if plant == "qss" else DI())
    sv.set_objective(pycddp.QuadraticObjective(Qs * np.eye(nx), Rs * np.eye(1), Qfs * np.eye(nx), goal, [], dt))
    cons = {}
    if box is not None:
        sv.add_constraint("ControlConstraint", pycddp.ControlConstraint(np.array([box[0]]), np.array([box[1]])))
        cons["ControlConstraint"] = T.ControlBox([box[0]], [box[1]])
    for name, spec in term.items():
        sv.add_terminal_constraint(name, pycddp.TerminalEqualityConstraint(np.array(spec[1], float)) if spec[0] == "eq"
                                   else pycddp.TerminalInequalityConstraint(np.array(spec[1], float), np.array(spec[2], float)))
    assert sv._needs_host_plugins()
    sol = sv.solve(pycddp.SolverType.IPDDP)
    assert sol.route == "plugin"
    tw = T.Twin(dict(solver="IPDDP", model=TwinQSS() if plant == "qss" else TwinDI(), integrator="euler", dt=dt, N=N, Q=Qs * np.eye(nx), R=Rs * np.eye(1),
                     Qf=Qfs * np.eye(nx), xref=list(goal), constraints=cons, terminal=term,
                     options=dict(max_iterations=40, tolerance=1e-6, acceptable_tolerance=1e-6, reg_initial_value=1e-6, mu_initial=1e-1, use_ilqr=use_ilqr)))
    tw.set_initial(x0, None)
    r = tw.solve()
    assert sol.status_message == T.STATUS[r["status"]] and sol.iterations_completed == r["iterations"], (case, sol.status_message, sol.iterations_completed, r["status"], r["iterations"])
    assert abs(sol.final_objective - r["final_objective"]) < 1e-8 * max(1.0, abs(r["final_objective"]))
    assert np.max(np.abs(np.stack(sol.state_trajectory) - tw.X)) < 1e-7 and np.max(np.abs(np.stack(sol.control_trajectory) - tw.U)) < 1e-7
    # the terminal set did its work: equality rows closed to the solver's tolerance class, inequality rows satisfied
    xN = np.stack(sol.state_trajectory)[-1]
    for name, spec in term.items():
        if spec[0] == "eq" and sol.status_message in ("OptimalSolutionFound", "AcceptableSolutionFound"):
            assert np.max(np.abs(xN - np.array(spec[1]))) < 1e-3, (case, xN)
        if spec[0] == "ineq" and sol.status_message in ("OptimalSolutionFound", "AcceptableSolutionFound"):
            assert np.all(np.array(spec[1]) @ xN - np.array(spec[2]) < 1e-6), (case, xN)


@pytest.mark.gpu
def test_plugin_full_ddp_needs_hessian_virtuals(api, pycddp):
    class DI(pycddp.DynamicalSystem):
        def __init__(self): super().__init__(2, 1, 0.1, "euler")
        def get_continuous_dynamics(self, s, c, t=0.0): return np.array([s[1], c[0]])
        def get_state_jacobian(self, s, c, t=0.0): return np.array([[0.0, 1.0], [0.0, 0.0]])
        def get_control_jacobian(self, s, c, t=0.0): return np.array([[0.0], [1.0]])
    sv = pycddp.CDDP(np.array([1.0, 0.0]), np.zeros(2), 8, 0.1, _options(pycddp, max_iterations=3, use_ilqr=False))
    sv.set_dynamical_system(DI())
    sv.set_objective(pycddp.QuadraticObjective(np.eye(2), 0.1 * np.eye(1), 10.0 * np.eye(2), np.zeros(2), [], 0.1))
    with pytest.raises(RuntimeError, match="do not support getContinuousDynamicsAutodiff"):   # full DDP needs the Hessian callbacks
        sv.solve(pycddp.SolverType.IPDDP)


@pytest.mark.gpu
@pytest.mark.parametrize("with_box", [True, False])
def test_plugin_warm_start_with_provided_trajectory_matches_the_oracle(api, pycddp, oracle_built, with_box):
    """options.warm_start on the plug-in route (round 6): a stateless call takes the reference's "warm start with provided trajectory"
    branch (ipddp_solver.cpp:733-816: barrier parameter from the seed's largest constraint value, duals initialised from the seed).  The Python
    pendulum plug-in seeded with a control guess against a NEW oracle object started the same way."""
    PyPendulum, _, _ = make_plants(pycddp)
    p = api.pendulum_problem(api.SOLVER_IPDDP, with_box)
    p.options.warm_start = 1; p.options.max_iterations = 30; p.options.tolerance = 1e-4; p.options.acceptable_tolerance = 1e-5; p.options.reg_initial_value = 1e-6
    B = 4
    x0 = api.batch_x0(p, B, 20260931, [0.1, 0.1])
    rng = np.random.default_rng(5)
    U0 = 2.0 * rng.standard_normal((100, 1))
    o = _options(pycddp, max_iterations=30, tolerance=1e-4, acceptable_tolerance=1e-5, warm_start=True)
    o.regularization.initial_value = 1e-6
    cons = [("ControlConstraint", pycddp.ControlConstraint(np.array([-20.0]), np.array([20.0])))] if with_box else []
    sv = pycddp.CDDP(x0[0], np.zeros(2), 100, 0.02, o)
    sv.set_dynamical_system(PyPendulum(0.02, 0.5, 1.0, 0.01))
    sv.set_objective(pycddp.QuadraticObjective(np.zeros((2, 2)), 0.1 * np.eye(1), 100.0 * np.eye(2), np.zeros(2), [], 0.02))
    for name, c in cons:
        sv.add_constraint(name, c)
    sv.set_initial_trajectory([x0[0]] * 101, [U0[t] for t in range(100)])
    sols = sv.solve_batch(list(x0), pycddp.SolverType.IPDDP)
    for b in range(B):
        orc = api.Oracle(p); orc.set_warm_start(True); orc.set_initial(x0[b], U0, None); q = orc.solve()
        s = sols[b]
        assert s.status_message == api.STATUS_STRINGS[int(q["status"])] and s.iterations_completed == q["iterations"], (b, s.status_message, s.iterations_completed, q["status"], q["iterations"])
        assert abs(s.final_objective - q["final_objective"]) <= 1e-8 * max(1.0, abs(q["final_objective"]))


@pytest.mark.gpu
@pytest.mark.parametrize("plant", ["pendulum", "cartpole"])
def test_builtin_plant_with_python_objective(api, pycddp, plant):
    """The shape of the reference's car-parking test (tests/cddp_core/test_ipddp_solver.cpp:628-885): a BUILT-IN plant with a
    user-defined NonlinearObjective.  The plant's host virtuals evaluate the kernels' model source compiled for the host
    (cddp_hip_model_eval); the objective is the device path's quadratic written as a Python cost with the reference's
    finite-difference derivative defaults, so the plug-in solve must land on the device path's solution."""
    if plant == "pendulum":
        dt, N, mk = 0.02, 100, lambda: pycddp.Pendulum(0.02, 0.5, 1.0, 0.01, "euler")
        x0, goal, Q, R, Qf, lim = np.array([np.pi, 0.0]), np.zeros(2), np.zeros((2, 2)), 0.1 * np.eye(1), 100.0 * np.eye(2), 20.0
    else:
        dt, N, mk = 0.05, 60, lambda: pycddp.CartPole(0.05, "rk4", 1.0, 0.2, 0.5, 9.81, 0.0)
        x0, goal, Q, R, Qf, lim = np.zeros(4), np.array([0.0, np.pi, 0.0, 0.0]), np.zeros((4, 4)), 0.1 * np.eye(1), 100.0 * np.eye(4), 5.0

    class PyQuadratic(pycddp.NonlinearObjective):
        def running_cost(self, x, u, index): return float(dt * ((x - goal) @ Q @ (x - goal) + u @ R @ u))
        def terminal_cost(self, x): return float((x - goal) @ Qf @ (x - goal))
    o = _options(pycddp, max_iterations=40, tolerance=1e-4, acceptable_tolerance=1e-5)

    def build(obj):
        sv = pycddp.CDDP(x0, goal, N, dt, o)
        sv.set_dynamical_system(mk()); sv.set_objective(obj)
        sv.add_constraint("ControlConstraint", pycddp.ControlConstraint(np.array([-lim]), np.array([lim])))
        return sv
    host = build(PyQuadratic(dt))
    assert host._needs_host_plugins()
    hs = host.solve(pycddp.SolverType.IPDDP)
    dev = build(pycddp.QuadraticObjective(Q, R, Qf, goal, [], dt))
    assert not dev._needs_host_plugins()
    ds = dev.solve(pycddp.SolverType.IPDDP)
    print(plant, hs.status_message, hs.iterations_completed, hs.final_objective, "| device", ds.status_message, ds.iterations_completed, ds.final_objective)
    assert hs.status_message == ds.status_message      # finite-difference derivatives: same decisions are not guaranteed, same outcome is
    assert abs(hs.final_objective - ds.final_objective) < 1e-4 * abs(ds.final_objective)
    assert np.max(np.abs(np.stack(hs.state_trajectory) - np.stack(ds.state_trajectory))) < 1e-2
    # the plant's host step IS the device's step: rolling the returned controls out on the host reproduces the returned states
    x = x0.copy()
    for t in range(N):
        x = mk().get_discrete_dynamics(x, hs.control_trajectory[t]) if t == 0 else host._sys.get_discrete_dynamics(x, hs.control_trajectory[t])
        assert np.array_equal(x, hs.state_trajectory[t + 1])
