"""LogDDP (f4): the reference's single-shooting relaxed-log-barrier solver (src/cddp_core/logddp_solver.cpp:43-707,
include/cddp-cpp/cddp_core/barrier.hpp:37-296) as
  * a C++ oracle solver (oracle/cddp_oracle.cpp, solver id CDDP_HIP_SOLVER_LOGDDP),
  * a second, independently written numpy restatement (oracle/twin/logddp_twin.py::LogDDP),
  * the product: cddp_hip_plugin_solve(solver = LOGDDP) -- barrier terms folded into the cost-derivative stacks and the filter line
    search on the host, the Riccati sweep of the batch on the GPU (the stack-fed CDDP_HIP_STACKS_LOGDDP branch of round 2) --
    reached through the pycddp-compatible facade for built-in AND user plug-ins.
CPU: oracle == twin in iteration count, status, sweep / rollout counts, objective and trajectory on six problems (box, ball, cone and
thrust-magnitude rows, the discrete car), the reference's cold-start rollout test.  GPU: product vs oracle, and the reference's own
LogDDP solve tests (tests/cddp_core/test_logddp_solver.cpp:154-491) replayed with their assertions."""
import importlib.util
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.host_arithmetic   # host route of the library: glibc on both sides (tests/conftest.py)

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(REPO, "oracle", "twin"))
sys.path.insert(0, os.path.join(HERE, "golden"))


@pytest.fixture(scope="module")
def pycddp(api):
    name = "pycddp_amd"
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(name, os.path.join(REPO, "cddp-cpp_amd", "pycddp_amd.py"))
    mod = importlib.util.module_from_spec(spec); sys.modules[name] = mod; spec.loader.exec_module(mod)
    return mod


def _cases(api):
    import make_twin_golden as G
    def lg(p):
        p.c.solver = api.SOLVER_LOGDDP
        return p
    return {
        "pendulum_box": (lambda: G._pendulum("IPDDP", True), lambda: lg(api.pendulum_problem(api.SOLVER_IPDDP, True))),
        "cartpole_box": (lambda: G._cartpole("IPDDP", True), lambda: lg(api.cartpole_problem(api.SOLVER_IPDDP, True))),
        "unicycle_box_ball": (lambda: G._unicycle("IPDDP", True), lambda: lg(api.unicycle_problem(api.SOLVER_IPDDP, 100, True))),
        "unicycle_thrust": (G.CASES["unicycle_ipddp_thrust"], lambda: lg(api.unicycle_thrust_problem())),
        "unicycle_box_soc": (G.CASES["unicycle_ipddp_box_soc"], lambda: lg(api.unicycle_cone_problem())),
        "car_box": (G.CASES["car_ipddp_box"], lambda: lg(api.car_problem())),
    }


NAMES = ["pendulum_box", "cartpole_box", "unicycle_box_ball", "unicycle_thrust", "unicycle_box_soc", "car_box"]


@pytest.mark.parametrize("name", NAMES)
def test_oracle_matches_the_numpy_restatement(api, oracle_built, name):
    import logddp_twin as L
    mk_spec, mk_p = _cases(api)[name]
    spec = mk_spec(); p = mk_p()
    U0 = spec.get("U0")
    tw = L.LogDDP(spec); tw.set_initial(spec["x0"], U0); r = tw.solve()
    o = api.Oracle(p); o.set_initial(np.array(spec["x0"], float), U0); ro = o.solve()
    X, U = o.trajectory()
    assert (ro["iterations"], api.STATUS_STRINGS[ro["status"]], ro["n_backward"], ro["n_forward"]) == (r["iterations"], r["status"], r["n_backward"], r["n_forward"]), (name, r, ro)
    assert abs(ro["final_objective"] - r["final_objective"]) <= 1e-10 * max(1.0, abs(r["final_objective"]))
    assert np.max(np.abs(X - tw.X)) < 1e-9 and np.max(np.abs(U - tw.U)) < 1e-8
    assert abs(ro["barrier_mu"] - r["mu"]) <= 1e-15


def test_cold_start_rolls_out_the_provided_control_guess(api, oracle_built):
    """tests/cddp_core/test_logddp_solver.cpp:28-85: a same-sized STATE guess is only a guess -- initialize() re-rolls X from U."""
    o = api.default_options()
    p = api.Problem(api.SOLVER_LOGDDP, api.MODEL_PENDULUM, api.EULER, 2, 1, 4, 0.05, np.zeros((2, 2)), np.eye(1), np.eye(2), np.zeros(2),
                    model_params=[1.0, 1.0, 0.0, 9.81], options=o)
    x0 = np.array([np.pi, 0.2]); U = 0.1 * (np.arange(4) + 1.0).reshape(4, 1)
    Xg = np.full((5, 2), 42.0); Xg[0] = x0
    orc = api.Oracle(p); orc.set_initial(x0, U, Xg); orc.initialize()
    X, _ = orc.trajectory()
    x = x0.copy()
    for t in range(4):
        _, x, _, _ = orc.dynamics(x, U[t])
        assert np.max(np.abs(X[t + 1] - x)) < 1e-12
    assert np.array_equal(X[0], x0)


def _facade(pycddp, api, p, x0, U0):
    """A pyapi built-in problem posed on the pycddp-compatible facade (the product's user-facing entry)."""
    plants = {api.MODEL_PENDULUM: lambda: pycddp.Pendulum(p.dt, p.c.model_params[0], p.c.model_params[1], p.c.model_params[2], "euler"),
              api.MODEL_CARTPOLE: lambda: pycddp.CartPole(p.dt, "rk4", *list(p.c.model_params)[:5]),
              api.MODEL_UNICYCLE: lambda: pycddp.Unicycle(p.dt, "euler"), api.MODEL_CAR: lambda: pycddp.Car(p.dt, p.c.model_params[0], "euler")}
    o = pycddp.CDDPOptions(); o.verbose = False; o.print_solver_header = False
    o.max_iterations = p.options.max_iterations; o.tolerance = p.options.tolerance; o.acceptable_tolerance = p.options.acceptable_tolerance
    o.regularization.initial_value = p.options.reg_initial_value
    sv = pycddp.CDDP(x0, p.x_ref, p.N, p.dt, o)
    sv.set_dynamical_system(plants[p.c.model]())
    sv.set_objective(pycddp.QuadraticObjective(p.Q, p.R, p.Qf, p.x_ref, [], p.dt))
    return sv


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_product_logddp_matches_the_oracle(api, pycddp, oracle_built, name):
    """pycddp facade -> cddp_hip_plugin_solve(LOGDDP): status, iteration count, sweep and rollout counts of the oracle on every
    trajectory of a small batch; objective 1e-9, trajectory 1e-6.  (The GPU sweep adds the folded barrier terms in another association
    than the reference -- (l + L) + A^T V against (l + A^T V) + L -- so agreement is to rounding, not bitwise.)"""
    mk_spec, mk_p = _cases(api)[name]
    spec = mk_spec(); p = mk_p()
    B = 3
    x0 = api.batch_x0(p, B, 20260929, 0.02 * np.ones(p.nx)); x0[0] = np.array(spec["x0"], float)
    U0 = spec.get("U0")
    sv = _facade(pycddp, api, p, x0[0], U0)
    cons = spec["constraints"]
    import cddp_twin as T
    for cname in sorted(cons):
        c = cons[cname]
        if isinstance(c, T.ControlBox): sv.add_constraint(cname, pycddp.ControlConstraint(c.lo, c.up))
        elif isinstance(c, T.Ball): sv.add_constraint(cname, pycddp.BallConstraint(c.r, c.c))
        elif isinstance(c, T.SecondOrderCone): sv.add_constraint(cname, pycddp.SecondOrderConeConstraint([0.0, -0.5, 0.0], [0.0, 1.0, 0.0], np.pi / 4.0 + 0.35, 1e-6))
        elif isinstance(c, T.ThrustMagnitude): sv.add_constraint(cname, pycddp.ThrustMagnitudeConstraint(c.mn, c.mx, c.eps) if c.mn is not None else pycddp.MaxThrustMagnitudeConstraint(c.mx, c.eps))
    if U0 is not None:
        sv.set_initial_trajectory([x0[0]] * (p.N + 1), list(np.asarray(U0)))
    sv.logddp_route = "plugin"    # this module pins the HOST route (glibc on both sides); the resident kernels: tests/test_logddp_device.py
    sols = sv.solve_batch(list(x0), pycddp.SolverType.LogDDP)
    U0b = None if U0 is None else np.tile(np.asarray(U0)[None], (B, 1, 1))
    ores, oX, oU, _, _ = api.oracle_solve_batch(p, x0, U0b, None, n_threads=B)
    for b in range(B):
        s = sols[b]
        assert s.solver_name == "LogDDP"
        assert (s.status_message, s.iterations_completed) == (api.STATUS_STRINGS[int(ores["status"][b])], int(ores["iterations"][b])), (name, b)
        assert abs(s.final_objective - ores["final_objective"][b]) <= 1e-9 * max(1.0, abs(ores["final_objective"][b]))
        assert np.max(np.abs(np.stack(s.state_trajectory) - oX[b])) < 1e-6 and np.max(np.abs(np.stack(s.control_trajectory) - oU[b])) < 1e-5
        assert abs(s.final_barrier_mu - ores["barrier_mu"][b]) < 1e-15


@pytest.mark.gpu
def test_reference_logddp_pendulum_and_unicycle_solves(api, pycddp):
    """tests/cddp_core/test_logddp_solver.cpp:154-277 (SolvePendulum, N = 500) and :358-417 (SolveUnicycle, enable_parallel): the
    reference's problems, options and assertions."""
    N, dt = 500, 0.05
    o = pycddp.CDDPOptions(); o.verbose = False; o.print_solver_header = False
    o.max_iterations = 100; o.tolerance = 1e-3; o.acceptable_tolerance = 1e-4; o.regularization.initial_value = 1e-6; o.return_iteration_info = True
    x0 = np.array([np.pi, 0.0]); goal = np.zeros(2)
    obj = pycddp.QuadraticObjective(np.zeros((2, 2)), 0.1 * np.eye(1), 100.0 * np.eye(2), goal, [], dt)
    J = sum(obj.running_cost(x0, np.zeros(1), t) for t in range(N)) + obj.terminal_cost(x0)
    sv = pycddp.CDDP(x0, goal, N, dt, o)
    sv.set_dynamical_system(pycddp.Pendulum(dt, 1.0, 1.0, 0.0, "euler")); sv.set_objective(obj)
    sv.add_constraint("ControlConstraint", pycddp.ControlConstraint(np.array([-10.0]), np.array([10.0])))
    sv.set_initial_trajectory([x0] * (N + 1), [np.zeros(1)] * N)
    sol = sv.solve(pycddp.SolverType.LogDDP)
    print("LogDDP pendulum:", sol.status_message, sol.iterations_completed, sol.final_objective, "initial", J)
    assert sol.status_message in ("OptimalSolutionFound", "AcceptableSolutionFound")     # "Algorithm should converge"
    assert sol.iterations_completed > 0 and sol.final_objective < J
    assert np.max(np.abs(np.stack(sol.control_trajectory))) <= 10.0
    # SolveUnicycle
    N, dt = 100, 0.03
    o = pycddp.CDDPOptions(); o.verbose = False; o.print_solver_header = False; o.max_iterations = 40; o.tolerance = 1e-2; o.enable_parallel = True; o.num_threads = 10
    goal = np.array([2.0, 2.0, np.pi / 2.0])
    sv = pycddp.CDDP(np.array([0.0, 0.0, np.pi / 4.0]), goal, N, dt, o)
    sv.set_dynamical_system(pycddp.Unicycle(dt, "euler"))
    sv.set_objective(pycddp.QuadraticObjective(np.zeros((3, 3)), 0.5 * np.eye(2), 0.5 * np.diag([50.0, 50.0, 10.0]), goal, [], dt))
    sv.add_constraint("ControlConstraint", pycddp.ControlConstraint(np.array([-1.0, -np.pi]), np.array([1.0, np.pi])))
    sv.set_initial_trajectory([np.zeros(3)] * (N + 1), [np.zeros(2)] * N)
    sol = sv.solve(pycddp.SolverType.LogDDP)
    print("LogDDP unicycle:", sol.status_message, sol.iterations_completed, sol.final_objective)
    assert sol.status_message in ("OptimalSolutionFound", "AcceptableSolutionFound")


@pytest.mark.gpu
def test_logddp_with_a_python_plant(api, pycddp, oracle_built):
    """A USER plant through LogDDP: the Python pendulum of tests/test_host_plugins.py against the oracle's built-in pendulum."""
    import math
    class PyPendulum(pycddp.DynamicalSystem):
        def __init__(self): super().__init__(2, 1, 0.02, "euler")
        def get_continuous_dynamics(self, x, u, t=0.0): return np.array([x[1], (u[0] - 0.01 * x[1] + 9.81 * 0.5 * math.sin(x[0])) / 0.25])
        def get_state_jacobian(self, x, u, t=0.0): return np.array([[0.0, 1.0], [(9.81 / 0.5) * math.cos(x[0]), -0.01 / 0.25]])
        def get_control_jacobian(self, x, u, t=0.0): return np.array([[0.0], [4.0]])
    p = api.pendulum_problem(api.SOLVER_IPDDP, True); p.c.solver = api.SOLVER_LOGDDP
    o = pycddp.CDDPOptions(); o.verbose = False; o.max_iterations = 30; o.tolerance = 1e-4; o.acceptable_tolerance = 1e-5; o.regularization.initial_value = 1e-6
    sv = pycddp.CDDP(np.array([np.pi, 0.0]), np.zeros(2), 100, 0.02, o)
    sv.set_dynamical_system(PyPendulum())
    sv.set_objective(pycddp.QuadraticObjective(np.zeros((2, 2)), 0.1 * np.eye(1), 100.0 * np.eye(2), np.zeros(2), [], 0.02))
    sv.add_constraint("ControlConstraint", pycddp.ControlConstraint(np.array([-20.0]), np.array([20.0])))
    s = sv.solve(pycddp.SolverType.LogDDP)
    orc = api.Oracle(p); orc.set_initial(np.array([np.pi, 0.0]), None); r = orc.solve()
    assert (s.status_message, s.iterations_completed) == (api.STATUS_STRINGS[r["status"]], r["iterations"])
    assert abs(s.final_objective - r["final_objective"]) < 1e-9
