"""The reference's Python-facing tests that are about the API surface (no solve, or a short one), restated against the facade so that they
travel to the GPU box: python/tests/test_constraints.py, python/tests/test_solver_errors.py:21-124, python/tests/test_all_dynamics.py.
In the build container the reference's own files were also run UNCHANGED against the facade (`sys.modules["pycddp"] = pycddp_amd`):
every test that needs no GPU passes (34 of 50; the remainder are 16 solves -- replayed
on the GPU here and in test_pycddp_facade / test_host_plugins / test_pycddp_portfolio; with the five host-only plants at the bottom all
13 tests of test_all_dynamics.py pass).
Error types and message fragments are the ones the reference's tests match on (bind_solver.cpp:106-152, 478-510, 640-650)."""
import importlib.util
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)


@pytest.fixture(scope="module")
def pycddp(api):
    name = "pycddp_amd"
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(name, os.path.join(REPO, "cddp-cpp_amd", "pycddp_amd.py"))
    mod = importlib.util.module_from_spec(spec); sys.modules[name] = mod; spec.loader.exec_module(mod)
    return mod


def _make_solver(pycddp, horizon=6, dt=0.1):
    opts = pycddp.CDDPOptions(); opts.verbose = False; opts.print_solver_header = False
    return pycddp.CDDP(np.zeros(2), np.zeros(2), horizon, dt, opts)


# ---------------------------------------------------------------------------------------------- python/tests/test_constraints.py
def test_builtin_constraints(pycddp):
    c = pycddp.ControlConstraint(np.array([-1.0, -2.0]), np.array([1.0, 2.0]))
    assert c.get_dual_dim() == 4 and c.name == "ControlConstraint"                   # :42-45
    assert pycddp.StateConstraint(np.array([-5.0, -5.0]), np.array([5.0, 5.0])).get_dual_dim() == 4   # :48-50
    center = np.array([1.0, 1.0])
    b = pycddp.BallConstraint(radius=0.5, center=center)                              # :53-63
    assert b.get_dual_dim() == 1
    np.testing.assert_array_equal(b.get_center(), center)
    assert b.evaluate(np.array([5.0, 5.0]), np.array([0.0])).shape[0] == 1
    lin = pycddp.LinearConstraint(np.array([[1.0, 1.0], [-1.0, 1.0]]), np.array([1.0, 1.0]))   # :66-75
    assert lin.get_dual_dim() == 2 and lin.evaluate(np.zeros(2), np.zeros(1)).shape[0] == 2
    # the rest of the bound surface (bind_constraints.cpp:95-117): index argument, bounds, violations, Hessian lists, names
    x, u = np.array([0.3, -0.2]), np.array([1.5, 0.5])
    assert np.all(np.isneginf(c.get_lower_bound())) and c.get_upper_bound().shape == (4,)
    assert c.compute_violation(x, u, 3) == pytest.approx(0.5) and c.compute_violation_from_value(c.evaluate(x, u, 0)) == pytest.approx(0.5)
    assert [h.shape for h in c.get_control_hessian(x, u, 0)] == [(2, 2)] * 4 and [h.shape for h in c.get_cross_hessian(x, u)] == [(2, 2)] * 4
    assert np.array_equal(b.get_state_hessian(x, u)[0], -2.0 * np.eye(2))
    with pytest.raises(RuntimeError, match="does not have a center"):
        c.get_center()
    assert [k.name for k in (lin, b, pycddp.ThrustMagnitudeConstraint(0.0, 1.0), pycddp.MaxThrustMagnitudeConstraint(1.0))] == [
        "LinearConstraint", "BallConstraint", "ThrustMagnitudeConstraint", "MaxThrustMagnitudeConstraint"]


def test_constraint_base_is_rejected_cleanly(pycddp):                                 # :113-139
    solver = _make_solver(pycddp, 8)
    solver.set_dynamical_system(pycddp.LTISystem(np.array([[0.0, 1.0], [0.0, 0.0]]), np.array([[0.0], [1.0]]), 0.1))
    with pytest.raises(TypeError, match="Constraint is an abstract base class"):
        solver.add_constraint("bad", pycddp.Constraint("bad"))
    with pytest.raises(RuntimeError, match="pure virtual"):
        pycddp.Constraint("bare").evaluate(np.zeros(2), np.zeros(1))


class _Counting:
    """CountingAffineConstraint of test_constraints.py:7-39, built on the facade's Constraint at call time."""
    @staticmethod
    def make(pycddp, counters):
        class CountingAffineConstraint(pycddp.Constraint):
            def __init__(self):
                super().__init__("CountingAffineConstraint")
            def get_dual_dim(self): return 1
            def evaluate(self, state, control, index=0): counters["evaluate"] += 1; return np.array([state[0] - 10.0])
            def get_lower_bound(self): return np.array([-np.inf])
            def get_upper_bound(self): return np.array([0.0])
            def get_state_jacobian(self, state, control, index=0): counters["state_jacobian"] += 1; return np.array([[1.0, 0.0]])
            def get_control_jacobian(self, state, control, index=0): counters["control_jacobian"] += 1; return np.array([[0.0]])
            def compute_violation(self, state, control, index=0): return max(0.0, float(self.evaluate(state, control, index)[0]))
            def compute_violation_from_value(self, g): return max(0.0, float(g[0]))
        return CountingAffineConstraint()


@pytest.mark.gpu
def test_custom_python_constraint_with_solver(pycddp):                                # :78-110
    counters = {"evaluate": 0, "state_jacobian": 0, "control_jacobian": 0}
    dt, horizon = 0.05, 20
    xref = np.zeros(2)
    opts = pycddp.CDDPOptions(); opts.max_iterations = 10; opts.verbose = False; opts.print_solver_header = False
    solver = pycddp.CDDP(np.array([np.pi, 0.0]), xref, horizon, dt, opts)
    solver.set_dynamical_system(pycddp.Pendulum(dt, length=0.5, mass=1.0, damping=0.01))
    solver.set_objective(pycddp.QuadraticObjective(np.zeros((2, 2)), 0.1 * np.eye(1), 100.0 * np.eye(2), xref, [], dt))
    con = _Counting.make(pycddp, counters)
    assert con.name == "CountingAffineConstraint"
    solver.add_constraint("custom", con)
    solution = solver.solve(pycddp.SolverType.LogDDP)
    assert solution.solver_name == "LogDDP" and solution.status_message
    assert counters["evaluate"] > 0 and counters["state_jacobian"] > 0 and counters["control_jacobian"] > 0


# ---------------------------------------------------------------------------------------------- python/tests/test_solver_errors.py
def test_unknown_solver_and_abstract_bases(pycddp):
    with pytest.raises(ValueError, match="Unknown solver 'NONEXISTENT'"):             # :21-25
        _make_solver(pycddp).solve_by_name("NONEXISTENT")
    with pytest.raises(TypeError, match="DynamicalSystem is an abstract base class"):   # :76-80
        _make_solver(pycddp).set_dynamical_system(pycddp.DynamicalSystem(2, 1, 0.1))
    with pytest.raises(TypeError, match="No constructor defined"):                    # :83-85
        pycddp.Objective()


def test_set_initial_trajectory_validation(pycddp):
    solver = _make_solver(pycddp)
    N = solver.horizon
    with pytest.raises(ValueError, match="is a dynamical system set"):                # :66-73
        solver.set_initial_trajectory([np.zeros(2)] * (N + 1), [np.zeros(1)] * N)
    solver.set_dynamical_system(pycddp.Pendulum(0.1))
    with pytest.raises(ValueError, match="expected X length"):                        # :88-96
        solver.set_initial_trajectory([np.zeros(2)] * N, [np.zeros(1)] * N)
    X = [np.zeros(2) for _ in range(N + 1)]; X[2] = np.zeros(3)
    with pytest.raises(ValueError, match="state vector 2"):                           # :99-108
        solver.set_initial_trajectory(X, [np.zeros(1)] * N)
    U = [np.zeros(1) for _ in range(N)]; U[1] = np.zeros(2)
    with pytest.raises(ValueError, match="control vector 1"):                         # :111-120
        solver.set_initial_trajectory([np.zeros(2)] * (N + 1), U)
    solver.set_initial_trajectory([np.zeros(2)] * (N + 1), [np.zeros(1)] * N)         # a well-formed one is taken


@pytest.mark.gpu
@pytest.mark.parametrize("solver_name, expected", [("CLDDP", "CLDDP"), ("CLCDDP", "CLDDP"), ("LOGDDP", "LogDDP")])
def test_solve_by_name_accepts_core_aliases(pycddp, solver_name, expected):           # :28-63
    dt, horizon = 0.05, 20
    xref = np.zeros(2)
    opts = pycddp.CDDPOptions(); opts.max_iterations = 20; opts.verbose = False; opts.print_solver_header = False
    solver = pycddp.CDDP(np.array([np.pi, 0.0]), xref, horizon, dt, opts)
    solver.set_dynamical_system(pycddp.Pendulum(dt, length=0.5, mass=1.0, damping=0.01))
    solver.set_objective(pycddp.QuadraticObjective(np.zeros((2, 2)), 0.1 * np.eye(1), 100.0 * np.eye(2), xref, [], dt))
    solver.add_constraint("ctrl", pycddp.ControlConstraint(np.array([-50.0]), np.array([50.0])))
    solution = solver.solve_by_name(solver_name)
    assert solution.solver_name == expected and solution.status_message
    assert len(solution.state_trajectory) == horizon + 1


# ---------------------------------------------------------------------------------------------- python/tests/test_all_dynamics.py
def _check_model(model, x, u):                                                        # :6-19
    assert model.state_dim == x.shape[0] and model.control_dim == u.shape[0] and model.timestep > 0
    assert model.get_discrete_dynamics(x, u).shape == (model.state_dim,)
    assert model.get_state_jacobian(x, u).shape == (model.state_dim, model.state_dim)
    assert model.get_control_jacobian(x, u).shape == (model.state_dim, model.control_dim)


def test_all_provided_dynamics_models(pycddp):
    _check_model(pycddp.Pendulum(0.01, length=1.0, mass=1.0, damping=0.0), np.array([0.1, 0.0]), np.array([0.5]))          # :22-24
    _check_model(pycddp.Unicycle(0.1), np.zeros(3), np.array([1.0, 0.1]))                                                   # :27-29
    _check_model(pycddp.Bicycle(0.1, wheelbase=2.0), np.zeros(4), np.array([1.0, 0.1]))                                     # :32-34
    _check_model(pycddp.Car(0.03, wheelbase=2.0), np.zeros(4), np.array([1.0, 0.1]))                                        # :37-39
    _check_model(pycddp.CartPole(0.01), np.array([0.0, 0.0, 0.1, 0.0]), np.array([1.0]))                                    # :42-44
    _check_model(pycddp.Manipulator(0.01), np.zeros(6), np.array([0.1, 0.1, 0.1]))                                          # :57-59
    _check_model(pycddp.HCW(1.0, mean_motion=0.001, mass=1.0), np.zeros(6), np.array([0.01, 0.01, 0.01]))                   # :62-64
    _check_model(pycddp.LTISystem(np.array([[0, 1], [-1, 0]]), np.array([[0], [1]]), 0.01), np.array([1.0, 0.0]), np.array([0.5]))   # :82-86


def test_host_only_plants(pycddp):
    """DubinsCar, Acrobot, SpacecraftLinearFuel, DreyfusRocket, Usv3Dof (test_all_dynamics.py:47-79) have no device kernels: the facade
    restates them on the host (analytic / complex-step / the reference's finite-difference Jacobians) and solves them through the
    plug-in route.  Shapes as the reference checks them, Jacobians against central differences, and the properties the reference's C++
    plant tests assert (tests/dynamics_model/test_{dreyfus_rocket,acrobot,usv_3dof}.cpp)."""
    cases = [(pycddp.DubinsCar(1.0, 0.1), np.zeros(3), np.array([0.5])),                                                    # :47-49
             (pycddp.Acrobot(0.01), np.zeros(4), np.array([1.0])),                                                         # :52-54
             (pycddp.SpacecraftLinearFuel(1.0, mean_motion=0.001, isp=300.0), np.r_[np.zeros(6), 50.0, 0.0], np.array([0.01, 0.01, 0.01])),   # :67-69 (mass 50: the reference's all-zero state divides by the mass)
             (pycddp.DreyfusRocket(0.01), np.array([0.0, 100.0]), np.array([0.5])),                                         # :72-74
             (pycddp.Usv3Dof(0.1), np.zeros(6), np.array([1.0, 0.0, 0.0]))]                                                 # :77-79
    rng = np.random.default_rng(4)
    for model, x, u in cases:
        _check_model(model, x, u)
        xr = x + 0.3 * rng.standard_normal(x.size); ur = u + 0.3 * rng.standard_normal(u.size)
        if isinstance(model, pycddp.SpacecraftLinearFuel): xr[6] = 50.0
        A = model.get_state_jacobian(xr, ur); B = model.get_control_jacobian(xr, ur)
        Afd = pycddp._fd_jacobian(lambda s: model.get_continuous_dynamics(s, ur), xr, 1e-6)
        Bfd = pycddp._fd_jacobian(lambda c: model.get_continuous_dynamics(xr, c), ur, 1e-6)
        assert np.max(np.abs(A - Afd)) < 1e-6 and np.max(np.abs(B - Bfd)) < 1e-6, type(model).__name__
    rocket = pycddp.DreyfusRocket(0.05)                                   # test_dreyfus_rocket.cpp:27-68
    assert rocket.integration_type == "rk4" and rocket.get_thrust_acceleration() == 64.0 and rocket.get_gravity_acceleration() == 32.0
    st = np.array([0.0, 0.0]); h0 = st[0]
    for _ in range(100):
        st = rocket.get_discrete_dynamics(st, np.array([0.0]))
    assert st[0] > h0 and st[1] > 0.0
    acro = pycddp.Acrobot(0.01, integration_type="rk4")                   # test_acrobot.cpp:104-187
    B = acro.get_control_jacobian(np.array([0.1, 0.2, 0.0, 0.0]), np.array([0.5]))
    assert B[0, 0] == 0.0 and B[1, 0] == 0.0 and B[2, 0] != 0.0 and B[3, 0] != 0.0
    sd = acro.get_continuous_dynamics(np.array([np.pi / 4, np.pi / 6, 0.0, 0.0]), np.array([0.0]))
    assert abs(sd[0]) < 1e-10 and abs(sd[1]) < 1e-10 and abs(sd[2]) > 1e-6
    pos = acro.get_continuous_dynamics(np.array([0.1, 0.2, 0.0, 0.0]), np.array([1.0])); neg = acro.get_continuous_dynamics(np.array([0.1, 0.2, 0.0, 0.0]), np.array([-1.0]))
    assert pos[0] == neg[0] and pos[1] == neg[1] and pos[2] != neg[2] and pos[3] != neg[3]
    usv = pycddp.Usv3Dof(0.1)                                             # test_usv_3dof.cpp:30-90
    x0 = np.array([0.0, 0.0, 0.0, 1.0, 0.0, 0.0]); u0 = np.array([10.0, 0.0, 1.0])
    assert not np.allclose(usv.get_discrete_dynamics(x0, u0), x0)
    assert all(np.allclose(h, 0.0) and h.shape == (3, 3) for h in usv.get_control_hessian(x0, u0)) and len(usv.get_control_hessian(x0, u0)) == 6


def _host_plant_cases(pycddp):
    dub = dict(make=lambda dt: pycddp.DubinsCar(1.0, dt), dt=0.1, N=40, x0=np.zeros(3), goal=np.array([2.0, 1.0, 0.5]), Qf=50.0 * np.eye(3),
               R=0.1 * np.eye(1), box=1.5)
    Qf = np.diag([50.0] * 6 + [0.0, 0.0])
    sc = dict(make=lambda dt: pycddp.SpacecraftLinearFuel(dt, mean_motion=0.001, isp=300.0), dt=1.0, N=30,
              x0=np.array([10.0, 5.0, 2.0, 0.0, 0.0, 0.0, 50.0, 0.0]), goal=np.r_[np.zeros(6), 50.0, 0.0], Qf=Qf, R=0.1 * np.eye(3), box=2.0)
    return {"dubins_car": dub, "spacecraft_linear_fuel": sc}


@pytest.mark.gpu
@pytest.mark.parametrize("plant", ["dubins_car", "spacecraft_linear_fuel"])
def test_host_only_plant_solves_through_the_plugin_route(pycddp, plant):
    """A Dubins car steered to a pose and the fuel-aware HCW spacecraft brought to the origin, by CLDDP and IPDDP with a control box: GPU
    backward passes on the stack-fed sweeps of shapes (3, 1, .) and (8, 3, .), host rollouts of the restated plants."""
    c = _host_plant_cases(pycddp)[plant]
    dt, N, x0, goal = c["dt"], c["N"], c["x0"], c["goal"]
    nu = c["R"].shape[0]
    stand_still = float((x0 - goal) @ c["Qf"] @ (x0 - goal))
    for stype in (pycddp.SolverType.CLDDP, pycddp.SolverType.IPDDP):
        opts = pycddp.CDDPOptions(); opts.max_iterations = 60; opts.verbose = False; opts.print_solver_header = False
        solver = pycddp.CDDP(x0, goal, N, dt, opts)
        solver.set_dynamical_system(c["make"](dt))
        solver.set_objective(pycddp.QuadraticObjective(np.zeros((x0.size, x0.size)), c["R"], c["Qf"], goal, [], dt))
        solver.add_constraint("ControlConstraint", pycddp.ControlConstraint(-c["box"] * np.ones(nu), c["box"] * np.ones(nu)))
        sol = solver.solve(stype)
        X = np.stack(sol.state_trajectory); U = np.stack(sol.control_trajectory)
        print(plant, stype, sol.status_message, sol.iterations_completed, sol.final_objective, stand_still)
        assert sol.status_message and np.all(np.isfinite(X)) and np.max(np.abs(U)) <= c["box"] + 1e-9
        assert sol.final_objective < 0.5 * stand_still          # well below the cost of standing still
        x = x0.copy(); model = c["make"](dt)
        for t in range(N):                                      # the returned trajectory is a rollout of the plant
            x = model.get_discrete_dynamics(x, U[t])
            assert np.max(np.abs(x - X[t + 1])) < 1e-9 * max(1.0, float(np.max(np.abs(x))))
