"""pycddp-compatible front end (cddp-cpp_amd/pycddp_amd.py): the reference's Python tests
(python/tests/test_pendulum.py, test_options.py, test_solver_errors.py) restated against it."""
import importlib.util
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def pycddp():
    name = "pycddp_amd"
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(name, os.path.join(REPO, "cddp-cpp_amd", "pycddp_amd.py"))
    mod = importlib.util.module_from_spec(spec); sys.modules[name] = mod; spec.loader.exec_module(mod)
    return mod


def _pendulum(pycddp, horizon=50, dt=0.05, max_it=100, **kw):
    x0 = np.array([np.pi, 0.0]); xref = np.array([0.0, 0.0])
    opts = pycddp.CDDPOptions(); opts.max_iterations = max_it; opts.verbose = False; opts.print_solver_header = False
    for k, v in kw.items():
        setattr(opts, k, v)
    solver = pycddp.CDDP(x0, xref, horizon, dt, opts)
    solver.set_dynamical_system(pycddp.Pendulum(dt, length=0.5, mass=1.0, damping=0.01))
    solver.set_objective(pycddp.QuadraticObjective(np.zeros((2, 2)), 0.1 * np.eye(1), 100.0 * np.eye(2), xref, [], dt))
    solver.add_constraint("ctrl", pycddp.ControlConstraint(np.array([-50.0]), np.array([50.0])))
    return solver, x0, xref


# ---------------------------------------------------------------------------------------------- CPU: surface + errors
def test_options_defaults_and_pod_mapping(pycddp):                       # python/tests/test_options.py
    o = pycddp.CDDPOptions()
    assert o.tolerance == 1e-5 and o.max_iterations == 1 and o.use_ilqr and not o.warm_start
    assert o.line_search.max_iterations == 11 and o.regularization.initial_value == 1e-6
    assert o.ipddp.barrier.mu_initial == 1.0 and o.ipddp.barrier.strategy == pycddp.BarrierStrategy.ADAPTIVE
    o.max_iterations = 42; o.ipddp.barrier.mu_initial = 0.5; o.filter.armijo_constant = 3e-4; o.line_search.max_iterations = 7
    pod = o.to_pod()
    assert pod.max_iterations == 42 and pod.barrier_mu_initial == 0.5 and pod.filter_armijo_constant == 3e-4
    assert pod.ls_max_iterations == 7


def test_solver_errors(pycddp):                                           # python/tests/test_solver_errors.py
    x0 = np.zeros(2)
    solver = pycddp.CDDP(x0, x0, 6, 0.1, pycddp.CDDPOptions())
    with pytest.raises(ValueError, match="Unknown solver 'NONEXISTENT'"):
        solver.solve_by_name("NONEXISTENT")
    with pytest.raises(RuntimeError, match="Dynamical system must be set before solving."):
        solver.solve(pycddp.SolverType.IPDDP)
    solver.set_dynamical_system(pycddp.Pendulum(0.1))
    with pytest.raises(RuntimeError, match="Objective function must be set before solving."):
        solver.solve(pycddp.SolverType.IPDDP)
    with pytest.raises(ValueError):
        solver.set_initial_trajectory([x0] * 3, [np.zeros(1)] * 6)
    bare = pycddp.DynamicalSystem(2, 1, 0.1)                              # the trampoline base: constructible, nothing implemented
    with pytest.raises(RuntimeError, match="do not support getContinuousDynamicsAutodiff"):
        bare.get_state_jacobian(x0, np.zeros(1))
    with pytest.raises(ValueError, match="Q matrix must be square"):
        pycddp.QuadraticObjective(np.zeros((2, 3)), np.eye(1), np.eye(2), x0)
    unknown = solver.solve("FooDDP")                                      # cddp_core.cpp:243-265: no throw through solve()
    assert unknown.status_message.startswith("UnknownSolver") and unknown.iterations_completed == 0


# ---------------------------------------------------------------------------------------------- GPU: solves
def _check_fields(solution, horizon, name):                               # python/tests/test_pendulum.py:6-18
    assert solution.solver_name == name and solution.status_message
    assert solution.iterations_completed > 0 and solution.solve_time_ms >= 0
    assert np.isfinite(solution.final_objective) and np.isfinite(solution.final_step_length) and np.isfinite(solution.final_regularization)
    assert len(solution.time_points) == horizon + 1 and len(solution.state_trajectory) == horizon + 1
    assert len(solution.control_trajectory) == horizon and len(solution.feedback_gains) == horizon


@pytest.mark.gpu
@pytest.mark.parametrize("stype", ["CLDDP", "IPDDP"])
def test_pendulum_swing_up(pycddp, stype):                                # python/tests/test_pendulum.py:21-47
    solver, x0, xref = _pendulum(pycddp, return_iteration_info=True)
    solution = solver.solve(getattr(pycddp.SolverType, stype))
    _check_fields(solution, 50, stype)
    assert np.linalg.norm(solution.state_trajectory[-1] - xref) < np.linalg.norm(x0 - xref)
    h = solution.history
    assert len(h.objective) >= 1 and len(h.objective) == len(h.merit_function) == len(h.regularization)
    if stype == "IPDDP":
        assert len(h.barrier_mu) == len(h.objective)
    assert solver.solve_by_name("CLCDDP").solver_name == "CLDDP"          # alias (test_solver_errors.py:28-60)


@pytest.mark.gpu
def test_solve_batch_and_warm_start(pycddp, api):
    solver, x0, xref = _pendulum(pycddp, horizon=100, dt=0.02, max_it=60)
    single = solver.solve(pycddp.SolverType.IPDDP)
    x0s = [x0 + np.array([0.002 * b, 0.0]) for b in range(96)]
    solver2, _, _ = _pendulum(pycddp, horizon=100, dt=0.02, max_it=60)
    sols = solver2.solve_batch(x0s, pycddp.SolverType.IPDDP)
    assert len(sols) == 96
    assert sols[0].iterations_completed == single.iterations_completed and sols[0].final_objective == single.final_objective
    assert sum(s.status_message in ("OptimalSolutionFound", "AcceptableSolutionFound") for s in sols) >= 90
    # warm start from the previous solution (python/tests: warm_start option): not slower than cold + 5 iterations
    warm, _, _ = _pendulum(pycddp, horizon=100, dt=0.02, max_it=60, warm_start=True)
    warm.set_initial_trajectory(single.state_trajectory, single.control_trajectory)
    w = warm.solve(pycddp.SolverType.IPDDP)
    assert w.status_message in ("OptimalSolutionFound", "AcceptableSolutionFound")
    assert w.iterations_completed <= single.iterations_completed + 5


@pytest.mark.gpu
def test_unsupported_solver_is_loud(pycddp):
    solver, _, _ = _pendulum(pycddp)
    # every SolverType of the reference is served by now (CLDDP, IPDDP on the device-resident core; LogDDP, MSIPDDP on the host loop with
    # stack-fed GPU sweeps); a name nobody registered comes back as the reference's status string, not as an exception (cddp_core.cpp:243-265)
    s = solver.solve("ALDDP")
    assert s.status_message == "UnknownSolver - No solver registered for 'ALDDP'" and s.iterations_completed == 0
