"""f3 tail: the car and bicycle plants, the second-order-cone and thrust-magnitude path
constraints (constraint.hpp:626-1048), Hessian tensors for the 3-DOF manipulator and the quadrotor.

CPU: the reference's own known-answer values (tests/dynamics_model/test_car.cpp: MATLAB-derived step / Jacobian / Hessian entries at
1e-4; tests/dynamics_model/test_bicycle.cpp; tests/cddp_core/test_constraint.cpp:236-303) replayed on the oracle AND on the
product's host evaluation of the same plants (cddp_hip_model_eval); oracle vs numpy twin vs product at 1e-12; Hessians against
finite differences of the Jacobians.  The device-resident parity cases (step and solve level, full-DDP variants) live in
tests/test_gpu_parity.py (F3_CASES), tests/test_twin_golden.py (fixtures) and tests/test_ddp_second_order.py; the reference's
car-parking solve is replayed through the plug-in path in tests/cpp/test_host_api.cpp."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "oracle", "twin"))


def _eval_all(api, p, x, u, hess=True):
    """(step, A = I + dt f_x, B = dt f_u, dt * Hessians) from the oracle and from the product's host models."""
    o = api.Oracle(p)
    _, xn, Fx, Fu = o.dynamics(x, u)
    H = o.hessians(x, u) if hess else None
    mp = np.array(list(p.c.model_params), dtype=np.float64)
    r = api.model_eval(p.c.model, p.c.integrator, p.dt, mp, p.nx, p.nu, x, u, want=("step", "jac") + (("hess",) if hess else ()))
    return (xn, Fx, Fu, H), (r["step"], r["jac"][0], r["jac"][1], r.get("hess"))


def test_car_known_answers_of_the_reference(api, oracle_built):
    """tests/dynamics_model/test_car.cpp:21-195 (values from the original MATLAB demo, tolerance 1e-4)."""
    p = api.car_problem(api.SOLVER_IPDDP, 10)
    dt = p.dt
    for src in (0, 1):
        def ev(x, u):
            return _eval_all(api, p, np.array(x, float), np.array(u, float))[src]
        xn, _, _, _ = ev([1.0, 1.0, 1.5 * np.pi, 0.0], [0.01, 0.01])
        assert np.allclose(xn, [1.0, 1.0, 4.7124, 0.0003], atol=1e-4)
        xn, Fx, Fu, H = ev([1.0, 1.0, 1.5 * np.pi, 1.0], [0.3, 0.1])
        assert np.allclose(xn, [1.0, 0.9713, 4.7168, 1.0030], atol=1e-4)
        A = np.eye(4) + dt * Fx; B = dt * Fu
        assert np.allclose(A, [[1, 0, 0.0287, 0], [0, 1, 0, -0.0287], [0, 0, 1, 0.0044], [0, 0, 0, 1]], atol=1e-4)
        assert np.allclose(B, [[0, 0], [0.0087, 0], [0.0143, 0], [0, 0.03]], atol=1e-4)
        Fxx, Fuu, Fux = H
        assert abs(dt * Fxx[2][3, 3] - 8.71e-08) < 1e-4 and abs(dt * Fuu[2][0, 0] + 0.00443) < 1e-4     # d2 theta / dv2, d2 theta / d delta2
        assert abs(dt * Fxx[0][3, 2] - 0.0287) < 1e-4 and abs(dt * Fuu[0][0, 1]) < 1e-4                   # d2 x / dv dtheta, d2 x / d delta da
        _, Fx0, Fu0, H0 = ev([1.0, 1.0, 1.5 * np.pi, 0.0], [0.01, 0.01])
        assert np.allclose(np.eye(4) + dt * Fx0, [[1, 0, 0, 0], [0, 1, 0, -0.03], [0, 0, 1, 0.0001], [0, 0, 0, 1]], atol=1e-4)
        assert np.allclose(dt * Fu0, [[0, 0], [0, 0], [0, 0], [0, 0.03]], atol=1e-4)
        assert dt * H0[0][2][3, 3] < 8.71e-08 and abs(dt * H0[1][2][0, 0]) < 0.00443


def test_hcw_properties(api, oracle_built):
    """The Hill-Clohessy-Wiltshire plant (src/dynamics_model/spacecraft_linear.cpp:24-120; the reference's own tests,
    tests/dynamics_model/test_spacecraft_linear.cpp:31-140, only propagate and check dimensions): the radial-hover thrust of its
    ContinuousDynamics case balances the tidal term; the RK4 step on its 500-km orbit (dt = 10 s, the DiscreteDynamics case) follows the
    closed-form solution of the equations over one revolution; constant Jacobians, zero Hessians; oracle, twin and the product's host model
    agree."""
    import cddp_twin as T
    p = api.hcw_problem(api.SOLVER_IPDDP, 10)
    n, mass = p.c.model_params[0], p.c.model_params[1]
    o = api.Oracle(p)
    x = np.zeros(6); x[0] = 100.0
    u = np.array([-mass * 3.0 * n * n * x[0], 0.0, 0.0])
    xd = o.dynamics(x, u)[0]
    assert np.max(np.abs(xd)) < 1e-12                                     # hover: nothing moves
    # closed form of the unforced equations (Clohessy & Wiltshire 1960) from the reference test's initial state
    x0 = p.x0.copy(); t = 0.0
    xs = x0.copy()
    period = 2.0 * np.pi / n
    steps = int(period / p.dt)
    for _ in range(steps):
        xs = o.dynamics(xs, np.zeros(3))[1]
    T_ = steps * p.dt; s, c = np.sin(n * T_), np.cos(n * T_)
    X, Y, Z, VX, VY, VZ = x0
    xa = (4 - 3 * c) * X + s / n * VX + 2 / n * (1 - c) * VY
    ya = 6 * (s - n * T_) * X + Y - 2 / n * (1 - c) * VX + (4 * s - 3 * n * T_) / n * VY
    za = c * Z + s / n * VZ
    assert np.allclose(xs[:3], [xa, ya, za], rtol=0, atol=1e-6 * np.max(np.abs(x0[:3])) + 2e-4)
    (xn, Fx, Fu, H), (sn, fx, fu, h) = _eval_all(api, p, x0, np.array([0.1, -0.2, 0.05]))
    assert np.array_equal(xn, sn) and np.array_equal(Fx, fx) and np.array_equal(Fu, fu)
    tw = T.HCW(n, mass)
    A, B = tw.jac(x0, np.zeros(3), 0.0)
    assert np.array_equal(A, Fx) and np.array_equal(B, Fu)
    assert all(np.all(np.asarray(b) == 0.0) for b in H) and all(np.all(np.asarray(b) == 0.0) for b in h)


def test_bicycle_properties_of_the_reference(api, oracle_built):
    """tests/dynamics_model/test_bicycle.cpp:26-140: straight motion, steering turns, analytic Jacobians against finite differences."""
    p = api.bicycle_problem(api.SOLVER_IPDDP, 10)
    o = api.Oracle(p)
    xd, _, _, _ = o.dynamics(np.array([0.0, 0.0, 0.0, 1.0]), np.zeros(2))
    assert np.allclose(xd, [1.0, 0.0, 0.0, 0.0], atol=1e-10)
    xd, _, _, _ = o.dynamics(np.array([0.0, 0.0, 0.0, 1.0]), np.array([0.0, 0.1]))
    assert abs(xd[2]) > 0.0
    x = np.array([1.0, 2.0, np.pi / 6, 1.5]); u = np.array([0.5, 0.1])
    (_, Fx, Fu, _), (_, fx, fu, _) = _eval_all(api, p, x, u)
    h = 1e-6
    An = np.stack([(o.dynamics(x + h * e, u)[0] - o.dynamics(x - h * e, u)[0]) / (2 * h) for e in np.eye(4)], axis=1)
    Bn = np.stack([(o.dynamics(x, u + h * e)[0] - o.dynamics(x, u - h * e)[0]) / (2 * h) for e in np.eye(2)], axis=1)
    assert np.linalg.norm(Fx - An) < 1e-5 and np.linalg.norm(Fu - Bn) < 1e-5
    assert np.array_equal(fx, Fx) and np.array_equal(fu, Fu)


@pytest.mark.parametrize("plant", ["bicycle", "car", "manipulator", "quadrotor"])
def test_hessians_three_ways(api, oracle_built, plant):
    """Oracle (second-order duals / analytic overrides as the reference has them) vs the product's host models (closed forms or duals,
    written separately) vs finite differences of the oracle's Jacobians; car and bicycle also vs the numpy twin's hand-derived forms."""
    import cddp_twin as T
    p = {"bicycle": lambda: api.bicycle_problem(api.SOLVER_IPDDP, 10), "car": lambda: api.car_problem(api.SOLVER_IPDDP, 10),
         "manipulator": lambda: api.manipulator_problem(api.SOLVER_IPDDP, 10), "quadrotor": lambda: api.quadrotor_problem(api.SOLVER_IPDDP, 10, True)}[plant]()
    o = api.Oracle(p)
    rng = np.random.default_rng(11)
    for _ in range(3):
        x = rng.uniform(-0.8, 0.8, p.nx); u = rng.uniform(-0.4, 0.4, p.nu)
        if plant == "quadrotor":
            x[3:7] = [0.9, 0.2, -0.1, 0.3]
        (xn, Fx, Fu, H), (xn2, fx, fu, H2) = _eval_all(api, p, x, u)
        scale = max(1.0, max(np.max(np.abs(a)) for a in H))
        for a, b in zip(H, H2):
            assert np.max(np.abs(a - b)) < 1e-10 * scale, (plant, np.max(np.abs(a - b)))
        if plant in ("bicycle", "car"):
            tw = T.Bicycle(2.0) if plant == "bicycle" else T.Car(2.0, p.dt)
            for a, b in zip(H, tw.hess(x, u, 0.0)):
                assert np.max(np.abs(a - b)) < 1e-10 * scale
            A, B = tw.jac(x, u, 0.0)
            assert np.max(np.abs(A - Fx)) < 1e-12 and np.max(np.abs(B - Fu)) < 1e-12
        Fxx, Fuu, Fux = H
        h = 1e-6
        for j in range(p.nx):
            e = np.zeros(p.nx); e[j] = h
            _, _, Ap, Bp = o.dynamics(x + e, u); _, _, Am, Bm = o.dynamics(x - e, u)
            if plant != "manipulator":     # (manipulator.cpp:72-86 overrides the state / control Hessians with zeros)
                assert np.max(np.abs(Fxx[:, :, j] - (Ap - Am) / (2 * h))) < 2e-5 * scale, (plant, j)
            tol = 5e-4 if plant == "manipulator" else 2e-5     # the manipulator's JACOBIANS are central differences with h = 2e-5 themselves
            assert np.max(np.abs(Fux[:, :, j] - (Bp - Bm) / (2 * h))) < tol * scale, (plant, j)
        if plant != "manipulator":
            for j in range(p.nu):
                e = np.zeros(p.nu); e[j] = h
                _, _, _, Bp = o.dynamics(x, u + e); _, _, _, Bm = o.dynamics(x, u - e)
                assert np.max(np.abs(Fuu[:, :, j] - (Bp - Bm) / (2 * h))) < 2e-5 * scale
        else:
            assert not np.any(Fxx) and not np.any(Fuu) and np.any(Fux)


def test_second_order_cone_values_of_the_reference(api, oracle_built):
    """tests/cddp_core/test_constraint.cpp:236-303: inside / outside / boundary values and the analytic gradient."""
    o = api.default_options()
    p = api.Problem(api.SOLVER_IPDDP, api.MODEL_UNICYCLE, api.EULER, 3, 2, 4, 0.1, np.zeros((3, 3)), np.eye(2), np.eye(3), np.zeros(3), options=o)
    fov, eps = np.pi / 4.0, 1e-8
    p.add_second_order_cone("SecondOrderConeConstraint", [0.0, 0.0, 0.0], [0.0, 1.0, 0.0], fov, eps)
    orc = api.Oracle(p)
    u = np.zeros(2)
    g, _, _ = orc.constraint_eval(np.array([0.0, 1.0, 0.0]), u); assert g[0] < 0.0
    g, _, _ = orc.constraint_eval(np.array([0.0, -1.0, 0.0]), u); assert g[0] > 0.0
    g, _, _ = orc.constraint_eval(np.array([1.5 * np.tan(fov), 1.5, 0.0]), u); assert abs(g[0]) < 1e-6
    x = np.array([0.1, 0.5, 0.1])
    g, gx, gu = orc.constraint_eval(x, u)
    rn = np.sqrt(x @ x + eps)
    assert np.allclose(gx[0], np.cos(fov) * (x / rn) - np.array([0.0, 1.0, 0.0]), rtol=1e-6, atol=1e-12) and not np.any(gu)
    with pytest.raises(ValueError, match="Cone angle must be between 0 and PI"):
        p.add_second_order_cone("c", [0, 0, 0], [0, 1, 0], 4.0)
    with pytest.raises(ValueError, match="Regularization epsilon must be positive"):
        p.add_second_order_cone("c", [0, 0, 0], [0, 1, 0], 0.5, 0.0)
    with pytest.raises(ValueError, match="Opening direction cannot be zero vector"):
        p.add_second_order_cone("c", [0, 0, 0], [0, 0, 0], 0.5)


@pytest.mark.parametrize("two_sided", [True, False])
def test_thrust_magnitude_rows(api, oracle_built, two_sided):
    """constraint.hpp:840-880, 955-993: the value uses the plain norm, the Jacobian the regularised one; oracle vs twin vs closed form."""
    import cddp_twin as T
    p = api.unicycle_thrust_problem(api.SOLVER_IPDDP, 8, two_sided)
    orc = api.Oracle(p)
    tw = T.ThrustMagnitude(0.3 if two_sided else None, 2.0, 1e-6)
    rng = np.random.default_rng(5)
    for k in range(5):
        x = rng.normal(size=3); u = rng.normal(size=2) * (1e-9 if k == 4 else 1.0)
        g, gx, gu = orc.constraint_eval(x, u)
        n = np.sqrt(u @ u); d = u / np.sqrt(u @ u + 1e-6)
        if two_sided:
            assert np.allclose(g, [0.3 - n, n - 2.0], rtol=0, atol=1e-15) and np.allclose(gu, [-d, d], rtol=0, atol=1e-15)
        else:
            assert np.allclose(g, [n - 2.0], rtol=0, atol=1e-15) and np.allclose(gu, [d], rtol=0, atol=1e-15)
        assert not np.any(gx)
        assert np.max(np.abs(tw.g(x, u) - g)) < 1e-15 and np.max(np.abs(tw.jac(x, u)[1] - gu)) < 1e-15     # (numpy's dot may sum in another order)
    with pytest.raises(ValueError, match="min_thrust_norm must be non-negative"):
        p.add_thrust_magnitude("t", -1.0, 1.0)
    with pytest.raises(ValueError, match="greater than or equal to min_thrust_norm"):
        p.add_thrust_magnitude("t", 2.0, 1.0)
    with pytest.raises(ValueError, match="epsilon must be positive"):
        p.add_max_thrust_magnitude("t", 1.0, 0.0)
