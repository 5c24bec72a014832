"""Known answers and input data the reference's own tests hold for pieces of the path that no other test file of this repo replays yet:

  * tests/cddp_core/test_boxqp.cpp -- the two BoxQP input sets (5 and 15 variables; the reference prints the solution, it asserts nothing):
    replayed through the oracle's and the twin's BoxQP (boxqp.cpp:25-250), which must agree with each other, satisfy the KKT sign
    conditions and match a projected-gradient brute force.  Input data in tests/golden/ref_boxqp_inputs.json (make_ref_boxqp_inputs.py).
  * tests/dynamics_model/test_lti_system.cpp:45-136 -- x+ of the 4-state / 2-control system against the printed x_true (1e-4), the
    Jacobian convention (A - I) / dt, B / dt against A_true, B_true (1e-3);
  * tests/dynamics_model/test_pendulum.cpp:27-73 -- 500 rk4 steps of the damped pendulum from pi / 4: mgl (1 + cos theta) falls;
  * tests/dynamics_model/test_manipulator.cpp:75-90 -- gravity accelerates the second joint at q2 = pi / 4 (state_dot(4) != 0);
  * tests/dynamics_model/test_unicycle.cpp:27-67 -- fifty euler steps at v = 1, omega = 0.5 from the origin (the reference asserts dimensions only; the
    closed form of the euler recursion is asserted here).

Every plant is evaluated three ways where it exists three ways: the oracle (C++), the numpy twin, and the product's host evaluation
(cddp_hip_model_eval of libcddp_hip.so -- host code, runs without a GPU).  The end-effector kinematics of test_manipulator.cpp:27-73
(getEndEffectorPosition) are not on the solver path and are not restated."""
import json
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "oracle", "twin"))


def _cases():
    return json.load(open(os.path.join(HERE, "golden", "ref_boxqp_inputs.json")))["cases"]


@pytest.mark.parametrize("case", _cases(), ids=lambda c: c["name"])
def test_boxqp_on_the_reference_inputs(api, oracle_built, case):
    import cddp_twin as T
    n = case["n"]
    H = np.array(case["Q"]).reshape(n, n); g = np.array(case["q"]); lo = np.array(case["lower"]); up = np.array(case["upper"])
    x0 = np.zeros(n)
    x, st, free, it, fc = api.oracle_boxqp(H, g, lo, up, x0)
    assert st in (4, 5), st                                  # SUCCESS / ALL_CLAMPED (boxqp.hpp:40-50)
    xt, stt, freet, _ = T.boxqp(H, g, lo, up, x0, T.default_options())
    assert stt == {4: "SUCCESS", 5: "ALL_CLAMPED"}[int(st)] and np.array_equal(np.asarray(freet, bool), free.astype(bool))
    assert np.max(np.abs(x - xt)) <= 1e-13
    assert np.all(x >= lo) and np.all(x <= up)
    grad = g + H @ x
    for i in range(n):
        if free[i]:
            assert abs(grad[i]) < 1e-7
        else:
            assert (x[i] == lo[i] and grad[i] > 0) or (x[i] == up[i] and grad[i] < 0)
    y = np.clip(x0, lo, up); L = np.linalg.eigvalsh(H).max()
    for _ in range(20000):
        y = np.clip(y - (g + H @ y) / L, lo, up)
    assert np.allclose(x, y, atol=1e-7)
    if n == 5:   # by hand: the unconstrained minimiser is (-0.2, 1.8, 0, 0.6, 1); with x0 clamped at 0 the rest solves to (5/3, 0, 2/3, 1)
        assert np.allclose(x, [0.0, 5.0 / 3.0, 0.0, 2.0 / 3.0, 1.0], rtol=0, atol=1e-9)


def _lti():
    from scipy.linalg import expm
    dt = 0.01
    A0 = np.array([[0, 0.2473, -0.7933, 0.3470], [-0.2473, 0, -0.7667, 2.1307], [0.7933, 0.7667, 0, 0.3154], [-0.3470, -2.1307, -0.3154, 0]])
    B0 = np.array([[-0.6387, -0.2026], [-0.4049, -0.1975], [2.3939, 1.5163], [-0.0496, -1.7322]])
    return expm(dt * A0), dt * B0, dt


def test_lti_known_answers(api, oracle_built):
    import cddp_twin as T
    A, B, dt = _lti()
    x = np.array([0.8378, 0.3794, 1.4796, 0.2382]); u = np.array([0.01, 0.01])
    x_true = np.array([0.8277, 0.3708, 1.4902, 0.2225])
    A_true = np.array([[1.0000, 0.0024, -0.0079, 0.0035], [-0.0025, 0.9997, -0.0077, 0.0213], [0.0079, 0.0076, 0.9999, 0.0032],
                       [-0.0035, -0.0213, -0.0031, 0.9998]])
    B_true = np.array([[-0.0064, -0.0020], [-0.0040, -0.0020], [0.0239, 0.0152], [-0.0005, -0.0173]])
    tw = T.LTI(A, B, dt)
    p = api.Problem(api.SOLVER_CLDDP, api.MODEL_LTI, api.EULER, 4, 2, 4, dt, np.eye(4), np.eye(2), np.eye(4), np.zeros(4), lti_A=A, lti_B=B,
                    options=api.default_options())
    _, xn, Fx, Fu = api.Oracle(p).dynamics(x, u)
    for nxt, (jx, ju) in ((tw.step(x, u, 0.0), tw.jac(x, u, 0.0)), (xn, (Fx, Fu))):
        assert np.allclose(nxt, A @ x + B @ u, rtol=0, atol=1e-15)                # test_lti_system.cpp:73-75
        assert np.linalg.norm(nxt - x_true) < 1e-4                                # :77-79
        assert np.linalg.norm((jx * dt + np.eye(4)) - A_true) < 1e-3              # :118-134
        assert np.linalg.norm(ju * dt - B_true) < 1e-3                            # :121-135


def test_pendulum_energy_falls_under_damping(api, oracle_built):
    import cddp_twin as T
    dt, length, mass, damping = 0.01, 1.0, 1.0, 0.1
    tw = T.Pendulum(length, mass, damping)
    p = api.Problem(api.SOLVER_CLDDP, api.MODEL_PENDULUM, api.RK4, 2, 1, 4, dt, np.eye(2), np.eye(1), np.eye(2), np.zeros(2),
                    model_params=[length, mass, damping, 9.81], options=api.default_options())
    orc = api.Oracle(p)
    u = np.zeros(1)
    xs = {"twin": np.array([np.pi / 4, 0.0]), "oracle": np.array([np.pi / 4, 0.0]), "host": np.array([np.pi / 4, 0.0])}
    last_theta = dict.fromkeys(xs)
    for i in range(500):
        for k in xs: last_theta[k] = xs[k][0]           # the reference stores theta BEFORE the step; its check reads the last stored one
        xs["twin"] = T.discrete_step(tw, "rk4", dt, xs["twin"], u, 0.0)
        xs["oracle"] = orc.dynamics(xs["oracle"], u)[1]
        xs["host"] = api.model_eval(api.MODEL_PENDULUM, api.RK4, dt, [length, mass, damping, 9.81], 2, 1, xs["host"], u)["step"]
    e0 = 9.81 * (1.0 + np.cos(np.pi / 4))
    for k in xs:
        assert 9.81 * (1.0 + np.cos(last_theta[k])) < e0, k                      # test_pendulum.cpp:70-73
    assert np.max(np.abs(xs["twin"] - xs["oracle"])) < 1e-9
    assert np.max(np.abs(xs["host"] - xs["oracle"])) < 1e-9


def test_manipulator_gravity_accelerates_the_second_joint(api, oracle_built):
    p = api.manipulator_problem(api.SOLVER_CLDDP, horizon=4, constrained=False)
    x = np.zeros(6); x[1] = np.pi / 4; u = np.zeros(3)
    xd, xn, _, _ = api.Oracle(p).dynamics(x, u)
    assert abs(xd[4]) > 0.0                                                       # test_manipulator.cpp:86-89
    host = api.model_eval(api.MODEL_MANIPULATOR, api.RK4, p.dt, list(p.model_params) if hasattr(p, "model_params") else [], 6, 3, x, u)["step"]
    assert np.max(np.abs(host - xn)) < 1e-12 and abs(host[4]) > 0.0


def test_unicycle_euler_steps(api, oracle_built):
    import cddp_twin as T
    dt = 0.1                                                                      # test_unicycle.cpp:29-31
    p = api.Problem(api.SOLVER_CLDDP, api.MODEL_UNICYCLE, api.EULER, 3, 2, 4, dt, np.eye(3), np.eye(2), np.eye(3), np.zeros(3),
                    options=api.default_options())
    orc = api.Oracle(p); tw = T.Unicycle()
    u = np.array([1.0, 0.5])
    xo = np.zeros(3); xt = xo.copy(); xh = xo.copy(); xc = xo.copy()             # :37-44: from the origin, 50 steps
    for _ in range(50):
        xc = xc + dt * np.array([u[0] * np.cos(xc[2]), u[0] * np.sin(xc[2]), u[1]])
        xo = orc.dynamics(xo, u)[1]
        xt = T.discrete_step(tw, "euler", dt, xt, u, 0.0)
        xh = api.model_eval(api.MODEL_UNICYCLE, api.EULER, dt, [], 3, 2, xh, u)["step"]
    for v in (xo, xt, xh):
        assert np.max(np.abs(v - xc)) < 1e-12
    assert abs(xc[2] - 2.5) < 1e-12
