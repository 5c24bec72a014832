"""Rows a7 / a21 / a22 / a23 directly: the device's rollout step and the linearisation stacks A_t = I + dt f_x, B_t = dt f_u
(precomputeDynamicsDerivatives, cddp_solver_base.cpp:319-394; the plants' Jacobian sources dynamics_model/*.cpp) against the
oracle's evaluation of the same plant at the same (x_t, u_t), through cddp_hip_get_linearization."""
import numpy as np
import pytest

from test_gpu_parity import make, spread_for

CASES = ["pendulum_ipddp_box", "cartpole_ipddp_box", "unicycle_ipddp_box_ball", "quadrotor_ipddp_box", "quad12_ipddp_box",
         "manipulator_ipddp_box", "manip7_ipddp_box"]
# central finite differences with h = 2e-5 (manipulator.cpp:53-70) amplify the last bit of f by 1 / (2h) = 2.5e4
JAC_TOL = {"manipulator_ipddp_box": 1e-8}


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_rollout_step_and_linearization_match_the_oracle(api, oracle_built, name):
    p = make(api, name)
    B = 5
    x0 = api.batch_x0(p, B, 20261101, spread_for(p) if p.nx > 1 else 0.05 * np.ones(1))
    U0 = api.batch_U0(p, B)
    hs = api.HipBatchSolver(p, B); hs.set_initial(x0, U0); hs.initialize(); hs.backward()
    X, U = hs.trajectory(); A, Bm = hs.linearization(); hs.close()
    o = api.Oracle(p)
    worst = {"step": 0.0, "A": 0.0, "B": 0.0}
    scale = lambda r: np.maximum(1.0, np.abs(r))
    for b in range(B):
        for t in range(p.N):
            _, xn, Fx, Fu = o.dynamics(X[b, t], U[b, t], t * p.dt)
            Ao = np.eye(p.nx) + p.dt * Fx
            worst["step"] = max(worst["step"], float(np.max(np.abs(X[b, t + 1] - xn) / scale(xn))))
            worst["A"] = max(worst["A"], float(np.max(np.abs(A[b, t] - Ao) / scale(Ao))))
            worst["B"] = max(worst["B"], float(np.max(np.abs(Bm[b, t] - p.dt * Fu) / scale(Fu))))
    print(name, worst)
    assert worst["step"] < 1e-13, worst
    tol = JAC_TOL.get(name, 1e-12)
    assert worst["A"] < tol and worst["B"] < tol, worst
