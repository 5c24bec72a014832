"""cddp_hip_model_eval (csrc/host_models.cpp): the built-in plants' model source compiled for the HOST, which is what lets a
built-in DynamicalSystem be paired with user Objective / Constraint subclasses in the plug-in solve.  Checked on CPU against the
oracle's independently written plants (oracle/models.hpp): discrete step, continuous-time Jacobians and Hessian tensors at seeded
points, in both trig builds (default: the host libm on both sides; parity build: the shared straight-line routines on both sides).
Bitwise for the plants whose device source mirrors the oracle's expression trees; the cart-pole's default build uses hand-derived
Jacobians (1e-12, see dev_models.hpp) and dual numbers in the parity build (bitwise)."""
import numpy as np
import pytest


def _cases(api):
    return [
        ("pendulum", api.pendulum_problem(api.SOLVER_IPDDP, True), True),
        ("cartpole", api.cartpole_problem(api.SOLVER_IPDDP, True), True),
        ("unicycle", api.unicycle_problem(api.SOLVER_IPDDP, 20, True), True),
        ("quadrotor", api.quadrotor_problem(api.SOLVER_IPDDP, 10, True), True),     # host-only Hessians (17-seed second-order duals)
        ("quad12", api.quadrotor12_problem(api.SOLVER_IPDDP, 10, True), True),       # round 4: dual2nd default on the plant's expression
        ("manipulator", api.manipulator_problem(api.SOLVER_IPDDP, 10), True),      # closed-form cross Hessian vs the oracle's dual2nd LU
        ("manip7", api.manipulator7_problem(api.SOLVER_IPDDP, 10), True),
        ("bicycle", api.bicycle_problem(api.SOLVER_IPDDP, 10), True),
        ("bicycle_rk4", api.bicycle_problem(api.SOLVER_IPDDP, 10, integrator=api.RK4), True),
        ("car", api.car_problem(api.SOLVER_IPDDP, 10), True),
        ("hcw", api.hcw_problem(api.SOLVER_IPDDP, 10), True),
    ]


def test_host_model_eval_matches_the_oracle(api, oracle_built, trig="libm"):
    # HOST code of the library (host_models.cpp, the plug-in route) calls the host libm -- glibc, the reference's own arithmetic --
    # so it is compared with the oracle in its default mode; the DEVICE code runs the shared straight-line routines (round 4).
    rng = np.random.default_rng(20260929)
    for name, p, has_hess in _cases(api):
        o = api.Oracle(p)
        mp = np.array(list(p.c.model_params), dtype=np.float64)
        for k in range(8):
            x = rng.normal(0.0, 0.7, p.nx); u = rng.normal(0.0, 1.0, p.nu)
            if name == "quadrotor":
                x[3:7] = [1.0, 0.1 * x[4], 0.1 * x[5], 0.1 * x[6]]
            if trig == "shared":
                with api.shared_trig():
                    _, xn, Fx, Fu = o.dynamics(x, u)
                    H = o.hessians(x, u) if has_hess else None
            else:
                _, xn, Fx, Fu = o.dynamics(x, u)
                H = o.hessians(x, u) if has_hess else None
            r = api.model_eval(p.c.model, p.c.integrator, p.dt, mp, p.nx, p.nu, x, u, want=("step", "jac") + (("hess",) if has_hess else ()))
            fx, fu = r["jac"]
            if name in ("quad12", "manip7") and trig == "libm":
                # the two synthetic plants evaluate the straight-line sincos in EVERY build (dev_trig.hpp: trig_n); the oracle's default
                # mode is the host libm: <= 1 ulp per sine
                assert np.max(np.abs(r["step"] - xn)) < 1e-13 and np.max(np.abs(fx - Fx)) < 1e-11 and np.max(np.abs(fu - Fu)) < 1e-11
                continue
            assert np.array_equal(r["step"], xn), (name, trig, k, np.max(np.abs(r["step"] - xn)))
            if name == "cartpole" and trig == "libm":
                assert np.max(np.abs(fx - Fx)) < 1e-11 and np.max(np.abs(fu - Fu)) < 1e-11
            else:
                assert np.array_equal(fx, Fx) and np.array_equal(fu, Fu), (name, trig, k, np.max(np.abs(fx - Fx)), np.max(np.abs(fu - Fu)))
            if has_hess:
                for a, b in zip(r["hess"], H):
                    assert np.max(np.abs(a - b)) <= 1e-10 * max(1.0, max(np.max(np.abs(c)) for c in H)), (name, trig, k, np.max(np.abs(a - b)))


def test_host_model_eval_errors(api):
    p = api.pendulum_problem(api.SOLVER_IPDDP, True)
    mp = np.array(list(p.c.model_params), dtype=np.float64)
    with pytest.raises(api.HipError, match="dimensions do not match"):
        api.model_eval(p.c.model, p.c.integrator, p.dt, mp, 3, 1, np.zeros(3), np.zeros(1))
    with pytest.raises(api.HipError, match="no host evaluation"):
        api.model_eval(api.MODEL_LTI, p.c.integrator, p.dt, mp, 2, 1, np.zeros(2), np.zeros(1))
    # (every built-in plant has Hessian tensors since round 4: the 'no Hessian tensors' refusal has no subject left)
    with pytest.raises(api.HipError, match="Integration type not supported"):
        api.model_eval(p.c.model, 9, p.dt, mp, 2, 1, np.zeros(2), np.zeros(1))
