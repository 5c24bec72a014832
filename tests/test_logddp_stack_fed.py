"""f4, first slice (LogDDP; the unconstrained MSIPDDP recursion at the end of the file): the LogDDP backward pass (LogDDPSolver::backwardPass, logddp_solver.cpp:470-575, with the relaxed log
barrier of barrier.hpp:37-262) on the GPU through the stack-fed boundary -- branch CDDP_HIP_STACKS_LOGDDP.  A host LogDDP
solver keeps its outer loop and forward pass; it folds the barrier gradients / Hessians into the cost stacks and hands the
(N x batch) stacks over.

CPU: the numpy restatement (oracle/twin/logddp_twin.py) against finite differences of its own barrier value and against a
plain discrete Riccati recursion.  GPU: the HIP sweep against that restatement on pendulum / cart-pole / unicycle rollouts
with control (and state) boxes, inside and outside the relaxation zone, incl. the regularisation-retry loop."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.host_arithmetic   # host route of the library: glibc on both sides (tests/conftest.py)

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "oracle", "twin"))

TOL = 1e-9


def rel(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b))))


def _case(name, seed):
    """plant, dt, integrator, Q, R, Qf, x_ref, constraints, rollout (X, U) with controls partly OUTSIDE the relaxation zone."""
    import cddp_twin as T
    import logddp_twin as Lg
    rng = np.random.default_rng(seed)
    if name == "pendulum":
        m = T.Pendulum(0.5, 1.0, 0.01); dt, integ, N = 0.05, "euler", 40
        Q, R, Qf = np.diag([0.1, 0.05]), np.diag([0.02]), np.diag([50.0, 5.0]); xref = np.zeros(2)
        cons = [Lg.BoxRows(T.ControlBox([-2.0], [2.0]))]
        x0 = np.array([np.pi, 0.0]) + rng.uniform(-0.2, 0.2, 2); U = rng.uniform(-2.2, 2.2, (N, 1))      # some |u| > bound
    elif name == "cartpole":
        m = T.CartPole(1.0, 0.2, 0.5, 9.81, 0.0); dt, integ, N = 0.05, "rk4", 50
        Q, R, Qf = np.diag([0.1, 0.1, 0.05, 0.05]), np.diag([0.1]), np.diag([100.0, 100.0, 10.0, 10.0]); xref = np.array([0.0, np.pi, 0.0, 0.0])
        cons = [Lg.BoxRows(T.ControlBox([-5.0], [5.0]))]
        x0 = rng.uniform(-0.2, 0.2, 4); U = rng.uniform(-4.99, 4.99, (N, 1))
    else:
        m = T.Unicycle(); dt, integ, N = 0.03, "euler", 60
        Q, R, Qf = np.diag([0.0, 0.0, 0.0]), np.diag([0.05, 0.05]), np.diag([100.0, 100.0, 50.0]); xref = np.array([2.0, 2.0, np.pi / 2])
        cons = [Lg.BoxRows(T.ControlBox([-1.1, -np.pi], [1.1, np.pi])), Lg.BoxRows(T.StateBox([-0.5, -0.5, -4.0], [3.0, 3.0, 4.0]))]
        x0 = rng.uniform(-0.1, 0.1, 3); U = np.column_stack([rng.uniform(0.2, 1.05, N), rng.uniform(-0.5, 0.5, N)])
    X = np.zeros((N + 1, m.nx)); X[0] = x0
    for t in range(N): X[t + 1] = T.discrete_step(m, integ, dt, X[t], U[t], t * dt)
    A = []; B = []
    for t in range(N):
        Fx, Fu = m.jac(X[t], U[t], t * dt)
        A.append(np.eye(m.nx) + dt * Fx); B.append(dt * Fu)
    Qd, Rd = Q * dt, R * dt
    lx = np.array([2.0 * Qd @ (X[t] - xref) for t in range(N)]); lu = np.array([2.0 * Rd @ U[t] for t in range(N)])
    lxx = np.tile(2.0 * Qd, (N, 1, 1)); luu = np.tile(2.0 * Rd, (N, 1, 1)); lux = np.zeros((N, m.nu, m.nx))
    VxN = 2.0 * Qf @ (X[N] - xref); VxxN = 2.0 * Qf
    hess = [tuple(dt * h for h in m.hess(X[t], U[t], t * dt)) for t in range(N)]      # dt-scaled F_xx, F_uu, F_ux (cddp_solver_base.cpp:346-356)
    return dict(A=np.array(A), B=np.array(B), lx=lx, lu=lu, lxx=lxx, luu=luu, lux=lux, VxN=VxN, VxxN=VxxN, cons=cons, X=X, U=U, nx=m.nx, nu=m.nu, N=N, hess=hess)


def test_relaxed_log_barrier_derivatives_match_finite_differences():
    import cddp_twin as T
    import logddp_twin as Lg
    cons = [Lg.BoxRows(T.ControlBox([-1.0, -2.0], [1.0, 2.0])), Lg.BoxRows(T.StateBox([-0.5, -0.5, -4.0], [3.0, 3.0, 4.0]))]
    coeff, delta = 0.3, 0.1
    for x, u in ((np.array([0.2, -0.1, 0.3]), np.array([0.3, -1.0])),                 # deep inside: -log branch everywhere
                 (np.array([2.95, -0.47, 0.3]), np.array([0.97, 2.1]))):              # inside the relaxation zone and beyond the bound
        gx = np.zeros(3); gu = np.zeros(2); Hxx = np.zeros((3, 3)); Huu = np.zeros((2, 2)); Hux = np.zeros((2, 3))
        for c in cons:
            a, b = Lg.barrier_gradients(c, x, u, coeff, delta); gx += a; gu += b
            a, b, cc = Lg.barrier_hessians(c, x, u, coeff, delta); Hxx += a; Huu += b; Hux += cc
        h = 1e-6
        val = lambda xx, uu: Lg.barrier_value(cons, xx, uu, coeff, delta)
        for i in range(3):
            e = np.zeros(3); e[i] = h
            assert abs((val(x + e, u) - val(x - e, u)) / (2 * h) - gx[i]) < 1e-5 * max(1.0, abs(gx[i]))
        for i in range(2):
            e = np.zeros(2); e[i] = h
            assert abs((val(x, u + e) - val(x, u - e)) / (2 * h) - gu[i]) < 1e-5 * max(1.0, abs(gu[i]))
        # linear constraints: the barrier Hessian is exactly the derivative of its gradient (no constraint-curvature term)
        for i in range(2):
            e = np.zeros(2); e[i] = h
            gp = sum(Lg.barrier_gradients(c, x, u + e, coeff, delta)[1] for c in cons); gm = sum(Lg.barrier_gradients(c, x, u - e, coeff, delta)[1] for c in cons)
            assert np.max(np.abs((gp - gm) / (2 * h) - Huu[:, i])) < 1e-4 * max(1.0, np.max(np.abs(Huu)))
        assert np.max(np.abs(Hux)) == 0.0      # control box and state box do not couple


def test_logddp_backward_without_constraints_is_the_discrete_riccati_recursion():
    import logddp_twin as Lg
    rng = np.random.default_rng(3)
    nx, nu, N = 3, 2, 12
    A = [np.eye(nx) + 0.1 * rng.standard_normal((nx, nx)) for _ in range(N)]; B = [0.3 * rng.standard_normal((nx, nu)) for _ in range(N)]
    Q = np.diag([1.0, 2.0, 0.5]); R = np.diag([0.3, 0.7]); Qf = np.diag([5.0, 5.0, 1.0])
    z = np.zeros
    ok, K, k, Vx, Vxx, dV, _ = Lg.backward(A, B, z((N, nx)), z((N, nu)), np.tile(Q, (N, 1, 1)), np.tile(R, (N, 1, 1)), z((N, nu, nx)), z(nx), Qf, [], None, None, 0.0, 0.1, 0.0)
    assert ok
    P = Qf.copy()
    for t in range(N - 1, -1, -1):
        Kt = -np.linalg.solve(R + B[t].T @ P @ B[t], B[t].T @ P @ A[t])
        assert np.max(np.abs(K[t] - Kt)) < 1e-10
        P = Q + A[t].T @ P @ A[t] + A[t].T @ P @ B[t] @ Kt
        assert np.max(np.abs(Vxx[t] - 0.5 * (P + P.T))) < 1e-9
    assert np.max(np.abs(k)) == 0.0 and np.max(np.abs(dV)) == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["pendulum", "cartpole", "unicycle"])
def test_hip_logddp_sweep_matches_the_restatement(api, name):
    import logddp_twin as Lg
    Bn = 6
    coeff, delta = 0.5, 0.1
    cases = [_case(name, 100 + b) for b in range(Bn)]
    c0 = cases[0]
    hs = api.HipStackSolver(Bn, c0["nx"], c0["nu"], 0, c0["N"])
    folded = [Lg.folded_cost_stacks(c["lx"], c["lu"], c["lxx"], c["luu"], c["lux"], c["cons"], c["X"], c["U"], coeff, delta) for c in cases]
    st = lambda key: np.stack([c[key] for c in cases])
    hs.set_stacks(st("A"), st("B"), np.stack([f[0] for f in folded]), np.stack([f[1] for f in folded]), np.stack([f[2] for f in folded]),
                  np.stack([f[3] for f in folded]), np.stack([f[4] for f in folded]), st("VxN"), st("VxxN"))
    opt = api.default_options()
    for reg0 in (0.0, 1e-6, 1e-2):
        reg = np.full(Bn, reg0)
        ok = hs.backward(api.STACKS_LOGDDP, opt, reg, None, retry=False)
        K, k, Vx, Vxx, dV = hs.gains(); sc = hs.scalars()
        for b, c in enumerate(cases):
            okr, Kr, kr, Vxr, Vxxr, dVr, qu = Lg.backward(c["A"], c["B"], c["lx"], c["lu"], c["lxx"], c["luu"], c["lux"], c["VxN"], c["VxxN"],
                                                        c["cons"], c["X"], c["U"], coeff, delta, reg0)
            assert bool(ok[b]) == okr, (name, reg0, b)
            if not okr: continue
            for nm, got, ref in (("K", K[b], Kr), ("k", k[b], kr), ("Vx", Vx[b], Vxr), ("Vxx", Vxx[b], Vxxr), ("dV", dV[b], dVr)):
                assert rel(got, ref) < TOL, (name, reg0, b, nm, rel(got, ref))
            assert rel(sc["inf_du"][b], qu) < TOL          # raw max |Q_u| (logddp_solver.cpp:572), no CLDDP scaling
    # use_ilqr = false (logddp_solver.cpp:505-515): the dt-scaled Hessian tensors through cddp_hip_set_hessian_stacks
    hs.set_hessian_stacks(np.array([[h[0] for h in c["hess"]] for c in cases]), np.array([[h[1] for h in c["hess"]] for c in cases]),
                          np.array([[h[2] for h in c["hess"]] for c in cases]))
    ok = hs.backward(api.STACKS_LOGDDP, opt, np.full(Bn, 1e-6), None, retry=False)
    K, k, Vx, Vxx, dV = hs.gains()
    for b, c in enumerate(cases):
        okr, Kr, kr, Vxr, Vxxr, dVr, qu = Lg.backward(c["A"], c["B"], c["lx"], c["lu"], c["lxx"], c["luu"], c["lux"], c["VxN"], c["VxxN"],
                                                    c["cons"], c["X"], c["U"], coeff, delta, 1e-6, hess=c["hess"])
        assert bool(ok[b]) == okr
        if okr: assert max(rel(K[b], Kr), rel(k[b], kr), rel(Vx[b], Vxr), rel(Vxx[b], Vxxr)) < 1e-6, (name, b)   # (indefinite Q_uu under full DDP: gains 1e3, measured 2e-8)
    hs.set_hessian_stacks(None, None, None)
    # the retry loop of the outer solver (cddp_solver_base.cpp:93-111) with an indefinite start: l_uu made negative
    bad = np.stack([f[3] for f in folded]).copy(); bad[:, :, 0, 0] -= 50.0
    hs.set_stacks(luu=bad)
    reg = np.full(Bn, 1e-6)
    ok = hs.backward(api.STACKS_LOGDDP, opt, reg, None, retry=True)
    sc = hs.scalars()
    assert np.all(sc["reg"] >= 1e-6)
    K, k, Vx, Vxx, dV = hs.gains()
    for b, c in enumerate(cases):
        luu_b = c["luu"].copy(); luu_b[:, 0, 0] -= 50.0
        okr = Lg.backward(c["A"], c["B"], c["lx"], c["lu"], c["lxx"], luu_b, c["lux"], c["VxN"], c["VxxN"], c["cons"], c["X"], c["U"], coeff, delta, float(sc["reg"][b]))
        assert bool(ok[b]) == okr[0]
        if okr[0]: assert rel(K[b], okr[1]) < 1e-7 and rel(Vxx[b], okr[4]) < 1e-7      # (indefinite blocks: gains up to 1e3)
    hs.close()


def test_msipddp_backward_without_defects_is_the_discrete_riccati_recursion():
    import msipddp_twin as Ms
    rng = np.random.default_rng(5)
    nx, nu, N = 3, 2, 10
    A = [np.eye(nx) + 0.1 * rng.standard_normal((nx, nx)) for _ in range(N)]; B = [0.3 * rng.standard_normal((nx, nu)) for _ in range(N)]
    Q = np.diag([1.0, 2.0, 0.5]); R = np.diag([0.3, 0.7]); Qf = np.diag([5.0, 5.0, 1.0]); z = np.zeros
    out = Ms.backward(A, B, z((N, nx)), z((N, nu)), np.tile(Q, (N, 1, 1)), np.tile(R, (N, 1, 1)), z((N, nu, nx)), z(nx), Qf, z((N, nx)), z((N, nx)), 0.0)
    assert out[0]
    P = Qf.copy()
    for t in range(N - 1, -1, -1):
        Kt = -np.linalg.solve(R + B[t].T @ P @ B[t], B[t].T @ P @ A[t])
        assert np.max(np.abs(out[1][t] - Kt)) < 1e-10
        P = Q + A[t].T @ P @ A[t] + A[t].T @ P @ B[t] @ Kt
    # a defect moves the feed-forward term only: k = -(R + B^T P B)^-1 B^T P d at the last step
    d = z((N, nx)); d[N - 1] = np.array([0.1, -0.2, 0.05])
    out2 = Ms.backward(A, B, z((N, nx)), z((N, nu)), np.tile(Q, (N, 1, 1)), np.tile(R, (N, 1, 1)), z((N, nu, nx)), z(nx), Qf, d, z((N, nx)), 0.0)
    kref = -np.linalg.solve(R + B[N - 1].T @ Qf @ B[N - 1], B[N - 1].T @ Qf @ d[N - 1])
    assert np.max(np.abs(out2[2][N - 1] - kref)) < 1e-10 and np.max(np.abs(out2[1] - out[1])) < 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["pendulum", "cartpole", "unicycle"])
def test_hip_msipddp_sweep_matches_the_restatement(api, name):
    """Branch CDDP_HIP_STACKS_MSIPDDP (msipddp_solver.cpp:1112-1208): defects through cddp_hip_set_defect_stack; the costate gains
    follow on the host from the returned value stacks (:1192-1194)."""
    import msipddp_twin as Ms
    Bn = 5
    rng = np.random.default_rng(77)
    cases = [_case(name, 300 + b) for b in range(Bn)]
    c0 = cases[0]; N, nx = c0["N"], c0["nx"]
    dfc = [0.05 * rng.standard_normal((N, nx)) for _ in range(Bn)]
    lam = [rng.standard_normal((N, nx)) for _ in range(Bn)]
    hs = api.HipStackSolver(Bn, nx, c0["nu"], 0, N)
    st = lambda key: np.stack([c[key] for c in cases])
    hs.set_stacks(st("A"), st("B"), st("lx"), st("lu"), st("lxx"), st("luu"), st("lux"), st("VxN"), st("VxxN"))
    opt = api.default_options()
    with pytest.raises(api.HipError):
        hs.backward(api.STACKS_MSIPDDP, opt, np.full(Bn, 1e-6))          # defect stack missing
    hs.set_defect_stack(np.stack(dfc))
    for reg0 in (0.0, 1e-3):
        ok = hs.backward(api.STACKS_MSIPDDP, opt, np.full(Bn, reg0), None, retry=False)
        K, k, Vx, Vxx, dV = hs.gains(); sc = hs.scalars()
        for b, c in enumerate(cases):
            ref = Ms.backward(c["A"], c["B"], c["lx"], c["lu"], c["lxx"], c["luu"], c["lux"], c["VxN"], c["VxxN"], dfc[b], lam[b], reg0)
            assert bool(ok[b]) == ref[0]
            if not ref[0]: continue
            for nm, got, want in (("K", K[b], ref[1]), ("k", k[b], ref[2]), ("Vx", Vx[b], ref[3]), ("Vxx", Vxx[b], ref[4]), ("dV", dV[b], ref[5])):
                assert rel(got, want) < TOL, (name, reg0, b, nm, rel(got, want))
            assert rel(sc["inf_du"][b], ref[6]) < TOL and rel(sc["step_norm"][b], ref[7]) < TOL
            kl = np.array([-lam[b][t] + Vx[b][t + 1] + Vxx[b][t + 1] @ dfc[b][t] for t in range(N)])      # host side, :1192
            assert rel(kl, ref[9]) < TOL and rel(Vxx[b][1:], ref[10]) < TOL
    # without defects the branch IS the unconstrained IPDDP sweep
    hs.set_defect_stack(np.zeros((Bn, N, nx)))
    hs.backward(api.STACKS_MSIPDDP, opt, np.full(Bn, 1e-6)); K1 = hs.gains()[0]
    hs.backward(api.STACKS_IPDDP, opt, np.full(Bn, 1e-6)); K2 = hs.gains()[0]
    assert np.array_equal(K1, K2)
    hs.close()
