"""Stack-fed mode (`cddp_hip_backward_stacks`, include/cddp_hip.h): the host evaluates the plug-ins and hands
over the (N x batch) derivative stacks; the GPU runs the Riccati sweep on them.  Checked against the device
solver's own sweep on the same iterate (which tests/test_gpu_parity.py checks against the oracle) and against
the oracle's gains directly.  Tolerance 1e-9 relative: the stacks are assembled in numpy, not in the kernels'
association order."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-9


def rel(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b))))


def build_stacks(api, p, X, U):
    """Host-side plug-in evaluation: A = I + dt f_x, B = dt f_u from the oracle's model, quadratic objective."""
    B, N = U.shape[0], U.shape[1]
    nx, nu, dt = p.nx, p.nu, p.dt
    o = api.Oracle(p)
    fx = np.zeros((B, N, nx, nx)); fu = np.zeros((B, N, nx, nu))
    for b in range(B):
        for t in range(N):
            _, _, Fx, Fu = o.dynamics(X[b, t], U[b, t])
            fx[b, t] = dt * Fx + np.eye(nx); fu[b, t] = dt * Fu
    Qd, Rd = p.Q * dt, p.R * dt
    lx = np.einsum("ij,btj->bti", 2.0 * Qd, X[:, :N] - p.x_ref[None, None, :])
    lu = np.einsum("ij,btj->bti", 2.0 * Rd, U)
    lxx = np.broadcast_to(2.0 * Qd, (B, N, nx, nx)).copy()
    luu = np.broadcast_to(2.0 * Rd, (B, N, nu, nu)).copy()
    lux = np.zeros((B, N, nu, nx))
    VxN = np.einsum("ij,bj->bi", 2.0 * p.Qf, X[:, N] - p.x_ref[None, :])
    H = 2.0 * p.Qf
    VxxN = np.broadcast_to(0.5 * (H + H.T), (B, nx, nx)).copy()
    return fx, fu, lx, lu, lxx, luu, lux, VxN, VxxN


@pytest.mark.parametrize("case", ["pendulum_ipddp", "cartpole_ipddp", "cartpole_clddp", "unicycle_clddp"])
def test_stack_fed_sweep_matches_solver_sweep(api, case):
    S = api
    if case == "pendulum_ipddp":
        p = S.pendulum_problem(S.SOLVER_IPDDP, False); reg_in_value = 1
    elif case == "cartpole_ipddp":
        p = S.cartpole_problem(S.SOLVER_IPDDP, False); reg_in_value = 1
    elif case == "cartpole_clddp":
        p = S.cartpole_problem(S.SOLVER_CLDDP, False); reg_in_value = 0
    else:
        p = S.unicycle_problem(S.SOLVER_CLDDP, 60, False); p._cons = []; p._rebuild(); reg_in_value = 0
    B = 6
    x0 = S.batch_x0(p, B, 20260930)
    U0 = S.batch_U0(p, B) if hasattr(S, "batch_U0") else None
    hs = S.HipBatchSolver(p, B)
    hs.set_initial(x0, U0); hs.initialize()
    ok = hs.backward()
    assert ok.all()
    X, U = hs.trajectory()
    K, k = hs.gains(); Vx, Vxx = hs.value(); dV, reg = hs.backward_scalars()
    hs.close()
    assert np.all(reg == reg[0])
    stacks = build_stacks(S, p, X, U)
    K2, k2, Vx2, Vxx2, dV2, ok2, ms = S.hip_backward_stacks(*stacks, reg=float(reg[0]), reg_in_value=reg_in_value)
    assert ok2.all()
    assert rel(K2, K) < TOL and rel(k2, k) < TOL
    assert rel(Vx2, Vx) < TOL and rel(Vxx2, Vxx) < TOL
    assert rel(dV2, dV) < TOL
    # and against the oracle directly (trajectory 0)
    o = S.Oracle(p); o.set_initial(x0[0], None if U0 is None else U0[0]); o.initialize()
    assert o.backward()
    Ko, ko = o.gains()
    assert rel(K2[0], Ko) < 1e-8 and rel(k2[0], ko) < 1e-8


def test_stack_fed_rejects_uninstantiated_shape(api):
    B, N, nx, nu = 2, 5, 5, 2   # (5, 2) is not an instantiated (nx, nu) pair
    z = lambda *s: np.zeros(s)
    with pytest.raises(api.HipError):
        api.hip_backward_stacks(z(B, N, nx, nx), z(B, N, nx, nu), z(B, N, nx), z(B, N, nu), z(B, N, nx, nx), z(B, N, nu, nu),
                                z(B, N, nu, nx), z(B, nx), z(B, nx, nx), reg=1e-6, reg_in_value=1)
