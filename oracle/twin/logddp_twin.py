"""LogDDP backward pass, restated in numpy (test infrastructure, like the rest of oracle/; never imported by the product).

What is restated, from cddp-cpp v0.5.2:
  * RelaxedLogBarrier  include/cddp-cpp/cddp_core/barrier.hpp:37-262 -- beta_delta(z) = -log z for z > delta, the quadratic
    extension 0.5 (((z - 2 delta) / delta)^2 - 1) - log delta otherwise (:247-262), summed over the finite bounds of a
    constraint L <= g(x, u) <= U with s_L = g - L, s_U = U - g; gradients (:95-135) and Gauss-Newton Hessians plus the
    constraint-Hessian term (:137-213; zero for the linear constraint kinds used here), all scaled by the barrier coefficient;
  * LogDDPSolver::backwardPass  src/cddp_core/logddp_solver.cpp:470-575 -- Q blocks from the cost derivatives and
    A = I + dt f_x, B = dt f_u, the barrier terms added per constraint, Q_uu + reg I then symmetrised, Eigen LDLT,
    [k | K] = -solve([Q_u | Q_ux]), dV, V_x, V_xx (un-regularised Q_uu; symmetrised), inf_du = max |Q_u|.

Purpose: the checker of the stack-fed LogDDP branch of the HIP library (CDDP_HIP_STACKS_LOGDDP, tests/test_logddp_stack_fed.py):
a host LogDDP solver keeps its own outer loop and forward pass and hands the (N x batch) stacks -- cost derivatives with the
barrier terms folded in -- to the GPU for the Riccati sweep."""
import math

import numpy as np

from cddp_twin import EigenLDLT


def beta(z, delta):
    """(beta, beta', beta'') of barrier.hpp:247-262."""
    if z > delta:
        if z <= 1e-12:
            return -math.log(1e-12), -1.0 / 1e-12, 1.0 / (1e-12 * 1e-12)
        return -math.log(z), -1.0 / z, 1.0 / (z * z)
    td = (z - 2.0 * delta) / delta
    return 0.5 * (td * td - 1.0) - math.log(delta), td / delta, 1.0 / (delta * delta)


class BoxRows:
    """One reference constraint object with finite upper bounds only (BoxConstraint / LinearConstraint convention,
    constraint.hpp:174-180, 270-276): raw value g(x, u), upper bound U, Jacobians; second derivatives vanish."""

    def __init__(self, con):
        self.con = con          # a cddp_twin constraint: .g() returns g_raw - U, .jac() the raw Jacobians

    def slack_upper(self, x, u):
        return -self.con.g(x, u)                      # U - g_raw

    def jac(self, x, u):
        return self.con.jac(x, u)


def barrier_value(cons, x, u, coeff, delta):          # evaluate, barrier.hpp:61-91
    tot = 0.0
    for c in cons:
        acc = 0.0
        for z in c.slack_upper(x, u): acc += beta(float(z), delta)[0]
        tot += coeff * acc
    return tot


def barrier_gradients(c, x, u, coeff, delta):         # getGradients, barrier.hpp:95-135
    Gx, Gu = c.jac(x, u)
    gx = np.zeros(x.size); gu = np.zeros(u.size)
    for i, z in enumerate(c.slack_upper(x, u)):
        d = 0.0
        d -= beta(float(z), delta)[1]                  # upper side: dCost/dg_i -= beta'(s_U)
        gx = gx + d * Gx[i, :]
        gu = gu + d * Gu[i, :]
    return coeff * gx, coeff * gu


def barrier_hessians(c, x, u, coeff, delta):          # getHessians, barrier.hpp:137-213 (linear constraints: term 2 is zero)
    Gx, Gu = c.jac(x, u)
    Hxx = np.zeros((x.size, x.size)); Huu = np.zeros((u.size, u.size)); Hux = np.zeros((u.size, x.size))
    for i, z in enumerate(c.slack_upper(x, u)):
        t1 = beta(float(z), delta)[2]
        Hxx = Hxx + t1 * np.outer(Gx[i, :], Gx[i, :])
        Huu = Huu + t1 * np.outer(Gu[i, :], Gu[i, :])
        Hux = Hux + t1 * np.outer(Gu[i, :], Gx[i, :])
    return coeff * Hxx, coeff * Huu, coeff * Hux


def backward(A, B, lx, lu, lxx, luu, lux, VxN, VxxN, cons, X, U, coeff, delta, reg, hess=None):
    """logddp_solver.cpp:470-575 for one trajectory; A[t] = I + dt f_x, B[t] = dt f_u; cons = list of BoxRows (may be empty).
    hess (use_ilqr = false, :505-515): per step the dt-scaled tensors (F_xx, F_uu, F_ux), each indexed by the output row i.
    Returns ok, K, k, Vx (N+1), Vxx (N+1), dV (2), inf_du."""
    N = len(A); nx = A[0].shape[0]; nu = B[0].shape[1]
    V_x = np.array(VxN, float)
    V_xx = 0.5 * (np.array(VxxN, float) + np.array(VxxN, float).T)
    K = np.zeros((N, nu, nx)); k = np.zeros((N, nu)); Vx = np.zeros((N + 1, nx)); Vxx = np.zeros((N + 1, nx, nx))
    Vx[N] = V_x; Vxx[N] = V_xx
    dV = np.zeros(2); qu_err = 0.0
    for t in range(N - 1, -1, -1):
        Q_x = lx[t] + A[t].T @ V_x
        Q_u = lu[t] + B[t].T @ V_x
        Q_xx = lxx[t] + A[t].T @ V_xx @ A[t]
        Q_ux = lux[t] + B[t].T @ V_xx @ A[t]
        Q_uu = luu[t] + B[t].T @ V_xx @ B[t]
        if hess is not None:
            Fxx, Fuu, Fux = hess[t]
            for i in range(nx):
                Q_xx = Q_xx + V_x[i] * Fxx[i]; Q_ux = Q_ux + V_x[i] * Fux[i]; Q_uu = Q_uu + V_x[i] * Fuu[i]
        for c in cons:
            gx, gu = barrier_gradients(c, X[t], U[t], coeff, delta)
            Q_x = Q_x + gx; Q_u = Q_u + gu
            Hxx, Huu, Hux = barrier_hessians(c, X[t], U[t], coeff, delta)
            Q_xx = Q_xx + Hxx; Q_uu = Q_uu + Huu; Q_ux = Q_ux + Hux
        Q_uu_reg = Q_uu.copy()
        Q_uu_reg[np.diag_indices(nu)] += reg
        Q_uu_reg = 0.5 * (Q_uu_reg + Q_uu_reg.T)
        f = EigenLDLT(Q_uu_reg)
        if not f.ok:
            return False, K, k, Vx, Vxx, dV, qu_err
        kK = -f.solve(np.concatenate([Q_u.reshape(nu, 1), Q_ux], axis=1))
        k_u = kK[:, 0]; K_u = kK[:, 1:]
        k[t] = k_u; K[t] = K_u
        dV = dV + np.array([float(Q_u @ k_u), 0.5 * float(k_u @ (Q_uu @ k_u))])
        V_x = Q_x + K_u.T @ Q_uu @ k_u + Q_ux.T @ k_u + K_u.T @ Q_u
        V_xx = Q_xx + K_u.T @ Q_uu @ K_u + Q_ux.T @ K_u + K_u.T @ Q_ux
        V_xx = 0.5 * (V_xx + V_xx.T)
        Vx[t] = V_x; Vxx[t] = V_xx
        qu_err = max(qu_err, float(np.max(np.abs(Q_u))))
    return True, K, k, Vx, Vxx, dV, qu_err


def folded_cost_stacks(lx, lu, lxx, luu, lux, cons, X, U, coeff, delta):
    """What a host LogDDP solver hands to cddp_hip_set_stacks: the cost derivatives with every constraint's barrier terms
    added (the association differs from backward() in the last bit only: (l + L) + A^T V instead of (l + A^T V) + L)."""
    N = len(lx)
    out = [np.array(a, float).copy() for a in (lx, lu, lxx, luu, lux)]
    for t in range(N):
        for c in cons:
            gx, gu = barrier_gradients(c, X[t], U[t], coeff, delta)
            Hxx, Huu, Hux = barrier_hessians(c, X[t], U[t], coeff, delta)
            out[0][t] += gx; out[1][t] += gu; out[2][t] += Hxx; out[3][t] += Huu; out[4][t] += Hux
    return out
