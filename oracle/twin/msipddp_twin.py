"""MSIPDDP backward pass without path constraints, restated in numpy (test infrastructure, never imported by the product).

MSIPDDPSolver::backwardPass, src/cddp_core/msipddp_solver.cpp:1112-1208 (cddp-cpp v0.5.2): the IPDDP recursion over a
multiple-shooting trajectory whose nodes need not satisfy the dynamics -- the defect d_t = f(x_t, u_t) - x_{t+1} enters
through V_x + V_xx d_t in Q_x and Q_u; Q_uu is symmetrised, THEN regularised, factored with Eigen's LDLT and kept (regularised)
in the value update; the costate gains are k_lambda = -lambda_t + V_x + V_xx d_t, K_lambda = sym(V_xx) of step t + 1.
Gauss-Newton only (the use_ilqr = false terms weigh the dynamics Hessians with the costates, :1151-1163 -- not restated).

The path-constrained branch of the same function (:1222-1420) is NOT restated: its line 1398 adds an (nx x nu) product to the
(nu x nx) block Q_ux, which defines nothing for nx != nu.

Checker of the stack-fed branch CDDP_HIP_STACKS_MSIPDDP (tests/test_logddp_stack_fed.py)."""
import numpy as np

from cddp_twin import EigenLDLT


def backward(A, B, lx, lu, lxx, luu, lux, VxN, VxxN, d, lam, reg):
    """Returns ok, K, k, Vx (N+1), Vxx (N+1), dV, inf_du, step_norm, inf_defect, k_lambda (N), K_lambda (N)."""
    N = len(A); nx = A[0].shape[0]; nu = B[0].shape[1]
    V_x = np.array(VxN, float)
    V_xx = 0.5 * (np.array(VxxN, float) + np.array(VxxN, float).T)
    K = np.zeros((N, nu, nx)); k = np.zeros((N, nu)); Vx = np.zeros((N + 1, nx)); Vxx = np.zeros((N + 1, nx, nx))
    kl = np.zeros((N, nx)); Kl = np.zeros((N, nx, nx))
    Vx[N] = V_x; Vxx[N] = V_xx
    dV = np.zeros(2); inf_du = step_norm = inf_defect = 0.0
    for t in range(N - 1, -1, -1):
        w = V_x + V_xx @ d[t]
        Q_x = lx[t] + A[t].T @ w
        Q_u = lu[t] + B[t].T @ w
        Q_xx = lxx[t] + A[t].T @ V_xx @ A[t]
        Q_ux = lux[t] + B[t].T @ V_xx @ A[t]
        Q_uu = luu[t] + B[t].T @ V_xx @ B[t]
        Q_uu = 0.5 * (Q_uu + Q_uu.T)
        Q_uu[np.diag_indices(nu)] += reg
        f = EigenLDLT(Q_uu)
        if not f.ok:
            return (False, K, k, Vx, Vxx, dV, inf_du, step_norm, inf_defect, kl, Kl)
        k_u = -f.solve(Q_u); K_u = -f.solve(Q_ux)
        k[t] = k_u; K[t] = K_u
        kl[t] = -lam[t] + V_x + V_xx @ d[t]
        Kl[t] = 0.5 * (V_xx + V_xx.T)
        V_x_new = Q_x + K_u.T @ Q_u + Q_ux.T @ k_u + K_u.T @ Q_uu @ k_u
        V_xx_new = Q_xx + K_u.T @ Q_ux + Q_ux.T @ K_u + K_u.T @ Q_uu @ K_u
        V_x, V_xx = V_x_new, 0.5 * (V_xx_new + V_xx_new.T)
        dV = dV + np.array([float(k_u @ Q_u), 0.5 * float(k_u @ (Q_uu @ k_u))])
        Vx[t] = V_x; Vxx[t] = V_xx
        inf_du = max(inf_du, float(np.max(np.abs(Q_u)))); step_norm = max(step_norm, float(np.max(np.abs(k_u))))
        inf_defect = max(inf_defect, float(np.max(np.abs(d[t]))))
    return (True, K, k, Vx, Vxx, dV, inf_du, step_norm, inf_defect, kl, Kl)
