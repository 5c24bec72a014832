"""numpy twin, terminal-equality branch of IPDDPSolver::backwardPass -- TEST INFRASTRUCTURE (see cddp_twin.py).

Follows /root/reference/src/cddp_core/ipddp_solver.cpp:413-476 (solveSequentialLQR), :478-639
(solveTerminalEqualityLQR: (p + 1) LQR sweeps + linear rollouts, regularised normal equations over five scales) and
:1120-1353 (the LQ model with the path constraints condensed in).  Written from the reference alone.
"""
import math

import numpy as np

from cddp_twin import EPS_SLACK, MAX_BARRIER_RATIO, EigenLDLT, clip_pos, clip_sgn, sym


def sequential_lqr(Q, q, R, r, M, A, B, d):
    """solveSequentialLQR (:413-476). Returns (ok, K, k, P, p)."""
    T = len(R)
    n = Q[0].shape[0]; m = R[0].shape[0]
    K = [np.zeros((m, n)) for _ in range(T)]; k = [np.zeros(m) for _ in range(T)]
    P = [np.zeros((n, n)) for _ in range(T + 1)]; p = [np.zeros(n) for _ in range(T + 1)]
    P[T] = 0.5 * (Q[T] + Q[T].T); p[T] = q[T].copy()
    for t in range(T - 1, -1, -1):
        Pn, pn = P[t + 1], p[t + 1]
        BtP = B[t].T @ Pn
        Q_uu = 0.5 * (R[t] + BtP @ B[t] + R[t].T + B[t].T @ Pn.T @ B[t])
        Q_ux = BtP @ A[t] + M[t].T
        Q_xu = Q_ux.T
        drift = pn + Pn @ d[t]
        Q_x = q[t] + A[t].T @ drift
        Q_u = r[t] + B[t].T @ drift
        f = EigenLDLT(Q_uu)
        if not f.ok:
            return False, K, k, P, p
        K[t] = -f.solve(Q_ux); k[t] = -f.solve(Q_u)
        Pt = Q[t] + A[t].T @ Pn @ A[t] + Q_xu @ K[t] + K[t].T @ Q_ux + K[t].T @ Q_uu @ K[t]
        P[t] = 0.5 * (Pt + Pt.T)
        p[t] = Q_x + Q_xu @ k[t] + K[t].T @ Q_u + K[t].T @ Q_uu @ k[t]
        if not (np.all(np.isfinite(P[t])) and np.all(np.isfinite(p[t])) and np.all(np.isfinite(K[t])) and np.all(np.isfinite(k[t]))):
            return False, K, k, P, p
    return True, K, k, P, p


def rollout_linear(A, B, d, K, k, dx0):
    """rolloutLinearPolicy (:368-392)."""
    T = len(K)
    dX = [np.zeros_like(dx0) for _ in range(T + 1)]; dU = [None] * T
    dX[0] = dx0.copy()
    for t in range(T):
        dU[t] = k[t] + K[t] @ dX[t]
        dX[t + 1] = A[t] @ dX[t] + B[t] @ dU[t] + d[t]
    return dX, dU


def terminal_equality_lqr(Q, q, R, r, M, A, B, d, dx0, H_T, b_T, mu, reg_scale, reg_exponent, lambda_prev):
    """solveTerminalEqualityLQR (:478-639). Returns (ok, K, k, P, p, lambda_total, lambda_delta)."""
    p_dim = H_T.shape[0]
    q_base = [v.copy() for v in q]
    lam_prev = np.zeros(p_dim)
    if lambda_prev is not None and lambda_prev.size == p_dim:
        lam_prev = lambda_prev.copy()
        q_base[-1] = q_base[-1] + H_T.T @ lam_prev
    T = len(R)
    Kv, kv, Pv, pv, xT = [], [], [], [], []
    for i in range(p_dim + 1):
        qv = [v.copy() for v in q_base]
        if i > 0:
            qv[-1] = qv[-1] + H_T[i - 1, :]
        ok, K_, k_, P_, p_ = sequential_lqr(Q, qv, R, r, M, A, B, d)
        if not ok:
            return False, None, None, None, None, None, None
        dX, _ = rollout_linear(A, B, d, K_, k_, dx0)
        Kv.append(K_); kv.append(k_); Pv.append(P_); pv.append(p_); xT.append(dX[-1])
    n = Q[0].shape[0]
    S_mat = np.zeros((n, p_dim))
    for i in range(p_dim):
        S_mat[:, i] = xT[i + 1] - xT[0]
    A_small = H_T @ S_mat
    rhs = b_T - H_T @ xT[0]
    AtA = A_small.T @ A_small
    Atb = A_small.T @ rhs
    tr = float(np.trace(AtA))
    trace_term = tr / max(p_dim, 1) if tr > 1.0 else 1.0
    base_floor = max(1e-10, reg_scale * math.pow(max(mu, 0.0), reg_exponent))
    reg = max(base_floor, 1e-6 * trace_term)
    sing = np.linalg.svd(A_small, compute_uv=False)            # JacobiSVD singular values (:566-569)
    smax = float(np.max(sing)) if sing.size else 0.0
    smin = float(np.min(sing)) if sing.size else 0.0
    svd_reg = max(1e-8 * smax - smin, 0.0)
    reg_base = max(reg, svd_reg)
    cap = 100.0 * (1.0 + float(np.linalg.norm(rhs)))
    best = np.zeros(p_dim); best_res = float("inf"); found = False
    for scale in (1.0, 10.0, 100.0, 1e3, 1e4):
        reg_i = max(reg_base * scale, 1e-12)
        f = EigenLDLT(AtA + reg_i * np.eye(p_dim))
        if not f.ok:
            continue
        lam = f.solve(Atb)
        if not np.all(np.isfinite(lam)):
            continue
        ln = float(np.linalg.norm(lam))
        if ln > cap:
            lam = lam * (cap / max(ln, 1e-12))
        res = float(np.linalg.norm(A_small @ lam - rhs))
        if not math.isfinite(res):
            continue
        if (not found) or res < best_res:
            best, best_res, found = lam, res, True
    if not found:
        best = np.zeros(p_dim)
    K_out = [v.copy() for v in Kv[0]]; k_out = [v.copy() for v in kv[0]]
    P_out = [v.copy() for v in Pv[0]]; p_out = [v.copy() for v in pv[0]]
    for i in range(p_dim):
        c = best[i]
        for t in range(T):
            k_out[t] = k_out[t] + c * (kv[i + 1][t] - kv[0][t])
        for t in range(T + 1):
            p_out[t] = p_out[t] + c * (pv[i + 1][t] - pv[0][t])
    return True, K_out, k_out, P_out, p_out, lam_prev + best, best


def backward_term_eq(tw, AB, GJ, V_x, V_xx, hT, HT, inf_pr, inf_comp):
    """The has_terminal_eq branch of IPDDPSolver::backwardPass (:1120-1353) on the Twin instance `tw`."""
    N, nx, nu, m, mu, o = tw.N, tw.nx, tw.nu, tw.m, tw.mu, tw.o
    Q = [np.zeros((nx, nx)) for _ in range(N + 1)]; q = [np.zeros(nx) for _ in range(N + 1)]
    R = [None] * N; r = [None] * N; M = [None] * N
    A = [AB[t][0] for t in range(N)]; B = [AB[t][1] for t in range(N)]; d = [np.zeros(nx) for _ in range(N)]
    Q[N] = V_xx; q[N] = V_x
    models = [None] * N
    for t in range(N):
        lx, lu, lxx, luu, lux = tw.cost_derivs(t)
        Q[t] = sym(lxx); q[t] = lx.copy(); R[t] = sym(luu); r[t] = lu.copy(); M[t] = lux.T.copy()
        if not o["use_ilqr"]:            # :1160-1178: the current costate iterate stands in for the value gradient
            lam = tw.Lam[t + 1] if (tw.Lam.shape == (N + 1, nx) and np.all(np.isfinite(tw.Lam[t + 1]))) else np.zeros(nx)
            Fxx, Fuu, Fux = tw.hess_stack(t)
            for i in range(nx):
                Q[t] = Q[t] + lam[i] * Fxx[i]; M[t] = M[t] + lam[i] * Fux[i].T; R[t] = R[t] + lam[i] * Fuu[i]
            Q[t] = sym(Q[t]); R[t] = sym(R[t])
        if m > 0:
            Q_yx, Q_yu = GJ[t]
            y, s, g = tw.Y[t], tw.S[t], tw.G[t]
            s_safe = np.maximum(s, max(mu * 1e-3, EPS_SLACK))
            YS = np.array([clip_pos(y[i], s_safe[i]) for i in range(m)])
            rp = g + s; rc = y * s - mu; rhat = y * rp - rc
            Sir = np.array([clip_sgn(rhat[i], s_safe[i]) for i in range(m)])
            q[t] = q[t] + Q_yx.T @ (y + Sir)
            r[t] = r[t] + Q_yu.T @ (y + Sir)
            Q[t] = Q[t] + Q_yx.T @ np.diag(YS) @ Q_yx
            M[t] = M[t] + (Q_yu.T @ np.diag(YS) @ Q_yx).T
            R[t] = R[t] + Q_yu.T @ np.diag(YS) @ Q_yu
            Q[t] = sym(Q[t]); R[t] = sym(R[t])
            models[t] = (y, s, Q_yx, Q_yu, YS, rp, rhat, s_safe)
            inf_pr = max(inf_pr, float(np.max(np.abs(rp)))); inf_comp = max(inf_comp, float(np.max(np.abs(rc))))
        R[t] = R[t].copy(); R[t][np.diag_indices(nu)] += tw.reg
    ok, K, k, P, p, lam_total, lam_delta = terminal_equality_lqr(
        Q, q, R, r, M, A, B, d, np.zeros(nx), HT, -hT, mu, o["jacobian_regularization_value"],
        o["jacobian_regularization_exponent"], tw.LamT)
    if not ok:
        return False
    for t in range(N):
        tw.K_u[t] = K[t]; tw.k_u[t] = k[t]
    for t in range(N + 1):
        tw.Vxx[t] = P[t]; tw.Vx[t] = p[t]
    tw.dLamT = lam_delta
    inf_du = 0.0; step_norm = 0.0
    for t in range(N):
        Q_u = r[t] + B[t].T @ p[t + 1]
        inf_du = max(inf_du, float(np.max(np.abs(Q_u)))); step_norm = max(step_norm, float(np.max(np.abs(k[t]))))
    dX, dU = rollout_linear(A, B, d, K, k, np.zeros(nx))
    tw.dX = np.array(dX); tw.dU = np.array(dU)
    if m > 0:
        for t in range(N):
            y, s, Q_yx, Q_yu, YS, rp, rhat, s_safe = models[t]
            temp = Q_yu @ k[t]
            tw.k_y[t] = np.array([clip_sgn(rhat[i] + y[i] * temp[i], s_safe[i]) for i in range(m)])
            tw.K_y[t] = np.clip(np.diag(YS) @ (Q_yx + Q_yu @ K[t]), -MAX_BARRIER_RATIO, MAX_BARRIER_RATIO)
            tw.k_s[t] = -rp - temp
            tw.K_s[t] = -Q_yx - Q_yu @ K[t]
            tw.dS[t] = tw.k_s[t] + tw.K_s[t] @ dX[t]
            tw.dY[t] = np.clip(tw.k_y[t] + tw.K_y[t] @ dX[t], -MAX_BARRIER_RATIO, MAX_BARRIER_RATIO)
    for name, At, bt in tw.term_ineq:
        gT = At @ tw.X[N] - bt; ST, YT = tw.S_T[name], tw.Y_T[name]
        rp = gT + ST; rd = ST * YT - mu
        tw.dS_T[name] = -rp - At @ dX[N]
        dYT = np.zeros(gT.size)
        for i in range(gT.size):
            ss = max(ST[i], max(mu * 1e-3, EPS_SLACK))
            ratio = min(max(YT[i] / ss, 0.0), MAX_BARRIER_RATIO)
            aff = min(max(-rd[i] / ss, -MAX_BARRIER_RATIO), MAX_BARRIER_RATIO)
            dYT[i] = min(max(aff - ratio * tw.dS_T[name][i], -MAX_BARRIER_RATIO), MAX_BARRIER_RATIO)
        tw.dY_T[name] = dYT
    tw.dV = np.zeros(2)
    tw.inf_pr, tw.inf_du, tw.inf_comp, tw.step_norm = inf_pr, inf_du, inf_comp, step_norm
    return True
