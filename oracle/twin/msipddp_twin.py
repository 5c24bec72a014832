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


# ==================================================================================================================
# Full MSIPDDP solver (second restatement of src/cddp_core/msipddp_solver.cpp:33-1930 on the CDDPSolverBase loop,
# src/cddp_core/cddp_solver_base.cpp:29-186), written against the reference, not against oracle/cddp_oracle.cpp.
# Two properties of the reference are kept as they are:
#   * the unconstrained backward pass caches one LDLT per step and refactors a step only while its cached factor is invalid
#     (:1169-1185) -- from the second sweep on every step is solved with the factor of its FIRST sweep;
#   * the constrained backward pass adds the (nx x nu) product Q_yx^T (Y S^-1) Q_yu to the (nu x nx) block Q_ux (:1398): with
#     nu = 1 both have the same linear layout (the transpose lands), with nx = nu it is an untransposed elementwise add, any
#     other shape reads past the end of Q_yu in the reference and is refused here.
# ==================================================================================================================
import math


class MSIPDDP:
    """spec: the dictionary cddp_twin.Twin takes, plus options ms_costate_var_init_scale (1e-6), ms_segment_length (5),
    ms_rollout_type ("nonlinear" | "hybrid" | anything else), ms_use_controlled_rollout (False), warm_start (False)."""

    def __init__(self, spec):
        import cddp_twin as T
        import logddp_twin as L
        self.T = T
        self.model = spec["model"]; self.integrator = spec["integrator"]; self.dt = spec["dt"]; self.N = spec["N"]
        self.Qdt = np.array(spec["Q"], float) * self.dt; self.Rdt = np.array(spec["R"], float) * self.dt; self.Qf = np.array(spec["Qf"], float)
        self.xref = np.array(spec["xref"], float)
        self.cons = [spec["constraints"][k] for k in sorted(spec.get("constraints", {}))]
        self.rows = [L.ConRows(c) for c in self.cons]
        o = T.default_options()
        o.update(dict(ms_costate_var_init_scale=1e-6, ms_segment_length=5, ms_rollout_type="nonlinear", ms_use_controlled_rollout=False, warm_start=False))
        o.update(spec.get("options", {})); self.o = o
        self.nx, self.nu = self.model.nx, self.model.nu
        self.m = sum(c.dim for c in self.cons)
        if self.cons and not (self.nu == 1 or self.nx == self.nu):
            raise ValueError("MSIPDDP with path constraints is only defined for nu = 1 or nx = nu (msipddp_solver.cpp:1398)")
        self.ldlt = [None] * self.N                      # workspace_.ldlt_solvers / ldlt_valid
        self.history = []

    # ---------------------------------------------------------------- problem pieces
    def run_cost(self, x, u):
        e = x - self.xref
        return float(e @ self.Qdt @ e) + float(u @ self.Rdt @ u)

    def term_cost(self, x):
        e = x - self.xref
        return float(e @ self.Qf @ e)

    def step(self, x, u, t):
        return self.T.discrete_step(self.model, self.integrator, self.dt, x, u, t * self.dt)

    def g_all(self, x, u):
        return np.concatenate([c.g(x, u) for c in self.cons]) if self.cons else np.zeros(0)

    def jac_all(self, x, u):
        gx = [c.jac(x, u)[0] for c in self.cons]; gu = [c.jac(x, u)[1] for c in self.cons]
        return np.vstack(gx), np.vstack(gu)

    def set_initial(self, x0, U0=None, X0=None):
        self.x0 = np.array(x0, float)
        self.U = np.zeros((self.N, self.nu)) if U0 is None else np.array(U0, float).reshape(self.N, self.nu)
        self.X = np.tile(self.x0, (self.N + 1, 1)) if X0 is None else np.array(X0, float).reshape(self.N + 1, self.nx)
        self.X[0] = self.x0
        self.cost = math.inf; self.merit = math.inf     # CDDP::initializeProblemIfNecessary (cddp_core.cpp:297-301)

    # ---------------------------------------------------------------- initialisation
    def init_pair(self, g):
        o = self.o
        s = np.maximum(o["slack_var_init_scale"], -g)
        y = np.where(s < 1e-12, self.mu / 1e-12, self.mu / np.where(s < 1e-12, 1.0, s))
        y = np.maximum(o["dual_var_init_scale"] * 0.01, np.minimum(y, o["dual_var_init_scale"] * 100.0))
        return s, y

    def evaluate_trajectory(self):                       # :425-455
        c = 0.0
        self.X[0] = self.x0
        for t in range(self.N):
            c += self.run_cost(self.X[t], self.U[t])
            self.G[t] = self.g_all(self.X[t], self.U[t])
            self.F[t] = self.step(self.X[t], self.U[t], t)
            self.X[t + 1] = self.F[t]
        self.cost = c + self.term_cost(self.X[-1])

    def evaluate_trajectory_warm(self):                  # :457-495
        c = 0.0
        self.G = np.zeros((self.N, self.m))
        for t in range(self.N):
            c += self.run_cost(self.X[t], self.U[t])
            self.G[t] = self.g_all(self.X[t], self.U[t])
            self.F[t] = self.step(self.X[t], self.U[t], t)
            if self.o["ms_use_controlled_rollout"]:
                self.X[t + 1] = self.F[t]
        self.cost = c + self.term_cost(self.X[-1])

    def zero_gains(self):
        N, m, nx = self.N, self.m, self.nx
        self.k_y = np.zeros((N, m)); self.k_s = np.zeros((N, m)); self.K_y = np.zeros((N, m, nx)); self.K_s = np.zeros((N, m, nx))
        self.k_l = np.zeros((N, nx)); self.K_l = np.zeros((N, nx, nx))

    def reset_filter(self):                              # resetBarrierFilter :711-763
        mf = self.cost; ipr = fcv = icomp = idef = 0.0
        if self.cons:
            for t in range(self.N):
                off = 0
                for c in self.cons:
                    sl = slice(off, off + c.dim); off += c.dim
                    s = self.S[t, sl]; g = self.G[t, sl]; y = self.Y[t, sl]
                    acc = 0.0
                    for v in s: acc += math.log(v) if v > 0 else (-math.inf if v == 0 else math.nan)
                    mf -= self.mu * acc
                    pr = g + s
                    ipr = max(ipr, float(np.max(np.abs(pr)))); fcv += float(np.sum(np.abs(pr)))
                    icomp = max(icomp, float(np.max(np.abs(y * s - self.mu))))
                dres = self.F[t] - self.X[t + 1]
                idef = max(idef, float(np.max(np.abs(dres)))); fcv += float(np.sum(np.abs(dres)))
        self.inf_pr = max(ipr, idef); self.merit = mf; self.inf_comp = icomp
        self.filter = [(mf, fcv)]

    def initialize(self):                                # :33-264
        o = self.o; N, nx, nu, m = self.N, self.nx, self.nu, self.m
        a = o["ls_initial_step_size"]; self.alphas = []
        for _ in range(o["ls_max_iterations"]): self.alphas.append(a); a *= o["ls_step_reduction_factor"]
        self.alpha_pr = o["ls_initial_step_size"]; self.alpha_du = 0.0
        self.n_backward = self.n_forward = 0
        self.inf_du = math.inf
        self.seg = o["ms_segment_length"]
        self.K = np.zeros((N, nu, nx)); self.k = np.zeros((N, nu)); self.dV = np.zeros(2)
        self.Lam = o["ms_costate_var_init_scale"] * np.ones((N, nx))
        self.F = np.zeros((N, nx))
        self.zero_gains()
        if o["warm_start"]:                              # no earlier solve on this object: the "else" branch :108-160
            if not self.cons:
                self.mu = 1e-8
                self.G = np.zeros((N, 0))
            else:
                self.evaluate_trajectory_warm()
                mv = float(np.max(self.G))
                mv = max(mv, 0.0)
                self.mu = o["tolerance"] * 0.01 if mv <= o["tolerance"] else (o["tolerance"] if mv <= 0.1 else o["mu_initial"] * 0.1)
            self.reg = o["reg_initial_value"]; self.step_norm = 0.0
            self.S = np.zeros((N, m)); self.Y = np.zeros((N, m))
            for t in range(N):
                if m: self.S[t], self.Y[t] = self.init_pair(self.G[t])
            self.reset_filter()
            return
        self.mu = 1e-8 if not self.cons else o["mu_initial"]
        self.G = np.zeros((N, m)); self.S = np.zeros((N, m)); self.Y = np.zeros((N, m))
        for t in range(N):                               # initializeDualSlackCostateVariables :643-709, on the GUESS trajectory
            if m:
                self.G[t] = self.g_all(self.X[t], self.U[t])
                self.S[t], self.Y[t] = self.init_pair(self.G[t])
        c = 0.0
        for t in range(N): c += self.run_cost(self.X[t], self.U[t])
        self.cost = c + self.term_cost(self.X[-1])
        self.reg = o["reg_initial_value"]; self.step_norm = 0.0
        self.evaluate_trajectory()
        self.reset_filter()

    # ---------------------------------------------------------------- backward pass :1112-1430
    def backward_pass(self):
        self.n_backward += 1
        N, nx, nu, m, dt, o = self.N, self.nx, self.nu, self.m, self.dt, self.o
        V_x = 2.0 * self.Qf @ (self.X[-1] - self.xref)
        V_xx = 2.0 * self.Qf; V_xx = 0.5 * (V_xx + V_xx.T)
        dV = np.zeros(2); idu = ipr = icomp = idef = snorm = 0.0
        for t in range(N - 1, -1, -1):
            x, u, lam = self.X[t], self.U[t], self.Lam[t]
            d = self.F[t] - self.X[t + 1]
            Fx, Fu = self.model.jac(x, u, t * dt)
            A = dt * Fx + np.eye(nx); B = dt * Fu
            lx = 2.0 * self.Qdt @ (x - self.xref); lu = 2.0 * self.Rdt @ u
            w = V_x + V_xx @ d
            if m:
                y, s, g = self.Y[t], self.S[t], self.G[t]
                Qyx, Qyu = self.jac_all(x, u)
                Q_x = lx + Qyx.T @ y + A.T @ w
                Q_u = lu + Qyu.T @ y + B.T @ w
            else:
                Q_x = lx + A.T @ w
                Q_u = lu + B.T @ w
            Q_xx = 2.0 * self.Qdt + A.T @ V_xx @ A
            Q_ux = np.zeros((nu, nx)) + B.T @ V_xx @ A
            Q_uu = 2.0 * self.Rdt + B.T @ V_xx @ B
            if not o["use_ilqr"]:
                Fxx, Fuu, Fux = self.model.hess(x, u, t * dt)
                for i in range(nx):
                    Q_xx = Q_xx + (dt * lam[i]) * Fxx[i]; Q_ux = Q_ux + (dt * lam[i]) * Fux[i]; Q_uu = Q_uu + (dt * lam[i]) * Fuu[i]
                off = 0
                for r in self.rows:
                    H = r.hessians(x, u)
                    if H is None: raise ValueError("constraint without Hessians under use_ilqr = false")
                    for i in range(r.con.dim):
                        Q_xx = Q_xx + y[off + i] * H[0][i]; Q_ux = Q_ux + y[off + i] * H[2][i]; Q_uu = Q_uu + y[off + i] * H[1][i]
                    off += r.con.dim
            if not m:
                Q_uu = 0.5 * (Q_uu + Q_uu.T)
                Q_uu[np.diag_indices(nu)] += self.reg
                if self.ldlt[t] is None:                 # need_recompute :1169-1176
                    self.ldlt[t] = self.T.EigenLDLT(Q_uu)
                if not self.ldlt[t].ok:
                    self.ldlt[t] = None
                    return False
                f = self.ldlt[t]
                k_u = -f.solve(Q_u); K_u = -f.solve(Q_ux)
                self.k[t] = k_u; self.K[t] = K_u
                self.k_l[t] = -lam + V_x + V_xx @ d
                self.K_l[t] = 0.5 * (V_xx + V_xx.T)
                V_x_n = Q_x + K_u.T @ Q_u + Q_ux.T @ k_u + K_u.T @ Q_uu @ k_u
                V_xx_n = Q_xx + K_u.T @ Q_ux + Q_ux.T @ K_u + K_u.T @ Q_uu @ K_u
                dV = dV + np.array([float(k_u @ Q_u), 0.5 * float(k_u @ (Q_uu @ k_u))])
            else:
                ys = y / s
                pres = g + s; cres = y * s - self.mu; rhat = y * pres - cres
                Qr = 0.5 * (Q_uu + Q_uu.T)
                Qr = Qr + (Qyu.T * ys) @ Qyu
                Qr[np.diag_indices(nu)] += self.reg
                f = self.T.EigenLDLT(Qr)
                if not f.ok:
                    return False
                sir = rhat / s
                rhs0 = Q_u + Qyu.T @ sir
                rhs1 = Q_ux + (Qyu.T * ys) @ Qyx
                k_u = -f.solve(rhs0); K_u = -f.solve(rhs1)
                self.k[t] = k_u; self.K[t] = K_u
                temp = Qyu @ k_u
                self.k_y[t] = (rhat + y * temp) / s
                self.K_y[t] = ys[:, None] * (Qyx + Qyu @ K_u)
                self.k_s[t] = -pres - temp
                self.K_s[t] = -Qyx - Qyu @ K_u
                self.k_l[t] = -lam + V_x + V_xx @ d
                self.K_l[t] = 0.5 * (V_xx + V_xx.T)
                Q_u = Q_u + Qyu.T @ sir
                Q_x = Q_x + Qyx.T @ sir
                Q_xx = Q_xx + (Qyx.T * ys) @ Qyx
                P = (Qyx.T * ys) @ Qyu                   # :1398
                Q_ux = Q_ux + (P.T if nu == 1 else P)
                Q_uu = Q_uu + (Qyu.T * ys) @ Qyu
                dV = dV + np.array([float(k_u @ Q_u), 0.5 * float(k_u @ (Q_uu @ k_u))])
                V_x_n = Q_x + K_u.T @ Q_u + Q_ux.T @ k_u + K_u.T @ Q_uu @ k_u
                V_xx_n = Q_xx + K_u.T @ Q_ux + Q_ux.T @ K_u + K_u.T @ Q_uu @ K_u
                ipr = max(ipr, float(np.max(np.abs(pres)))); icomp = max(icomp, float(np.max(np.abs(cres))))
            V_x, V_xx = V_x_n, 0.5 * (V_xx_n + V_xx_n.T)
            idu = max(idu, float(np.max(np.abs(Q_u)))); snorm = max(snorm, float(np.max(np.abs(k_u)))); idef = max(idef, float(np.max(np.abs(d))))
        self.dV = dV; self.inf_du = idu; self.step_norm = snorm
        if m: self.inf_pr = max(ipr, idef); self.inf_comp = icomp
        else: self.inf_pr = idef; self.inf_comp = 0.0
        return True

    # ---------------------------------------------------------------- forward pass :1432-1724
    def next_state(self, t, Fn, dx, a):
        boundary = self.seg > 1 and (t + 1) % self.seg == 0 and t + 1 < self.N
        if not boundary:
            return Fn
        rt = self.o["ms_rollout_type"]
        if rt == "nonlinear":
            return self.X[t + 1] + (Fn - self.F[t]) + a * (self.F[t] - self.X[t + 1])
        if rt == "hybrid":
            Fx, Fu = self.model.jac(self.X[t], self.U[t], t * self.dt)
            A = np.eye(self.nx) + self.dt * Fx; B = self.dt * Fu
            return self.X[t + 1] + (A + B @ self.K[t]) @ dx + a * (B @ self.k[t] + self.F[t] - self.X[t + 1])
        return Fn

    def forward_pass(self, a):
        self.n_forward += 1
        N, o = self.N, self.o
        tau = max(o["min_fraction_to_boundary"], 1.0 - self.mu)
        X = self.X.copy(); U = self.U.copy(); X[0] = self.x0
        F = self.F.copy(); Lam = self.Lam.copy()
        if not self.cons:
            cost = 0.0
            for t in range(N):
                dx = X[t] - self.X[t]
                U[t] = self.U[t] + a * self.k[t] + self.K[t] @ dx
                Lam[t] = self.Lam[t] + a * self.k_l[t] + self.K_l[t] @ dx
                F[t] = self.step(X[t], U[t], t)
                X[t + 1] = self.next_state(t, F[t], dx, a)
                cost += self.run_cost(X[t], U[t])
            cost += self.term_cost(X[-1])
            dJ = self.cost - cost
            expected = -a * (self.dV[0] + 0.5 * a * self.dV[1])
            ratio = dJ / expected if expected > 0.0 else math.copysign(1.0, dJ)
            if not ratio > 1e-6:
                return None
            return dict(X=X, U=U, F=F, Lam=Lam, cost=cost, merit=cost, cv=0.0, alpha=a, alpha_du=1.0, ip=False)
        S = self.S.copy(); dxs = np.zeros((N, self.nx))
        for t in range(N):
            dx = X[t] - self.X[t]; dxs[t] = dx
            s_new = self.S[t] + a * self.k_s[t] + self.K_s[t] @ dx
            if np.any(s_new < (1.0 - tau) * self.S[t]):
                return None
            S[t] = s_new
            U[t] = self.U[t] + a * self.k[t] + self.K[t] @ dx
            F[t] = self.step(X[t], U[t], t)
            X[t + 1] = self.next_state(t, F[t], dx, a)
        Y = None; adu = None
        for ay in self.alphas:
            Yt = self.Y.copy(); ok = True
            for t in range(N):
                y_new = self.Y[t] + ay * self.k_y[t] + self.K_y[t] @ dxs[t]
                if np.any(y_new < (1.0 - tau) * self.Y[t]):
                    ok = False; break
                Yt[t] = y_new
            if ok:
                Y = Yt; adu = ay; break
        if Y is None:
            return None
        for t in range(N):
            Lam[t] = self.Lam[t] + a * self.k_l[t] + self.K_l[t] @ dxs[t]
        cost = 0.0; merit = 0.0; cv = 0.0
        G = self.G.copy()
        for t in range(N):
            cost += self.run_cost(X[t], U[t])
            off = 0
            for c in self.cons:
                sl = slice(off, off + c.dim); off += c.dim
                G[t, sl] = c.g(X[t], U[t])
                acc = 0.0
                for v in S[t, sl]: acc += math.log(v) if v > 0 else (-math.inf if v == 0 else math.nan)
                merit -= self.mu * acc
                cv += float(np.sum(np.abs(G[t, sl] + S[t, sl])))
            cv += float(np.sum(np.abs(F[t] - X[t + 1])))
        cost += self.term_cost(X[-1]); merit += cost
        if not self.filter_acceptable(merit, cv, a * self.dV[0]):
            return None
        return dict(X=X, U=U, F=F, Lam=Lam, S=S, Y=Y, G=G, cost=cost, merit=merit, cv=cv, alpha=a, alpha_du=adu, ip=True)

    def filter_acceptable(self, mf, cv, expected):       # isFilterAcceptable :771-808
        o = self.o
        if not self.filter:
            return True
        for (fm, fv) in self.filter:
            if fm <= mf and fv <= cv:
                return False
        best_v = math.inf; best_m = math.inf
        for (fm, fv) in self.filter:
            if fv < best_v: best_v, best_m = fv, fm
        v_imp = cv < best_v * (1.0 - o["filter_violation_acceptance_threshold"])
        m_imp = mf < best_m - o["filter_merit_acceptance_threshold"] * cv
        if cv < o["filter_min_violation_for_armijo_check"] and expected < 0:
            return mf < best_m + o["filter_armijo_constant"] * expected
        if cv < 1e-6 and mf <= best_m * (1.0 + 1e-8):
            return True
        return v_imp or m_imp

    def filter_accept(self, mf, cv):                     # interior_point_utils.cpp:79-95
        for (fm, fv) in self.filter:
            if fm <= mf and fv <= cv:
                return False
        self.filter = [(fm, fv) for (fm, fv) in self.filter if not (mf <= fm and cv <= fv)]
        self.filter.append((mf, cv))
        return True

    def scaled_inf_du(self):                             # :1886-1930
        if not self.cons:
            return self.inf_du
        yn = float(np.sum(np.abs(self.Y))); sn = float(np.sum(np.abs(self.S)))
        mpn = self.Y.size + self.nu * self.N
        num = (yn + sn) / mpn if mpn > 0 else 0.0
        return self.inf_du / (max(100.0, num) / 100.0)

    def update_barrier(self, fp_success):                # :1751-1850
        o = self.o
        if not self.cons:
            return
        st = o["barrier_strategy"]
        if st == "MONOTONIC":
            self.mu = max(o["mu_min_value"], o["mu_update_factor"] * self.mu); self.reset_filter()
        elif st == "IPOPT":
            err = max(self.scaled_inf_du(), self.inf_pr, self.inf_comp)
            if err <= 10.0 * self.mu:
                self.mu = max(o["tolerance"] / 10.0, min(o["mu_update_factor"] * self.mu, self.mu ** o["mu_update_power"])); self.reset_filter()
        else:
            metric = max(self.scaled_inf_du(), self.inf_pr, self.inf_comp)
            thr = max(metric * 10.0, self.mu * 100.0) if self.mu < 1e-5 else max(o["mu_update_factor"] * self.mu, self.mu * 2.0)
            slow = fp_success and self.alpha_pr > 0 and metric < 1e-3
            if metric <= thr or slow:
                fac = o["mu_update_factor"]
                if self.mu > 1e-12:
                    r = metric / self.mu
                    if r < 0.01: fac = o["mu_update_factor"] * 0.1
                    elif r < 0.1: fac = o["mu_update_factor"] * 0.3
                    elif r < 0.5: fac = o["mu_update_factor"] * 0.6
                lin = fac * self.mu; sup = self.mu ** o["mu_update_power"]
                if slow and self.mu > o["tolerance"]: self.mu = min(lin, sup)
                else: self.mu = max(o["tolerance"] / 100.0, min(lin, sup))
                self.reset_filter()

    def record(self):
        self.history.append([self.cost, self.merit, self.alpha_pr, self.alpha_du, self.inf_du, self.inf_pr, self.inf_comp, self.mu, self.reg])

    def solve(self):                                     # CDDPSolverBase::solve with MSIPDDP's hooks
        o = self.o
        self.initialize()
        self.record()
        it = 0; status = "MaxIterationsReached"; converged = False
        while it < o["max_iterations"]:
            it += 1
            ok = False
            while not ok:
                ok = self.backward_pass()
                if not ok:
                    self.reg = min(self.reg * o["reg_update_factor"], o["reg_max_value"])
                    if self.reg >= o["reg_max_value"]:
                        status = "RegularizationLimitReached_NotConverged"; break
            if not ok:
                break
            best = None
            for a in self.alphas:
                r = self.forward_pass(a)
                if r is not None:
                    best = r; break
            if best is not None:
                dJ = self.cost - best["cost"]
                self.X, self.U, self.cost, self.merit, self.alpha_pr, self.alpha_du = best["X"], best["U"], best["cost"], best["merit"], best["alpha"], best["alpha_du"]
                if best["ip"]: self.Y, self.S, self.G = best["Y"], best["S"], best["G"]
                self.F, self.Lam = best["F"], best["Lam"]
                self.filter_accept(best["merit"], best["cv"])
                self.record()
                self.reg = max(self.reg / o["reg_update_factor"], o["reg_min_value"])
                metric = max(self.scaled_inf_du(), self.inf_pr, self.inf_comp)                    # checkConvergence :306-364
                if metric <= o["tolerance"]:
                    status = "OptimalSolutionFound"; converged = True
                elif abs(dJ) < o["acceptable_tolerance"] and it > 10 and self.inf_pr < math.sqrt(o["acceptable_tolerance"]) and self.inf_comp < math.sqrt(o["acceptable_tolerance"]):
                    status = "AcceptableSolutionFound"; converged = True
                elif it >= 1 and self.step_norm < o["tolerance"] * 10.0 and self.inf_pr < 1e-4:
                    status = "AcceptableSolutionFound"; converged = True
            else:                                        # handleForwardPassFailure :371-398
                needs = len(self.filter) > 5 or any(not (math.isfinite(fm) and math.isfinite(fv)) for (fm, fv) in self.filter)
                if needs and self.filter:
                    bv = min(self.filter, key=lambda p: p[1]); bm = min(self.filter, key=lambda p: p[0])
                    self.filter = [bv]
                    if abs(bm[1] - bv[1]) > 1e-12 or abs(bm[0] - bv[0]) > 1e-12: self.filter.append(bm)
                else:
                    self.reg = min(self.reg * o["reg_update_factor"], o["reg_max_value"])
                    if self.reg >= o["reg_max_value"]:
                        status = "RegularizationLimitReached_NotConverged"; break
            if converged:
                break
            self.update_barrier(best is not None)        # postIterationUpdate :366-369
        return dict(iterations=it, status=status, final_objective=self.cost, n_backward=self.n_backward, n_forward=self.n_forward, mu=self.mu)
