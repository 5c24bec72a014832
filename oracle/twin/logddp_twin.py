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


# ==================================================================================================================
# Full LogDDP solver (second restatement of src/cddp_core/logddp_solver.cpp:43-707 on the CDDPSolverBase loop,
# src/cddp_core/cddp_solver_base.cpp:29-186), written from the reference independently of the C++ checker.
# ==================================================================================================================
class ConRows:
    """A cddp_twin constraint seen through the Constraint interface RelaxedLogBarrier uses: upper slack U - g, Jacobians, and the
    second derivatives of the rows (None when the reference's getHessians throws std::logic_error)."""

    def __init__(self, con):
        self.con = con

    def slack_upper(self, x, u):
        return -self.con.g(x, u)

    def jac(self, x, u):
        return self.con.jac(x, u)

    def hessians(self, x, u):
        import cddp_twin as T
        c = self.con; nx, nu = x.size, u.size
        if isinstance(c, T.SecondOrderCone):
            return None                                   # constraint.hpp:772-786
        Hxx = np.zeros((c.dim, nx, nx)); Huu = np.zeros((c.dim, nu, nu)); Hux = np.zeros((c.dim, nu, nx))
        if isinstance(c, T.Ball):                         # :387-396
            d = c.c.size; Hxx[0, :d, :d] = -2.0 * c.scale * np.eye(d)
        if isinstance(c, T.ThrustMagnitude):              # :899-920, 1021-1042
            term = float(u @ u) + c.eps; den = term ** 1.5
            H = (term * np.eye(nu) - np.outer(u, u)) / den if den > np.finfo(float).tiny else np.zeros((nu, nu))
            if c.mn is None: Huu[0] = H
            else: Huu[0] = -H; Huu[1] = H
        return Hxx, Huu, Hux


def barrier_hessians_full(c, x, u, coeff, delta):       # getHessians incl. the constraint-curvature term (barrier.hpp:137-213)
    Gx, Gu = c.jac(x, u)
    H = c.hessians(x, u)
    Hxx = np.zeros((x.size, x.size)); Huu = np.zeros((u.size, u.size)); Hux = np.zeros((u.size, x.size))
    for i, z in enumerate(c.slack_upper(x, u)):
        _, b1, b2 = beta(float(z), delta)
        Hxx = Hxx + b2 * np.outer(Gx[i, :], Gx[i, :]); Huu = Huu + b2 * np.outer(Gu[i, :], Gu[i, :]); Hux = Hux + b2 * np.outer(Gu[i, :], Gx[i, :])
        if H is not None:
            Hxx = Hxx + (-b1) * H[0][i]; Huu = Huu + (-b1) * H[1][i]; Hux = Hux + (-b1) * H[2][i]
    return coeff * Hxx, coeff * Huu, coeff * Hux


class LogDDP:
    """spec: the dictionary cddp_twin.Twin takes (model, integrator, dt, N, Q, R, Qf, xref, constraints, options) plus the LogDDP
    options log_mu_initial / log_mu_min_value / log_mu_update_factor / log_relaxed_delta."""

    def __init__(self, spec):
        import cddp_twin as T
        self.T = T
        self.model = spec["model"]; self.integrator = spec["integrator"]; self.dt = spec["dt"]; self.N = spec["N"]
        self.Qdt = np.array(spec["Q"], float) * self.dt; self.Rdt = np.array(spec["R"], float) * self.dt; self.Qf = np.array(spec["Qf"], float)
        self.xref = np.array(spec["xref"], float)
        self.cons = [ConRows(spec["constraints"][k]) for k in sorted(spec.get("constraints", {}))]
        o = T.default_options(); o.update(dict(log_mu_initial=1.0, log_mu_min_value=1e-10, log_mu_update_factor=0.5, log_relaxed_delta=1e-10))
        o.update(spec.get("options", {})); self.o = o
        self.nx, self.nu = self.model.nx, self.model.nu
        self.history = []

    # -- objective (QuadraticObjective)
    def run_cost(self, x, u):
        e = x - self.xref
        return float(e @ self.Qdt @ e) + float(u @ self.Rdt @ u)

    def term_cost(self, x):
        e = x - self.xref
        return float(e @ self.Qf @ e)

    def step(self, x, u, t):
        return self.T.discrete_step(self.model, self.integrator, self.dt, x, u, t * self.dt)

    def set_initial(self, x0, U0=None):
        self.x0 = np.array(x0, float)
        self.U = np.zeros((self.N, self.nu)) if U0 is None else np.array(U0, float).reshape(self.N, self.nu)

    def evaluate_and_reset_filter(self):                # evaluateTrajectory + resetFilter (:316-361)
        c = 0.0
        for t in range(self.N): c += self.run_cost(self.X[t], self.U[t])
        c += self.term_cost(self.X[-1])
        self.cost = c
        self.reset_filter()

    def reset_filter(self):
        merit = self.cost; viol = 0.0
        for t in range(self.N):
            for c in self.cons:
                merit += barrier_value([c], self.X[t], self.U[t], self.mu, self.delta)
                for z in c.slack_upper(self.X[t], self.U[t]):
                    if -z > 0.0: viol += -z
        self.merit = merit; self.violation = viol; self.inf_pr = viol

    def initialize(self):                               # :45-205 cold start
        o = self.o
        self.X = np.zeros((self.N + 1, self.nx)); self.X[0] = self.x0
        for t in range(self.N): self.X[t + 1] = self.step(self.X[t], self.U[t], t)
        self.K = np.zeros((self.N, self.nu, self.nx)); self.k = np.zeros((self.N, self.nu))
        a = o["ls_initial_step_size"]; self.alphas = []
        for _ in range(o["ls_max_iterations"]): self.alphas.append(a); a *= o["ls_step_reduction_factor"]
        self.alpha_pr = o["ls_initial_step_size"]; self.dV = np.zeros(2); self.reg = o["reg_initial_value"]
        self.mu = o["log_mu_initial"]; self.delta = o["log_relaxed_delta"]
        self.inf_du = math.inf
        self.n_backward = self.n_forward = 0
        self.evaluate_and_reset_filter()

    def backward_pass(self):                            # :365-590
        self.n_backward += 1
        N, dt = self.N, self.dt
        A = []; B = []; hs = [] if not self.o["use_ilqr"] else None
        for t in range(N):
            Fx, Fu = self.model.jac(self.X[t], self.U[t], t * dt)
            A.append(dt * Fx + np.eye(self.nx)); B.append(dt * Fu)
            if hs is not None:
                Fxx, Fuu, Fux = self.model.hess(self.X[t], self.U[t], t * dt)
                hs.append((dt * Fxx, dt * Fuu, dt * Fux))
        lx = [2.0 * self.Qdt @ (self.X[t] - self.xref) for t in range(N)]; lu = [2.0 * self.Rdt @ self.U[t] for t in range(N)]
        lxx = [2.0 * self.Qdt] * N; luu = [2.0 * self.Rdt] * N; lux = [np.zeros((self.nu, self.nx))] * N
        ok, K, k, Vx, Vxx, dV, qe = backward_full(A, B, lx, lu, lxx, luu, lux, 2.0 * self.Qf @ (self.X[-1] - self.xref), 2.0 * self.Qf,
                                                  self.cons, self.X, self.U, self.mu, self.delta, self.reg, hs)
        if ok:
            self.K, self.k, self.Vx, self.Vxx, self.dV, self.inf_du = K, k, Vx, Vxx, dV, qe
        return ok

    def forward_pass(self, a):                          # :594-707
        self.n_forward += 1
        o = self.o
        X = self.X.copy(); U = self.U.copy(); X[0] = self.x0
        for t in range(self.N):
            U[t] = self.U[t] + a * self.k[t] + self.K[t] @ (X[t] - self.X[t])
            X[t + 1] = self.step(X[t], U[t], t)
            if not (np.all(np.isfinite(X[t + 1])) and np.all(np.isfinite(U[t]))):
                return None
        cost = 0.0; merit = 0.0; rp = 0.0
        for t in range(self.N):
            cost += self.run_cost(X[t], U[t])
            for c in self.cons:
                merit += barrier_value([c], X[t], U[t], self.mu, self.delta)
                for z in c.slack_upper(X[t], U[t]):
                    if -z > 0.0: rp += -z
        cost += self.term_cost(X[-1]); merit += cost
        cv_old, cv_new = self.violation, rp
        expected = a * self.dV[0]
        ok = False
        if cv_new > o["filter_max_violation_threshold"]:
            ok = cv_new < (1.0 - o["filter_violation_acceptance_threshold"]) * cv_old
        elif max(cv_new, cv_old) < o["filter_min_violation_for_armijo_check"] and expected < 0:
            ok = merit < self.merit + o["filter_armijo_constant"] * expected
        else:
            ok = merit < self.merit - o["filter_merit_acceptance_threshold"] * cv_old or cv_new < (1.0 - o["filter_violation_acceptance_threshold"]) * cv_old
        return dict(X=X, U=U, cost=cost, merit=merit, violation=cv_new, alpha=a) if ok else None

    def record(self):
        self.history.append([self.cost, self.merit, self.alpha_pr, self.inf_du, self.inf_pr, self.mu, self.reg])

    def solve(self):                                    # CDDPSolverBase::solve (cddp_solver_base.cpp:29-186) with LogDDP's hooks
        o = self.o
        self.initialize()
        self.evaluate_and_reset_filter()                # preIterationSetup (:211-214)
        self.record()
        it = 0; status = "MaxIterationsReached"; converged = False
        while it < o["max_iterations"]:
            it += 1
            ok = False
            while not ok:
                ok = self.backward_pass()
                if not ok:
                    self._reg_up()
                    if self.reg >= o["reg_max_value"]:
                        status = "RegularizationLimitReached_Converged"; converged = True    # :216-222
                        break
            if not ok:
                break
            best = None
            for a in self.alphas:                       # performForwardPass, first success (enable_parallel = false)
                r = self.forward_pass(a)
                if r is not None:
                    best = r; break
            if best is not None:
                dJ = self.cost - best["cost"]; dL = self.merit - best["merit"]
                self.X, self.U, self.cost, self.merit, self.alpha_pr = best["X"], best["U"], best["cost"], best["merit"], best["alpha"]
                self.violation = best["violation"]
                self.record()
                self._reg_down()
                if max(self.inf_du, self.inf_pr) <= o["tolerance"]:
                    status = "OptimalSolutionFound"; converged = True
                elif abs(dJ) < o["acceptable_tolerance"] and abs(dL) < o["acceptable_tolerance"]:
                    status = "AcceptableSolutionFound"; converged = True
            else:
                self._reg_up()
                if self.reg >= o["reg_max_value"]:
                    status = "RegularizationLimitReached_NotConverged"; break
            if converged:
                break
            if best is not None: self.mu = max(o["log_mu_min_value"], self.mu * o["log_mu_update_factor"])     # postIterationUpdate (:263-277)
            else: self.mu = min(o["log_mu_initial"], self.mu * 5.0)
            self.reset_filter()
        return dict(iterations=it, status=status, final_objective=self.cost, n_backward=self.n_backward, n_forward=self.n_forward, mu=self.mu)

    # CDDP::increaseRegularization / decreaseRegularization (cddp_core.cpp)
    def _reg_up(self):
        o = self.o
        self.reg = min(self.reg * o["reg_update_factor"], o["reg_max_value"]); return self.reg

    def _reg_down(self):
        o = self.o
        self.reg = max(self.reg / o["reg_update_factor"], o["reg_min_value"])


def backward_full(A, B, lx, lu, lxx, luu, lux, VxN, VxxN, cons, X, U, coeff, delta, reg, hess=None):
    """backward() above with the constraint-curvature term of the barrier Hessian (nonlinear constraint kinds)."""
    N = len(A); nx = A[0].shape[0]; nu = B[0].shape[1]
    V_x = np.array(VxN, float); V_xx = 0.5 * (np.array(VxxN, float) + np.array(VxxN, float).T)
    K = np.zeros((N, nu, nx)); k = np.zeros((N, nu)); Vx = np.zeros((N + 1, nx)); Vxx = np.zeros((N + 1, nx, nx))
    Vx[N] = V_x; Vxx[N] = V_xx
    dV = np.zeros(2); qe = 0.0
    for t in range(N - 1, -1, -1):
        Q_x = lx[t] + A[t].T @ V_x; Q_u = lu[t] + B[t].T @ V_x
        Q_xx = lxx[t] + A[t].T @ V_xx @ A[t]; Q_ux = lux[t] + B[t].T @ V_xx @ A[t]; Q_uu = luu[t] + B[t].T @ V_xx @ B[t]
        if hess is not None:
            Fxx, Fuu, Fux = hess[t]
            for i in range(nx):
                Q_xx = Q_xx + V_x[i] * Fxx[i]; Q_ux = Q_ux + V_x[i] * Fux[i]; Q_uu = Q_uu + V_x[i] * Fuu[i]
        for c in cons:
            gx, gu = barrier_gradients(c, X[t], U[t], coeff, delta)
            Hxx, Huu, Hux = barrier_hessians_full(c, X[t], U[t], coeff, delta)
            Q_x = Q_x + gx; Q_u = Q_u + gu; Q_xx = Q_xx + Hxx; Q_uu = Q_uu + Huu; Q_ux = Q_ux + Hux
        Qr = Q_uu.copy(); Qr[np.diag_indices(nu)] += reg; Qr = 0.5 * (Qr + Qr.T)
        f = EigenLDLT(Qr)
        if not f.ok:
            return False, K, k, Vx, Vxx, dV, qe
        kK = -f.solve(np.concatenate([Q_u.reshape(nu, 1), Q_ux], axis=1))
        k_u = kK[:, 0]; K_u = kK[:, 1:]
        k[t] = k_u; K[t] = K_u
        dV = dV + np.array([float(Q_u @ k_u), 0.5 * float(k_u @ (Q_uu @ k_u))])
        V_x = Q_x + K_u.T @ Q_uu @ k_u + Q_ux.T @ k_u + K_u.T @ Q_u
        V_xx = Q_xx + K_u.T @ Q_uu @ K_u + Q_ux.T @ K_u + K_u.T @ Q_ux
        V_xx = 0.5 * (V_xx + V_xx.T)
        Vx[t] = V_x; Vxx[t] = V_xx
        qe = max(qe, float(np.max(np.abs(Q_u))))
    return True, K, k, Vx, Vxx, dV, qe
